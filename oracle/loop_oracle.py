"""ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/transformer_oracle.py header).

CPU restatement of the reference's per-clip sampling loop, prediction/predict.py:117-197, for ONE clip, built from
the oracle networks, with every random draw passed in explicitly (the reference never seeds):
  :124      encode_batch(batch, use_sos=True)          -> noise["cond"]  (5,4,L,L)
  :144      predict(model, X)
  :149-159  decode -> uint8 -> nearest resize to R x R (R = 512 in the reference)
  :163-164  encode_batch(use_sos=False)                -> noise["e512"][k] (4,R/8,R/8)
  :168-170  gen_i2i_latents(guidance 0, start_step S)  -> noise["add"][k]  (4,R/8,R/8) when S > 0
  :173-179  decode -> uint8 -> nearest resize to F x F
  :183-185  encode_batch(use_sos=False)                -> noise["eF"][k]   (4,L,L)
  :187-196  append, drop the last conditioning frame, window of 5
"""
import torch

from . import sd_oracle as SO
from . import transformer_oracle as TO


def sample_clip(xf_sd, num_heads, vae_sd, clip_u8, pred_frames, noise, denoise=False, start_step=40, unet_sd=None,
                text_emb=None, vae_cfg=SO.SD_VAE, unet_cfg=SO.SD_UNET, res=512, num_inference_steps=50, txt=None,
                guidance_scale=0.0, unet=None, trace=None, latent_denoise=False):
    """clip_u8 (5,F,F,3) uint8 -> all_latents (1, 4+N, D_lat).  `txt` (1,384): the class embedding of the
    text-conditioned loop (prediction/predict_text.py:186-262 — the same loop with predict(model, X, cls_list)).
    `guidance_scale`: 0 at predict.py:169, 7.5 at evaluation/predict_fvd2_denoise.py:228.  `unet(x, t, ctx)` may stand in
    for the oracle UNet (fixture generation runs the duplicated batch once).  `trace` (a list) receives, per predicted
    frame, a dict with the Transformer prediction, the latent entering the DDIM loop and the loop's latent history."""
    T, F = clip_u8.shape[0], clip_u8.shape[1]
    down = 2 ** (len(vae_cfg["block_out"]) - 1)
    L = F // down
    D = 4 * L * L
    z = SO.encode_img(vae_sd, clip_u8, noise["cond"], vae_cfg).reshape(1, T, D)
    X = torch.cat((2.0 * torch.ones(1, 1, D), z), dim=1)
    inputs = z
    preds = torch.zeros(1, 0, D)
    all_latents = None
    for k in range(pred_frames):
        pred = TO.predict(xf_sd, X, num_heads, txt=txt)
        if denoise:
            if latent_denoise:      # evaluation/predict_fvd.py:163-165: the latent itself goes to the 512-pixel latent grid
                lat = torch.nn.functional.interpolate(pred.reshape(1, 4, L, L), (res // down, res // down), mode="bilinear")
            else:
                img = SO.decode_img_latents(vae_sd, pred.reshape(1, 4, L, L), vae_cfg)
                big = SO.resize_nearest_u8(img, res, res)
                lat = SO.encode_img(vae_sd, big, noise["e512"][k][None], vae_cfg)
            hist = SO.gen_i2i_latents(unet_sd, text_emb, lat, num_inference_steps, guidance_scale, start_step,
                                      noise=noise["add"][k][None] if start_step > 0 else None, cfg=unet_cfg,
                                      return_all_latents=True, unet=unet)
            den = hist[-1:]
            if trace is not None:
                trace.append({"pred": pred.clone(), "lat0": lat.clone(), "hist": hist.clone()})
                if not latent_denoise:
                    trace[-1]["img"] = img.clone()
            img2 = SO.decode_img_latents(vae_sd, den, vae_cfg)
            small = SO.resize_nearest_u8(img2, F, F)
            pred = SO.encode_img(vae_sd, small, noise["eF"][k][None], vae_cfg).flatten()
            if trace is not None:
                trace[-1].update(small=small.clone(), out=pred.clone())
        preds = torch.cat((preds, pred.reshape(1, 1, D)), dim=1)
        all_latents = torch.cat([inputs[:, :-1], preds], dim=1)
        X = all_latents[:, -5:]
    return all_latents
