"""Host-side mirror of the reference's Stable-Diffusion wrapper (utils/sd_utils.py:20-295): same class,
attributes and method contracts; the numerics run in libsvg_hip.so (HIP, gfx950).

Weights: the reference fetches 'CompVis/stable-diffusion-v1-4' and 'openai/clip-vit-large-patch14' from the
HF hub by name (sd_utils.py:52-66) — unreachable offline.  Here, in order: an explicit ``weights=`` dict of
diffusers-named state_dicts, a local directory in ``$SVG_SD_WEIGHTS`` (``vae/`` and ``unet/`` holding
``diffusion_pytorch_model.bin|.safetensors``).  When neither provides a network the constructor RAISES like the
reference's failed ``from_pretrained`` — seeded synthetic weights of the exact SD v1.4 architecture are an explicit
opt-in (``weights='synthetic'``, a ``'synthetic'`` entry in the dict, or ``SVG_ALLOW_SYNTHETIC_WEIGHTS=1``) used by the
bench, the smoke test and the parity tests; ``vae_source`` / ``unet_source`` / ``clip_source`` record where each network
came from.  The CLIP text encoder ('openai/clip-vit-large-patch14', sd_utils.py:59-60) is loaded the same way from
``text_encoder/`` (+ ``tokenizer/``) and runs in the library (f32); ``text_embeddings=`` bypasses it.
"""
import os

import numpy as np
import torch

from . import _lib, sd_layout
from .config import parse_config_args

SCALE = 0.18215


def _load_local(dirname, sub):
    for fn in ("diffusion_pytorch_model.safetensors", "diffusion_pytorch_model.bin"):
        p = os.path.join(dirname, sub, fn)
        if os.path.exists(p):
            if fn.endswith(".safetensors"):
                from safetensors.torch import load_file
                return load_file(p)
            return torch.load(p, map_location="cpu")
    return None


def synthetic_allowed():
    return os.environ.get("SVG_ALLOW_SYNTHETIC_WEIGHTS", "0") not in ("", "0")


def _local_arch(dirname, sub):
    """Architecture overrides from a diffusers-format ``<sub>/config.json`` (what from_pretrained reads), or {}."""
    import json
    p = os.path.join(dirname, sub, "config.json")
    if not os.path.exists(p):
        return {}
    with open(p) as f:
        c = json.load(f)
    a = {}
    if "block_out_channels" in c:
        a["block_out"] = tuple(c["block_out_channels"])
    if "layers_per_block" in c:
        a["layers"] = int(c["layers_per_block"])
    if "norm_num_groups" in c:
        a["groups"] = int(c["norm_num_groups"])
    if sub == "vae":
        if "latent_channels" in c:
            a["latent"] = int(c["latent_channels"])
    else:
        if "attention_head_dim" in c:
            a["heads"] = int(c["attention_head_dim"])           # diffusers 0.2.x: the NUMBER of heads
        if "cross_attention_dim" in c:
            a["ctx_dim"] = int(c["cross_attention_dim"])
        if "down_block_types" in c:
            a["attn"] = tuple(1 if "CrossAttn" in t else 0 for t in c["down_block_types"])
        for k_json, k in (("in_channels", "in_ch"), ("out_channels", "out_ch")):
            if k_json in c:
                a[k] = int(c[k_json])
    return a


class _Slot:
    """A network living in a model slot of a library context.  The slot holds one model: if someone else configured it
    since (a second SDUtils handed the same context), calls fail loudly instead of computing with foreign weights."""
    slot = None

    def __init__(self, ctx, n_params):
        self._ctx = ctx
        self.n_params = n_params
        ctx.claim(self.slot, self)

    @property
    def ctx(self):
        if self._ctx.owner(self.slot) is not self:
            raise RuntimeError("the %s slot of this library context now holds another model's weights; give each SDUtils "
                               "its own context (SDUtils(ctx=_lib.Context(dev)))" % type(self).__name__.strip("_"))
        return self._ctx


class _VAE(_Slot):
    """Stands where ``SDUtils.vae`` (diffusers AutoencoderKL) stands: encode(x).sample() / decode(z)."""
    slot = _lib.SVG_VAE

    class _Posterior:
        def __init__(self, ctx, imgs_u8):
            self.ctx, self.imgs = ctx, imgs_u8

        def sample(self, eps=None):
            N, H, W, _ = self.imgs.shape
            if eps is None:
                eps = torch.randn((N, 4, H // 8, W // 8), device=self.imgs.device)
            return self.ctx.vae_encode(self.imgs, eps=eps) / SCALE

        def mode(self):
            return self.ctx.vae_encode(self.imgs) / SCALE

    def encode_u8(self, imgs_u8):
        return _VAE._Posterior(self.ctx, imgs_u8)

    def decode(self, z):
        """z unscaled latents -> float NCHW image in about [-1,1]."""
        return self.ctx.vae_decode(z * SCALE, return_float=True)[1]


class _UNet(_Slot):
    in_channels = 4
    slot = _lib.SVG_UNET

    def __call__(self, sample, timestep, encoder_hidden_states=None):
        return {"sample": self.ctx.unet_forward(sample, timestep, encoder_hidden_states)}


class _CLIPText(_Slot):
    """Stands where ``SDUtils.text_encoder`` (transformers CLIPTextModel) stands: ``text_encoder(input_ids)[0]`` is the
    last hidden state (sd_utils.py:84,91).  f32, like the reference (the text encoder runs outside autocast)."""
    slot = _lib.SVG_CLIP_TEXT

    def __init__(self, ctx, n_params, d_model):
        super().__init__(ctx, n_params)
        self.d_model = d_model

    def __call__(self, input_ids):
        return (self.ctx.clip_text_forward(input_ids, self.d_model),)


class _TokenBatch:
    def __init__(self, ids):
        self.input_ids = ids


class StandInTokenizer:
    """CLIPTokenizer's call surface (sd_utils.py:80-82,87-89) without the hub-only vocab.json / merges.txt: BOS, one
    hashed id per whitespace-separated word, EOS, EOS padding to ``model_max_length``.  For the prompt '' — the only prompt
    of prediction/predict.py:148 — these are exactly the ids the real tokenizer returns (49406, 49407, 49407, ...).  Used
    with synthetic CLIP weights only; a local ``tokenizer/`` directory gets the real byte-pair encoder."""
    model_max_length = 77

    def __init__(self, vocab=49408):
        self.vocab = vocab
        self.bos_token_id, self.eos_token_id = vocab - 2, vocab - 1      # 49406 / 49407 at CLIP's vocabulary size

    def __call__(self, prompt, padding="max_length", max_length=None, truncation=True, return_tensors="pt"):
        import zlib
        L = max_length or self.model_max_length
        prompts = [prompt] if isinstance(prompt, str) else list(prompt)
        rows = []
        for p in prompts:
            ids = [self.bos_token_id] + [zlib.crc32(w.lower().encode()) % (self.vocab - 2) for w in p.split()][: L - 2] + [self.eos_token_id]
            rows.append(ids + [self.eos_token_id] * (L - len(ids)))
        return _TokenBatch(torch.tensor(rows, dtype=torch.long))


def _load_local_clip(dirname):
    for fn in ("model.safetensors", "pytorch_model.bin"):
        p = os.path.join(dirname, "text_encoder", fn)
        if os.path.exists(p):
            if fn.endswith(".safetensors"):
                from safetensors.torch import load_file
                return load_file(p)
            return torch.load(p, map_location="cpu", weights_only=True)
    return None


def _local_clip_arch(dirname):
    import json
    p = os.path.join(dirname, "text_encoder", "config.json")
    if not os.path.exists(p):
        return {}
    with open(p) as f:
        c = json.load(f)
    c = c.get("text_config", c)
    m = {"vocab_size": "vocab", "hidden_size": "d_model", "num_attention_heads": "heads", "num_hidden_layers": "layers",
         "intermediate_size": "ffn", "max_position_embeddings": "max_pos"}
    return {v: int(c[k]) for k, v in m.items() if k in c}


class LMSDiscreteScheduler:
    """The scheduler the reference builds at sd_utils.py:70-72 (diffusers 0.2.3 LMSDiscreteScheduler(beta_start=0.00085,
    beta_end=0.012, beta_schedule='scaled_linear', num_train_timesteps=1000)) and uses in denoise_img_latents
    (sd_utils.py:97-126): sigmas = sqrt((1 - abar) / abar), `set_timesteps` interpolates them on linspace(999, 0, n) and
    appends 0, `step` is the linear multistep update of order <= 4 whose coefficients integrate the Lagrange basis of the last
    sigmas (scipy.integrate.quad, as diffusers does).  Host-side arithmetic on a handful of scalars; the tensors stay on the GPU."""

    def __init__(self, beta_start=0.00085, beta_end=0.012, num_train_timesteps=1000):
        betas = np.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=np.float32) ** 2
        self.alphas_cumprod = np.cumprod(1.0 - betas, axis=0)
        self.train_sigmas = ((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5
        self.num_train_timesteps = num_train_timesteps
        self.set_timesteps(num_train_timesteps)

    def set_timesteps(self, num_inference_steps):
        self.num_inference_steps = num_inference_steps
        self.timesteps = np.linspace(self.num_train_timesteps - 1, 0, num_inference_steps, dtype=float)
        low = np.floor(self.timesteps).astype(int)
        high = np.ceil(self.timesteps).astype(int)
        frac = np.mod(self.timesteps, 1.0)
        sig = (1 - frac) * self.train_sigmas[low] + frac * self.train_sigmas[high]
        self.sigmas = np.concatenate([sig, [0.0]])
        self.derivatives = []

    def get_lms_coefficient(self, order, t, current_order):
        from scipy import integrate

        def lms_derivative(tau):
            prod = 1.0
            for k in range(order):
                if current_order == k:
                    continue
                prod *= (tau - self.sigmas[t - k]) / (self.sigmas[t - current_order] - self.sigmas[t - k])
            return prod
        return integrate.quad(lms_derivative, self.sigmas[t], self.sigmas[t + 1], epsrel=1e-4)[0]

    def step(self, model_output, timestep, sample, order=4):
        sigma = float(self.sigmas[timestep])
        pred_original_sample = sample - sigma * model_output
        self.derivatives.append((sample - pred_original_sample) / sigma)
        if len(self.derivatives) > order:
            self.derivatives.pop(0)
        order = min(timestep + 1, order)
        coeffs = [self.get_lms_coefficient(order, timestep, o) for o in range(order)]
        prev = sample + sum(c * d for c, d in zip(coeffs, reversed(self.derivatives)))
        return {"prev_sample": prev}


class SDUtils():
    def __init__(self, weights=None, text_embeddings=None, seed=0, verbose=True, arch=None, ctx=None, fp8=None, dtype=None, vae_dtype=None):
        self.config, self.args = parse_config_args()             # sd_utils.py:22
        self.device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        if self.device.type != "cuda":
            raise RuntimeError("SDUtils runs on the HIP library and needs a GPU (gfx950); there is no CPU fallback")
        if ctx is None:
            ctx = _lib.default_context()
            # the default context's SD slots are taken by a live SDUtils: this instance gets a context of its own
            # (a full weight replica; 288 GB of HBM make that the cheap answer) instead of replacing the other's weights
            if ctx.owner(_lib.SVG_VAE) is not None or ctx.owner(_lib.SVG_UNET) is not None:
                ctx = _lib.Context(ctx.device.index)
        self.ctx = ctx
        self._seed = seed
        self._verbose = verbose
        self._text_embeddings = text_embeddings
        # BASELINE configs[4]: MX block-scaled fp8 for the qualifying dense projections of the UNet (svg_hip.h, key fp8)
        self.fp8 = bool(int(os.environ.get("SVG_UNET_FP8", "0"))) if fp8 is None else bool(fp8)
        # storage type of the SD networks: 'fp16' (default) is what the reference's autocast executes in the UNet loop
        # (sd_utils.py:246); 'bf16' (BASELINE configs[1] names it) has 3 fewer mantissa bits and a stated, looser tolerance
        #  and runs ~3.6 % faster on random data (the chip holds a higher clock on bf16 operands)
        self.dtype = (dtype or os.environ.get("SVG_SD_DTYPE", "fp16")).lower().replace("float16", "fp16").replace("half", "fp16")
        if self.dtype not in ("bf16", "fp16"):
            raise ValueError("SDUtils dtype must be 'bf16' or 'fp16', got %r" % (self.dtype,))
        # The VAE keeps fp16 storage whatever the UNet runs in: its decoder ends in a uint8 image (sd_utils.py:156-169), and bf16's 8
        # significant bits flip 52 % of the pixels (max 5 LSB) against 7.5 % (max 1 LSB) in fp16 — the reference's own autocast floor
        # is 4.6 % (profiles/r05_vae_decoder_storage.txt).  `dtype` therefore names the UNet's storage (where BASELINE configs[1]'s
        # "bf16" buys its speed: 96 % of the FLOPs); vae_dtype / $SVG_VAE_DTYPE = 'bf16' restores the all-bf16 arithmetic for A/B.
        self.vae_dtype = (vae_dtype or os.environ.get("SVG_VAE_DTYPE", "fp16")).lower().replace("float16", "fp16").replace("half", "fp16")
        if self.vae_dtype not in ("bf16", "fp16"):
            raise ValueError("SDUtils vae_dtype must be 'bf16' or 'fp16', got %r" % (self.vae_dtype,))
        # `arch` overrides the SD v1.4 widths (reduced-size parity tests): {'vae': {...}, 'unet': {...}}
        # (a local diffusers directory's config.json plays the same role, as it does for from_pretrained)
        local = os.environ.get("SVG_SD_WEIGHTS")
        self.vae_arch = dict(sd_layout.SD_VAE, **(_local_arch(local, "vae") if local else {}))
        self.unet_arch = dict(sd_layout.SD_UNET, **(_local_arch(local, "unet") if local else {}))
        self.clip_arch = dict(sd_layout.SD_CLIP, **(_local_clip_arch(local) if local else {}))
        self.vae_arch.update((arch or {}).get('vae', {}))
        self.unet_arch.update((arch or {}).get('unet', {}))
        self.clip_arch.update((arch or {}).get('clip', {}))
        vae, tokenizer, text_encoder, unet, scheduler = self.load_models(weights)
        self.vae = vae
        # sd_utils.py:30 builds a default-size Transformer and throws it away; it consumes CPU RNG, which matters
        # to anyone seeding before construction — reproduced by drawing the same parameters.
        from .transformer import Transformer
        Transformer()
        self.SOS_token = torch.ones((1, 1, self.config.FRAME_SIZE ** 2 // 64 * 4), dtype=torch.float32, device=self.device) * 2
        self.tokenizer = tokenizer
        self.text_encoder = text_encoder
        self.unet = unet
        self.scheduler = scheduler

    # ---- sd_utils.py:39-76 ---------------------------------------------------------------------------
    def _weights_for(self, name, given, shapes_fn, seed):
        synthetic = given == "synthetic" or synthetic_allowed()
        if isinstance(given, dict) and name in given:
            if not isinstance(given[name], str):
                return given[name], "given"
            if given[name] != "synthetic":
                raise ValueError("weights['%s'] must be a state_dict or 'synthetic'" % name)
            synthetic = True
        d = os.environ.get("SVG_SD_WEIGHTS")
        if d:
            sd = _load_local(d, name)
            if sd is not None:
                return sd, "local:" + d
        if not synthetic:
            # the reference's from_pretrained raises here too (sd_utils.py:52-66 with no hub access)
            raise FileNotFoundError(
                "no %s weights: set $SVG_SD_WEIGHTS to a diffusers-format directory (%s/diffusion_pytorch_model.safetensors|.bin), "
                "pass weights={'%s': state_dict}, or opt in to seeded synthetic weights (weights='synthetic' or "
                "SVG_ALLOW_SYNTHETIC_WEIGHTS=1)" % (name, name, name))
        if self._verbose:
            print("[sd-video-gen] %s: seeded SYNTHETIC weights (seed %d) — outputs are not images" % (name, seed))
        return sd_layout.seeded_weights(shapes_fn(), seed, device=self.device), "synthetic"

    def load_models(self, weights=None):
        ctx = self.ctx
        va = self.vae_arch
        sd, self.vae_source = self._weights_for("vae", weights, lambda: sd_layout.vae_shapes(va), self._seed + 1)
        ctx.configure(_lib.SVG_VAE, block_out=list(va["block_out"]), layers=va["layers"], groups=va["groups"], latent=4,
                      f16=int(self.vae_dtype == "fp16"))
        ctx.load_state_dict(_lib.SVG_VAE, sd)
        vae = _VAE(ctx, ctx.finalize(_lib.SVG_VAE))
        del sd
        self.unet_source = self.clip_source = None
        if not self.args.denoise:
            return vae, None, None, None, None
        # sd_utils.py:59-60: tokenizer + CLIP text encoder (skipped when the caller supplies the embeddings)
        tokenizer = text_encoder = None
        if self._text_embeddings is None:
            tokenizer, text_encoder = self._load_clip(weights)
        c = self.unet_arch
        sd, self.unet_source = self._weights_for("unet", weights, lambda: sd_layout.unet_shapes(c), self._seed + 2)
        ctx.configure(_lib.SVG_UNET, block_out=list(c["block_out"]), layers=c["layers"], heads=c["heads"], ctx_dim=c["ctx_dim"],
                      groups=c["groups"], in_ch=4, out_ch=4, attn=list(c["attn"]), fp8=int(self.fp8),
                      f16=int(self.dtype == "fp16"))
        ctx.load_state_dict(_lib.SVG_UNET, sd)
        unet = _UNet(ctx, ctx.finalize(_lib.SVG_UNET))
        del sd
        # sd_utils.py:70-72: the LMS scheduler of the text-to-image sampler (denoise_img_latents); gen_i2i_latents builds its own
        # DDIM schedule (sd_utils.py:233-237), which lives in the library
        return vae, tokenizer, text_encoder, unet, LMSDiscreteScheduler()

    def _load_clip(self, weights):
        ctx, c = self.ctx, self.clip_arch
        d = os.environ.get("SVG_SD_WEIGHTS")
        sd = None
        if isinstance(weights, dict) and "text_encoder" in weights and not isinstance(weights["text_encoder"], str):
            sd, self.clip_source = weights["text_encoder"], "given"
        elif d:
            sd = _load_local_clip(d)
            if sd is not None:
                self.clip_source = "local:" + d
        if sd is None:
            # with any other UNet than the synthetic one a random text encoder would silently condition every
            # cross-attention on noise: refuse like a failed from_pretrained
            allowed = weights == "synthetic" or synthetic_allowed() or (isinstance(weights, dict) and weights.get("text_encoder") == "synthetic")
            if not allowed:
                raise FileNotFoundError("no CLIP text-encoder weights: put text_encoder/model.safetensors (+ tokenizer/) under "
                                        "$SVG_SD_WEIGHTS, pass weights={'text_encoder': state_dict}, or text_embeddings=")
            if self._verbose:
                print("[sd-video-gen] text_encoder: seeded SYNTHETIC weights (seed %d)" % (self._seed + 3))
            sd, self.clip_source = sd_layout.seeded_weights(sd_layout.clip_text_shapes(c), self._seed + 3, device=self.device), "synthetic"
        sd = {(k[len("text_model."):] if k.startswith("text_model.") else k): v for k, v in sd.items() if "position_ids" not in k}
        ctx.configure(_lib.SVG_CLIP_TEXT, vocab=c["vocab"], d_model=c["d_model"], heads=c["heads"], layers=c["layers"], ffn=c["ffn"],
                      max_pos=c["max_pos"])
        ctx.load_state_dict(_lib.SVG_CLIP_TEXT, sd)
        text_encoder = _CLIPText(ctx, ctx.finalize(_lib.SVG_CLIP_TEXT), c["d_model"])
        tokenizer = None
        if d and os.path.exists(os.path.join(d, "tokenizer", "vocab.json")):
            from transformers import CLIPTokenizer              # host-side string processing, as in the reference (sd_utils.py:59)
            tokenizer = CLIPTokenizer.from_pretrained(os.path.join(d, "tokenizer"))
        elif self.clip_source == "synthetic" or synthetic_allowed():
            tokenizer = StandInTokenizer(c["vocab"])
            tokenizer.model_max_length = c["max_pos"]
        else:
            raise FileNotFoundError("CLIP tokenizer files (tokenizer/vocab.json, merges.txt) not found under $SVG_SD_WEIGHTS")
        return tokenizer, text_encoder

    # ---- sd_utils.py:78-95 -----------------------------------------------------------------------------
    def encode_text(self, prompt):
        """-> (2*len(prompt), 77, 768) = [uncond ; text] (sd_utils.py:78-95): tokenizer (pad to model_max_length, truncate),
        CLIP text model's last hidden state for the prompts and for '' * len(prompt)."""
        if self._text_embeddings is not None:
            return self._text_embeddings.to(self.device)
        if self.text_encoder is None:
            raise RuntimeError("encode_text needs the text encoder: construct SDUtils with --denoise")
        if isinstance(prompt, str):
            prompt = [prompt]
        text_input = self.tokenizer(prompt, padding="max_length", max_length=self.tokenizer.model_max_length, truncation=True,
                                    return_tensors="pt")
        with torch.no_grad():
            text_embeddings = self.text_encoder(text_input.input_ids.to(self.device))[0]
        uncond_input = self.tokenizer([""] * len(prompt), padding="max_length", max_length=self.tokenizer.model_max_length,
                                      return_tensors="pt")
        with torch.no_grad():
            uncond_embeddings = self.text_encoder(uncond_input.input_ids.to(self.device))[0]
        return torch.cat([uncond_embeddings, text_embeddings])

    # ---- sd_utils.py:97-126 ----------------------------------------------------------------------------
    def denoise_img_latents(self, text_embeddings, height=512, width=512, num_inference_steps=50, guidance_scale=7.5, latents=None,
                            start_step=None):
        """text-conditioned LMS sampling from noise (the reference's text-to-image path): UNet calls at batch 2N in the library,
        the classifier-free-guidance combine and the multistep update on the (N,4,h,w) latents as device tensor arithmetic."""
        if self.unet is None:
            raise RuntimeError("denoise_img_latents needs the UNet: construct SDUtils with --denoise")
        if latents is None:
            latents = torch.randn((text_embeddings.shape[0] // 2, self.unet.in_channels, height // 8, width // 8))
        latents = latents.to(self.device).float()
        text_embeddings = text_embeddings.to(self.device)
        self.scheduler.set_timesteps(num_inference_steps)
        latents = latents * float(self.scheduler.sigmas[0])
        with torch.no_grad():
            for i, t in enumerate(self.scheduler.timesteps):
                latent_model_input = torch.cat([latents] * 2)
                sigma = float(self.scheduler.sigmas[i])
                latent_model_input = latent_model_input / ((sigma ** 2 + 1) ** 0.5)
                noise_pred = self.unet(latent_model_input, float(t), encoder_hidden_states=text_embeddings)["sample"]
                noise_pred_uncond, noise_pred_text = noise_pred.chunk(2)
                noise_pred = noise_pred_uncond + guidance_scale * (noise_pred_text - noise_pred_uncond)
                latents = self.scheduler.step(noise_pred, i, latents)["prev_sample"]
        return latents

    # ---- sd_utils.py:171-189 ---------------------------------------------------------------------------
    def prompt_to_img(self, prompts, height=512, width=512, num_inference_steps=50, guidance_scale=7.5, latents=None):
        if isinstance(prompts, str):
            prompts = [prompts]
        text_embeds = self.encode_text(prompts)
        latents = self.denoise_img_latents(text_embeds, height=height, width=width, latents=latents,
                                           num_inference_steps=num_inference_steps, guidance_scale=guidance_scale)
        return self.decode_img_latents(latents)

    # ---- sd_utils.py:128-154 ---------------------------------------------------------------------------
    def encode_img(self, imgs, eps=None):
        """imgs (N,H,W,3) uint8 (tensor, any device) -> (N,4,H/8,W/8) scaled latents on device.
        ``eps`` = the .sample() draws (sd_utils.py:142); drawn on the device generator when None."""
        imgs = torch.as_tensor(imgs).to(self.device)
        if imgs.dtype != torch.uint8:
            imgs = imgs.round().clamp(0, 255).to(torch.uint8)
        N, H, W, _ = imgs.shape
        if eps is None:
            eps = torch.randn((N, 4, H // 8, W // 8), device=self.device)
        return self.vae.ctx.vae_encode(imgs, eps=eps)

    def encode_batch(self, img_batch, use_sos=True, eps=None):
        img_batch = torch.as_tensor(img_batch)
        new_batch = self.encode_img(img_batch.reshape(-1, img_batch.shape[2], img_batch.shape[3], img_batch.shape[4]), eps)
        new_batch = new_batch.reshape(img_batch.shape[0], img_batch.shape[1], -1)
        if use_sos:
            SOS_token = self.SOS_token.repeat(new_batch.shape[0], 1, 1)
            new_batch = torch.cat((SOS_token, new_batch), dim=1)
        return new_batch

    # ---- sd_utils.py:156-169 ---------------------------------------------------------------------------
    def decode_img_latents(self, latents):
        """-> numpy (N,8h,8w,3) uint8 on the HOST, like the reference (callers do np.array(img[0]))."""
        return self.vae.ctx.vae_decode(latents.to(self.device)).cpu().numpy()

    def decode_img_latents_device(self, latents, out_hw=None):
        """same frames, kept on the device (no D2H sync), optionally nearest-resized (predict.py:158,178)."""
        return self.vae.ctx.vae_decode(latents.to(self.device), out_hw=out_hw)

    # ---- sd_utils.py:222-267 ---------------------------------------------------------------------------
    def gen_i2i_latents(self, text_embeddings, height=512, width=512, num_inference_steps=50, guidance_scale=7.5,
                        latents=None, return_all_latents=False, start_step=10, noise=None):
        if self.unet is None:
            raise RuntimeError("gen_i2i_latents needs the UNet: construct SDUtils with --denoise")
        if latents is None:
            latents = torch.randn((text_embeddings.shape[0] // 2, self.unet.in_channels, height // 8, width // 8))
        latents = latents.to(self.device)
        if start_step > 0 and noise is None:
            noise = torch.randn_like(latents)
        return self.unet.ctx.ddim_loop(latents, text_embeddings.to(self.device), num_steps=num_inference_steps, start_step=start_step,
                                  guidance=guidance_scale, noise=noise, return_hist=return_all_latents)

    def perturb_latents(self, latents, scale=0.1):
        noise = torch.randn_like(latents)
        new_latents = (1 - scale) * latents + scale * noise
        return (new_latents - new_latents.mean()) / new_latents.std()

    def img_to_img(self, prompts, height=512, width=512, num_inference_steps=50, guidance_scale=7.5, img=None,
                   return_all_latents=False, batch_size=2, start_step=10):
        if isinstance(prompts, str):
            prompts = [prompts]
        lat = self.encode_img(img)
        emb = self.encode_text(prompts)
        out = self.gen_i2i_latents(emb, height, width, num_inference_steps, guidance_scale, lat, return_all_latents, start_step)
        imgs = []
        for i in range(0, len(out), batch_size):
            imgs.extend(self.decode_img_latents(out[i:i + batch_size]))
        return imgs
