"""The sampling loop of the reference (prediction/predict.py:16-42 and :117-197), host side.

``predict(model, X)`` and the per-clip loop keep the reference's semantics, quirks included
(SURVEY §9): src == tgt, SOS only on the first iteration, window hard-coded to 5, last conditioning
frame dropped from the emitted clip.  The clip-batched loop runs C independent clips in lock step
(every GEMM gets C times the rows); each clip still sees PE row 0 exactly as at batch 1.
"""
import os

import torch


def predict(model, input_sequence, pe_row=None, cls_list=None):
    """prediction/predict.py:16-42 -> (D_lat,) for batch row 0 (or (B,D_lat) rows when ``pe_row`` is given).
    With ``cls_list`` (class names or a (B,384) tensor) the model is the text-conditioned one and the call is
    prediction/predict_text.py:48-74."""
    model.eval()
    with torch.no_grad():
        tgt_mask = model.get_tgt_mask(input_sequence.size(1)).to(input_sequence.device)
        lead = (input_sequence,) if cls_list is None else (input_sequence, cls_list)
        pred = model(*lead, input_sequence, tgt_mask, pe_row=pe_row) if pe_row is not None \
            else model(*lead, input_sequence, tgt_mask)
        pred = pred.permute(1, 0, 2)                       # (B, T, D)
    if pe_row is not None:
        return pred[:, -1]
    return pred[0, -1]


def clip_noise(seeds, shape, device):
    """One standard-normal draw of `shape` per clip from that clip's own generator, stacked: results do not
    depend on how clips are batched or sharded over ranks.  (Host generators draw on the host and are copied over.)"""
    return torch.stack([torch.randn(shape, generator=g, device=g.device) for g in seeds]).to(device)


def sample_clips(model, sd_utils, clips_u8, pred_frames, denoise=False, start_step=40, seeds=None,
                 text_embeddings=None, num_inference_steps=50, guidance_scale=0.0, return_frames=False, res=512,
                 cls_list=None, cpu_noise=False, latent_denoise=False, _planning=False):
    """The per-clip loop of prediction/predict.py:117-197 for C independent clips in lock step, device resident.

    clips_u8: (C,T,F,F,3) uint8 conditioning frames on the device (T = 5 in prediction/predict.py; the FVD loop of
    prediction/predict_text.py conditions on 16-frame UCF clips: the first forward sees SOS + T tokens, T <= 31).  Every stage is batched over clips; a clip's
    result equals running it alone (PE row 0 per clip, per-clip noise generators seeded ``seeds[c]``).  The noise
    draws follow the reference's order: VAE sample of the 5 conditioning frames; then per predicted frame the VAE
    sample @512, add_noise (start_step>0), the VAE sample @F.  The four host crossings per frame of the reference
    (decode -> numpy -> tensor -> interpolate, twice) become fused on-device uint8 nearest resizes with identical
    rounding.  ``cls_list`` (one class name per clip, or a (C,384) tensor) selects the text-conditioned loop of
    prediction/predict_text.py:186-262 (same loop, `predict(model, X, cls_list)`); the names are encoded once.
    ``latent_denoise``: the variant of evaluation/predict_fvd.py:160-178 — the predicted latent itself is resized (bilinear) to the
    512-pixel latent grid and denoised, instead of being decoded, resized as an image and re-encoded first.
    ``cpu_noise``: the per-clip generators live on the host (bit-reproducible on a machine without the GPU: the committed
    oracle fixtures of tests/golden/sd_*.pt were drawn that way); default is the device generator, like the reference.
    Workspace: the first call of a given shape signature on a context first runs itself in the library's planning mode (every model
    call records its workspace need, nothing is launched) and sizes the context's arena ONCE for the largest; the real calls then
    allocate nothing (SURVEY 8(b) Ownership; `ctx.workspace_growths()` stays constant).
    Returns all_latents (C, 4+N, D_lat) f32 [and the decoded frames (C,4+N,F,F,3) uint8].
    """
    ctx = sd_utils.vae.ctx                      # (checked: the context's slots still hold this SDUtils' networks)
    if denoise:
        assert sd_utils.unet is not None and sd_utils.unet.ctx is ctx, "sample_clips(denoise=True) needs SDUtils built with --denoise"
    if not _planning:
        plan_workspace(model, sd_utils, clips_u8, pred_frames, denoise=denoise, start_step=start_step, text_embeddings=text_embeddings,
                       num_inference_steps=num_inference_steps, guidance_scale=guidance_scale, return_frames=return_frames, res=res,
                       cls_list=cls_list, latent_denoise=latent_denoise)
    dev = clips_u8.device
    C, T, F = clips_u8.shape[0], clips_u8.shape[1], clips_u8.shape[2]
    assert 2 <= T <= 31, "conditioning frames: 5 in predict.py:57, 16 in predict_text.py:133 (the library serves sequences up to 32 tokens)"
    L = F // 8
    D = 4 * L * L
    if seeds is None:
        seeds = list(range(C))
    gens = [torch.Generator(device="cpu" if cpu_noise else dev).manual_seed(int(s)) for s in seeds]
    model.eval()
    with torch.no_grad():
        eps = clip_noise(gens, (T, 4, L, L), dev).reshape(C * T, 4, L, L)
        z = ctx.vae_encode(clips_u8.reshape(C * T, F, F, 3), eps=eps).reshape(C, T, D)     # predict.py:124
        X = torch.cat((sd_utils.SOS_token.repeat(C, 1, 1), z), dim=1)
        inputs = z                                                                          # :136-141
        pe0 = torch.zeros(C, dtype=torch.int32, device=dev)
        if cls_list is not None and not isinstance(cls_list, torch.Tensor):
            cls_list = model.encode_classes(cls_list).to(dev)
        preds = []
        if denoise:
            emb = text_embeddings if text_embeddings is not None else sd_utils.encode_text([""])   # :148 (constant: hoisted)
            n = emb.shape[0] // 2
            if n == 1 and C > 1:
                emb = torch.cat([emb[:1].repeat(C, 1, 1), emb[1:].repeat(C, 1, 1)])
        all_latents = None
        for _ in range(pred_frames):
            pred = predict(model, X, pe_row=pe0, cls_list=cls_list)                         # :144  (C, D)
            if denoise:
                if latent_denoise:
                    resized = ctx.resize_bilinear_f32(pred.reshape(C, 4, L, L), res // 8, res // 8)   # predict_fvd.py:163-165
                else:
                    noisy_img = ctx.vae_decode(pred.reshape(C, 4, L, L))                    # :149-153 (uint8, on device)
                    e512 = clip_noise(gens, (4, res // 8, res // 8), dev)
                    resized = ctx.vae_encode(noisy_img, H=res, W=res, eps=e512)             # :158 resize + :163-164
                noise = clip_noise(gens, (4, res // 8, res // 8), dev) if 0 < start_step else None
                den = ctx.ddim_loop(resized, emb, num_steps=num_inference_steps, start_step=start_step,
                                    guidance=guidance_scale, noise=noise)                   # :168-170
                small = ctx.vae_decode(den, out_hw=(F, F))                                  # :173-179
                eF = clip_noise(gens, (4, L, L), dev)
                pred = ctx.vae_encode(small, eps=eF).reshape(C, D)                          # :183-185
            preds.append(pred)
            all_latents = torch.cat([inputs[:, :-1], torch.stack(preds, dim=1)], dim=1)     # :193
            X = all_latents[:, -5:]                                                         # :196
        frames = None
        if return_frames:
            n_out = all_latents.shape[1]
            frames = ctx.vae_decode(all_latents.reshape(C * n_out, 4, L, L)).reshape(C, n_out, F, F, 3)   # :208-211
        if not _planning and C * (T + 1) <= min(176, int(os.environ.get("SVG_XF_WALK_ROWS", "96") or 96)):
            # small batches may have taken the layer-walking Transformer launch (up to $SVG_XF_WALK_ROWS rows, default 96), which can give up
            # under contention and NaN-fill its output: surface that HERE, where the clip is handed back, not at some later call
            # (svg_transformer_status).  Larger batches never take the walk: no synchronisation is added to their steps.
            mctx = getattr(model, "_ctx", None)
            if mctx is not None:
                mctx.transformer_status(sync=True)
        return (all_latents, frames) if return_frames else all_latents


def plan_workspace(model, sd_utils, clips_u8, pred_frames, **kw):
    """Size the context's workspace once for `sample_clips(model, sd_utils, clips_u8, pred_frames, **kw)`: the same call sequence in
    the library's planning mode (svg_plan_begin .. svg_plan_end: nothing is launched), once per shape signature and context.
    Returns True when it planned, False when that signature was already planned."""
    ctx = sd_utils.vae.ctx
    emb, cl = kw.get("text_embeddings"), kw.get("cls_list")
    key = ("sample_clips", id(model), tuple(clips_u8.shape), int(pred_frames), bool(kw.get("denoise")), int(kw.get("start_step", 40)),
           int(kw.get("num_inference_steps", 50)), float(kw.get("guidance_scale", 0.0)) != 0.0, bool(kw.get("return_frames")), int(kw.get("res", 512)),
           bool(kw.get("latent_denoise")), None if emb is None else tuple(emb.shape), cl is not None)
    kw = {k: v for k, v in kw.items() if k not in ("seeds", "cpu_noise")}
    seen = ctx.__dict__.setdefault("_planned", set())
    if key in seen:
        return False
    import contextlib
    from . import _lib
    mctx = getattr(model, "_bound_ctx", None) or _lib.default_context()      # the latent Transformer may live on another context
    with contextlib.ExitStack() as st:
        for c in ([ctx] if mctx is ctx else [ctx, mctx]):
            st.enter_context(c.planning())
        sample_clips(model, sd_utils, clips_u8, pred_frames, _planning=True, **kw)
    seen.add(key)
    return True


def sample_clips_streams(workers, clips_u8, pred_frames, seeds, **kw):
    """Run `sample_clips` on several (model, sd_utils, stream) workers at once: the clips are split into contiguous
    groups, each group is driven by its own host thread on its own HIP stream and library context (full weight
    replica each), so kernels of different groups can share the GPU (ALU/HBM-bound normalisation and softmax work of
    one group under the MFMA-bound convs of another).  Results equal one call on all clips up to the 16-bit rounding that a
    different rows-per-launch count brings (tile width / split-K selection); the f32 latent-Transformer part is bitwise equal
    (tests/test_streams_gpu.py: reduced width with denoising, and full size with start_step 48).  ``text_embeddings`` of per-clip
    form (2*C rows: [uncond(C); cond(C)]) are sliced per group.
    Every group's workspace is planned on the CALLING thread before the worker threads start (plan_workspace): a DDIM loop captures
    its step as a hipGraph, and no thread may allocate-and-free or synchronise the device while another one is capturing."""
    import threading
    n = clips_u8.shape[0]
    G = len(workers)
    bounds = [(n * g) // G for g in range(G + 1)]
    out = [None] * G
    errs = []
    cur = torch.cuda.current_stream()

    def group_kw(g):
        a, b = bounds[g], bounds[g + 1]
        kwg = dict(kw)
        emb = kwg.get("text_embeddings")
        if emb is not None and emb.shape[0] == 2 * n and n > 1:      # per-clip embeddings: this group's rows of each half
            kwg["text_embeddings"] = torch.cat([emb[a:b], emb[n + a:n + b]])
        cl = kwg.get("cls_list")
        if cl is not None:
            kwg["cls_list"] = cl[a:b]
        return kwg

    for g in range(G):
        if bounds[g + 1] > bounds[g]:
            plan_workspace(workers[g][0], workers[g][1], clips_u8[bounds[g]:bounds[g + 1]], pred_frames, **group_kw(g))

    def run(g):
        try:
            model, sdu, stream = workers[g]
            a, b = bounds[g], bounds[g + 1]
            stream.wait_stream(cur)
            with torch.cuda.stream(stream):
                out[g] = sample_clips(model, sdu, clips_u8[a:b], pred_frames, seeds=seeds[a:b], **group_kw(g))
        except Exception as e:      # surfaced on the caller's thread
            errs.append(e)
    threads = [threading.Thread(target=run, args=(g,)) for g in range(G) if bounds[g + 1] > bounds[g]]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errs:
        raise errs[0]
    for _, _, stream in workers:
        cur.wait_stream(stream)
    parts = [o for o in out if o is not None]
    if isinstance(parts[0], tuple):
        return tuple(torch.cat([p[i] for p in parts]) for i in range(len(parts[0])))
    return torch.cat(parts)


def bouncing_ball_clips(n_clips, frame_size, n_frames=5, seed=0, device="cpu"):
    """Synthetic stand-in for the bouncing-ball dataset (datasets are absent offline): black F x F x 3 uint8
    background, one white disc of radius F/8 moving at constant velocity with elastic wall bounces.
    Returns (n_clips, n_frames, F, F, 3) uint8, BGR == RGB (grey)."""
    import numpy as np
    F = frame_size
    out = np.zeros((n_clips, n_frames, F, F, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:F, 0:F]
    r = F / 8.0
    for c in range(n_clips):
        rng = np.random.default_rng(seed + c)
        pos = rng.uniform(r, F - r, size=2)
        vel = rng.uniform(-F / 10.0, F / 10.0, size=2)
        for t in range(n_frames):
            out[c, t][(yy - pos[0]) ** 2 + (xx - pos[1]) ** 2 <= r * r] = 255
            pos = pos + vel
            for a in range(2):
                if pos[a] < r:
                    pos[a], vel[a] = 2 * r - pos[a], -vel[a]
                if pos[a] > F - r:
                    pos[a], vel[a] = 2 * (F - r) - pos[a], -vel[a]
    return torch.from_numpy(out).to(device)


def rollout_latents(model, new_batch, pred_frames, post=None):
    """predict.py:124-197 for one clip with the VAE taken out: ``new_batch`` (1,6,D) = SOS + 5 encoded frames.
    ``post(pred)`` is the optional denoise round trip (predict.py:145-185).  Returns (all_latents, trace)."""
    X = new_batch
    inputs = new_batch[:, 1:]
    preds = new_batch.new_zeros((1, 0, new_batch.shape[-1]))
    trace, all_latents = [], None
    for _ in range(pred_frames):
        shape_in = tuple(X.shape)
        pred = predict(model, X)
        if post is not None:
            pred = post(pred)
        preds = torch.cat((preds, pred.reshape(1, 1, -1)), dim=1)
        all_latents = torch.cat([inputs[:, :-1], preds], dim=1)      # predict.py:193
        X = all_latents[:, -5:]                                      # predict.py:196
        trace.append((shape_in, tuple(all_latents.shape)))
    return all_latents, trace


def run_sharded(clips, sample_fn, base_seed=0):
    """The N>1 form of the per-clip loop: rank r samples clips ``shard_range(n, r, W)`` with per-clip seeds
    ``base_seed + clip`` (world-size invariant) through ``sample_fn(clips_local, seeds) -> tuple of per-clip tensors``,
    then ONE all-gather reassembles every output in clip order on every rank.  World size 1: no collective."""
    from . import sharding
    rank, ws = sharding.world()
    n = clips.shape[0]
    a, b = sharding.shard_range(n, rank, ws)
    out = sample_fn(clips[a:b], sharding.clip_seeds(base_seed, a, b))
    return sharding.gather_clips_packed(list(out), n)


# ---- `python -m prediction.predict` (reference prediction/predict.py:44-247) ---------------------------------------
def _png_clips(folder, frame_size, n_frames=5):
    """Fixed-length clips of consecutive PNG frames under `folder` (one clip per sub-directory run), BGR like cv2.imread."""
    import glob
    import os
    import numpy as np
    from PIL import Image
    clips = []
    dirs = sorted({os.path.dirname(p) for p in glob.glob(os.path.join(folder, "**", "*.png"), recursive=True)})
    for d in dirs:
        files = sorted(glob.glob(os.path.join(d, "*.png")))
        for i in range(0, len(files) - n_frames + 1, n_frames):
            fr = [np.asarray(Image.open(f).convert("RGB").resize((frame_size, frame_size), Image.NEAREST))[..., ::-1] for f in files[i:i + n_frames]]
            clips.append(np.stack(fr))
    if not clips:
        raise ValueError("no PNG clips of %d frames under %s" % (n_frames, folder))
    return torch.from_numpy(np.ascontiguousarray(np.stack(clips)))


def save_frames(frames_u8, is_pred, out_dir):
    """predict.py:201-223: one PNG per frame, predicted frames get a 1-px red border (BGR [0,0,255])."""
    import os
    import numpy as np
    from PIL import Image
    os.makedirs(out_dir, exist_ok=True)
    for i, img in enumerate(frames_u8):
        img = np.asarray(img)
        if is_pred[i]:
            b = np.zeros((img.shape[0] + 2, img.shape[1] + 2, 3), dtype=np.uint8)
            b[..., 2] = 255
            b[1:-1, 1:-1] = img
            img = b
        Image.fromarray(np.ascontiguousarray(img[..., ::-1])).save(os.path.join(out_dir, "%d.png" % i))


def main(argv=None):
    """Same flags and flow as the reference's __main__; datasets: 'synthetic-ball' (built in) or a folder of PNG
    clips via --folder.  Clips are sharded over ranks when launched under torch.distributed.run."""
    import os
    import numpy as np
    from . import config as svg_config, sharding
    from .sd_utils import SDUtils
    from .transformer import Transformer
    if argv is not None:
        svg_config.set_args(argv)
    config, args = svg_config.parse_config_args()
    if "RANK" in os.environ and int(os.environ.get("WORLD_SIZE", "1")) > 1:
        import torch.distributed as dist
        # SVG_DEVICE_OVERRIDE / SVG_DIST_BACKEND=gloo: rehearsal of the N>1 path on a one-GPU box (as in bench.py)
        torch.cuda.set_device(int(os.environ.get("SVG_DEVICE_OVERRIDE", os.environ.get("LOCAL_RANK", "0"))))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(os.environ.get("SVG_DIST_BACKEND", "nccl"))
    rank, ws = sharding.world()
    sd_utils = SDUtils(verbose=(rank == 0))
    device = sd_utils.device
    model = Transformer(num_tokens=0, dim_model=config.DIM_MODEL[0], num_heads=config.NUM_HEADS[0],
                        num_encoder_layers=config.NUM_ENCODER_LAYERS[0], num_decoder_layers=config.NUM_DECODER_LAYERS[0],
                        dropout_p=config.DROPOUT_P[0])
    ckpt = "./checkpoints/" + str(args.config) + "_" + str(args.index) + "_" + str(args.mode) + ".pt"     # predict.py:51
    from .sd_utils import synthetic_allowed
    if os.path.exists(ckpt):
        model.load_state_dict(torch.load(ckpt, map_location="cpu", weights_only=True))
        weights_source = ckpt
    elif synthetic_allowed():
        weights_source = "initial (seeded) parameters — SVG_ALLOW_SYNTHETIC_WEIGHTS"
    else:
        raise FileNotFoundError(ckpt)          # as torch.load(checkpoint_path) does in the reference (predict.py:52)
    if rank == 0:
        print("[sd-video-gen] weights: transformer=%s vae=%s unet=%s" % (weights_source, sd_utils.vae_source, sd_utils.unet_source))
    model.eval()
    F = config.FRAME_SIZE
    if args.dataset in ("synthetic-ball", "synthetic"):
        clips = bouncing_ball_clips(8, F, 5, seed=0)
    elif args.dataset in ("ball", "kitti"):                                                               # predict.py:56-58,111-114
        from .loaders import BouncingBall, Kitti
        if not args.folder:
            raise ValueError("dataset '%s' needs --folder (directory with a test/ stage of PNG frame folders)" % args.dataset)
        # unshuffled: clip c keeps its index (and its seed) whatever the world size — the reference shuffles (batch_size 1)
        ds = (BouncingBall if args.dataset == "ball" else Kitti)(num_frames=5, stride=1, dir=args.folder, stage="test", shuffle=False)
        items = [ds[i][1] for i in range(len(ds)) if len(ds.dataset[i]) == 5]
        if not items:
            raise ValueError("no 5-frame clips under %s" % os.path.join(args.folder, "test"))
        clips = torch.from_numpy(np.stack(items))
        if clips.shape[2] != F or clips.shape[3] != F:
            raise ValueError("frames are %dx%d but config %s has FRAME_SIZE %d" % (clips.shape[2], clips.shape[3], args.config, F))
    elif "ucf" in args.dataset:                                                                           # predict.py:60-109
        from .loaders import UCF101Frames, ucf_dirs, ucf_transform
        ucf_data_dir, ucf_label_dir = ucf_dirs(args.dataset)              # ValueError('Invalid dataset name') like :70
        if args.folder:                                                   # pre-extracted frames under <folder>/<the reference's layout>
            ucf_data_dir, ucf_label_dir = os.path.join(args.folder, ucf_data_dir), os.path.join(args.folder, ucf_label_dir)
        if rank == 0:
            print("Loading UCF dataset from", ucf_data_dir)
        train = args.mode == "train"
        ucf = UCF101Frames(ucf_data_dir, ucf_label_dir, frames_per_clip=5, train=train, transform=ucf_transform(F),
                           frame_rate=3 if train else None)               # :99-106
        if len(ucf) == 0:
            raise ValueError("no 5-frame clips under %s for the %s fold" % (ucf_data_dir, args.mode))
        n_clips = min(len(ucf), int(os.environ.get("SVG_MAX_CLIPS", "256")))
        clips = torch.from_numpy(np.stack([ucf[i][0] for i in range(n_clips)]))          # unshuffled: clip c keeps its seed
    elif args.folder:
        clips = _png_clips(args.folder, F)
    else:
        raise ValueError("Invalid dataset name")                                                          # predict.py:70
    n = clips.shape[0]
    lat, frames = run_sharded(clips, lambda c, seeds: sample_clips(
        model, sd_utils, c.to(device), args.pred_frames, denoise=bool(args.denoise), start_step=args.denoise_start_step,
        seeds=seeds, return_frames=True))                          # ends in the ONE collective of the path
    if rank == 0:
        print("all_latents shape: ", tuple(lat.shape))
        if args.save_output:
            n_in = 4
            for c in range(n):
                os.makedirs("outputs", exist_ok=True)
                folder_index = len(os.listdir("outputs"))
                save_frames(frames[c].cpu().numpy(), [False] * n_in + [True] * (frames.shape[1] - n_in),
                            os.path.join("outputs", str(folder_index)))
    return lat
