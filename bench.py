#!/usr/bin/env python3
"""bench.py — generated frames/sec of the sampling hot path on N MI355X (one process per GPU).

  python bench.py --gpus N --steps K --warmup W

N>1 runs one process per GPU: either the caller launches them (`python -m torch.distributed.run --nproc-per-node N
bench.py --gpus N ...`, RANK/LOCAL_RANK/WORLD_SIZE in the env) or, when WORLD_SIZE is not set, this script spawns
torch.distributed.run itself BEFORE it touches the GPU and exits with the children's code (no process that has
initialised HIP is ever re-exec'ed).

`--workload cfg1|cfg3|cfg4` runs BASELINE.json's other configs (S = 25 x 8 frames at F=64; F=128 x 16 frames; text + guidance 7.5, MX-fp8)
with the same contract; the default is configs[2]:

Workload (BASELINE.json configs[2], the one the metric is quoted on): config 1_16_kitti_L1_64 (F=64 frames,
latent Transformer d=2048 4enc/8dec, 437.6 M params) with --denoise --denoise_start_step 0: per generated
frame one Transformer forward, VAE decode @64, VAE encode @512x512, 50 DDIM steps of the SD-v1.4 UNet at
64x64 latents (guidance_scale 0 as predict.py:169 -> batch-1 UNet is algorithmic), VAE decode @512x512,
VAE encode @64.  One "step" = `clips` independent synthetic bouncing-ball clips x `pred_frames` frames per GPU
(weak scaling: per-GPU work fixed), inputs resident in HBM, followed by the one all-gather of the clips.
Weights are seeded random-init tensors of the exact architectures (no checkpoints offline).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16 = 2.5e15        # dense bf16 / fp16 MFMA, MI355X_MICROARCH.md "Chip-level parameters"
PEAK_FP8 = 5.0e15         # dense MX-fp8 MFMA (same table)
PEAK_HBM = 8.0e12
UNET_FLOP = 803.27e9      # per sample per call
VAE_FLOP = {"enc": {64: 16.9e9, 128: 67.8e9, 512: 1116.7e9}, "dec": {64: 38.8e9, 128: 155.1e9, 512: 2514.5e9}}

# BASELINE.json's workloads.  The driver's default line is configs[2] (the one the metric is quoted on); the others are
# `--workload cfgN` lines kept under profiles/ (VERDICT r03 #5).  `dtype` None = the --dtype default (fp16, the reference's autocast).
WORKLOADS = {
    "cfg1": dict(idx=1, config="1_19_ball_complex_L1_64", pred_frames=8, start_step=25, guidance=0.0, text=False, dtype="bf16",
                 what="bouncing-ball 64x64, 8 predicted frames, --denoise --denoise_start_step 25 (25 DDIM steps per frame)"),
    "cfg2": dict(idx=2, config="1_16_kitti_L1_64", pred_frames=1, start_step=0, guidance=0.0, text=False, dtype=None,
                 what="64x64 latents -> 512x512 VAE passes, 50-step DDIM"),
    "cfg3": dict(idx=3, config="11_27_ucf_final", pred_frames=16, start_step=0, guidance=0.0, text=False, dtype=None,
                 what="F=128, 16 predicted frames, 50-step DDIM per frame, clips sharded over the GPUs"),
    "cfg4": dict(idx=4, config="11_27_ucf_text_final", pred_frames=16, start_step=0, guidance=7.5, text=True, dtype="fp8",
                 what="text-conditioned Transformer (d = 2432) + CLIP prompt per clip, guidance_scale 7.5 (batch-2 UNet calls), F=128, "
                      "16 predicted frames, 50-step DDIM per frame"),
}


def frame_flop(F, n_unet_calls, unet_batch_per_clip):
    """algorithmic FLOP per generated frame: the UNet calls, VAE enc + dec at 512x512 and at FxF, ~5 GFLOP of Transformer"""
    return n_unet_calls * unet_batch_per_clip * UNET_FLOP + VAE_FLOP["enc"][512] + VAE_FLOP["dec"][512] + VAE_FLOP["enc"][F] + VAE_FLOP["dec"][F] + 5e9


DTYPES = ["bf16", "fp16", "fp8"]


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=2)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--clips", type=int, default=56, help="clips sampled in lock step per GPU")
    p.add_argument("--streams", type=int, default=2, help="concurrent clip groups per GPU (own context + stream each)")
    p.add_argument("--workload", choices=sorted(WORKLOADS), default="cfg2",
                   help="which BASELINE.json config the step runs (default cfg2 = configs[2], the one the metric is quoted on); "
                        "sets --config / --pred_frames / --start_step / guidance / text conditioning / the config's named dtype")
    p.add_argument("--pred_frames", type=int, default=None)
    p.add_argument("--start_step", type=int, default=None)
    p.add_argument("--config", type=str, default=None)
    p.add_argument("--no-denoise", action="store_true")
    p.add_argument("--dtype", choices=DTYPES, default=None,
                   help="storage type of the SD networks (f32 accumulation): fp16 (default: the reference's autocast arithmetic, "
                        "sd_utils.py:246; UNet call 1.2e-3 from the fp32 oracle), bf16 (BASELINE configs[1] names it; 1e-2, ~3.6 %% faster), fp8 = BASELINE configs[4]: qualifying dense projections of the UNet in MX "
                        "block-scaled fp8 (the rest stays bf16)")
    p.add_argument("--train", action="store_true",
                   help="SURVEY 8(f1): optimisation steps of the latent Transformer (trainers/trainer.py:141-165) instead of the sampling loop; "
                        "one step = forward in train mode + criterion + backward + Adam on the config's own batch")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-roofline", action="store_true")
    p.add_argument("--no-fp8-extra", action="store_true",
                   help="skip extras.fp8_same_box (one step of the same workload with the MX-fp8 convs, rank 0, after the headline is computed)")
    args = p.parse_args()
    wl = WORKLOADS[args.workload]
    if args.dtype is None:
        # argparse does not check a DEFAULT against `choices`: validate the environment's value by hand (a typo such as
        # 'float16' used to fall through to bf16 and the line was labelled bf16)
        env = os.environ.get("SVG_BENCH_DTYPE")
        if env is not None and env not in DTYPES:
            p.error("SVG_BENCH_DTYPE=%r is not one of %s" % (env, ", ".join(DTYPES)))
        args.dtype = env or wl["dtype"] or "fp16"
    for k in ("config", "pred_frames", "start_step"):
        if getattr(args, k) is None:
            setattr(args, k, wl[k])
    args.guidance, args.text = wl["guidance"], wl["text"]
    return args


def host_cores():
    """CPU cores this process may actually use: affinity mask, cgroup quota, and the GPU box's per-GPU share (16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("SVG_CPU_THREADS", "16"))))


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(cfg_name, start_step, denoise=True, guidance=0.0, text=False):
    """The CPU restatement (oracle/, kind "port": the reference's own Python cannot travel to the GPU box) timed on this
    host's cores on a bounded sample of the same per-frame work: the latent Transformer forward (the text-conditioned one for
    configs[4]), ONE batch-1 UNet step (the DDIM loop repeats it (50 - start_step) x (2 under guidance) times: linear
    extrapolation, stated), and the VAE passes at 512x512 and at F x F timed directly."""
    import torch
    from oracle import sd_oracle as SO, transformer_oracle as TO
    from sd_video_gen_amd import sd_layout
    cores = host_cores()
    torch.set_num_threads(cores)
    t = {}

    def timed(name, fn, reps=1):
        fn() if reps > 1 else None
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        t[name] = (time.perf_counter() - t0) / reps
    with torch.no_grad():
        # latent Transformer at the configured size (weights drawn by torch.nn init, as the reference's)
        from sd_video_gen_amd import config as svg_config
        cfg = svg_config.load_config(cfg_name)          # (read only: the argv pinned by main() stays — later passes build SDUtils from it)
        torch.manual_seed(0)
        kw = dict(dim_model=cfg.DIM_MODEL[0], num_heads=cfg.NUM_HEADS[0], num_encoder_layers=cfg.NUM_ENCODER_LAYERS[0],
                  num_decoder_layers=cfg.NUM_DECODER_LAYERS[0], dropout_p=cfg.DROPOUT_P[0])
        txt = None
        if text:
            from sd_video_gen_amd.transformer_text import Transformer as TextTransformer
            m = TextTransformer(text_encoder="hash", **kw)
            txt = torch.nn.functional.normalize(torch.randn(1, 384), dim=1)
        else:
            from sd_video_gen_amd.transformer import Transformer
            m = Transformer(**kw)
        sd = {k: v for k, v in m.state_dict().items() if not k.startswith("sent_transformer.")}
        D = m.d_lat
        X = torch.randn(1, 6, D)
        timed("transformer", lambda: TO.predict(sd, X, cfg.NUM_HEADS[0], txt=txt), reps=5 if denoise else 40)
        del m, sd
        if not denoise:
            return {"value": 1.0 / t["transformer"], "unit": "frames/s", "cores": cores, "kind": "port", "cpu": cpu_model(),
                    "sample": "oracle (plain torch fp32) latent Transformer forward, one clip (6 tokens), %.4f s per frame on %d threads (40 forwards timed)"
                              % (t["transformer"], cores)}
        usd = sd_layout.seeded_weights(sd_layout.unet_shapes(), 2)
        x = torch.randn(1, 4, 64, 64)
        c = torch.randn(1, 77, 768)
        timed("unet_step_b1", lambda: SO.unet_forward(usd, x, 500, c), reps=2)
        del usd
        vsd = sd_layout.seeded_weights(sd_layout.vae_shapes(), 1)
        F = cfg.FRAME_SIZE
        img = torch.randint(0, 256, (1, 512, 512, 3), dtype=torch.uint8)
        timed("vae_enc_512", lambda: SO.encode_img(vsd, img))
        z = torch.randn(1, 4, 64, 64) * 0.2
        timed("vae_dec_512", lambda: SO.decode_img_latents(vsd, z))
        imgF = torch.randint(0, 256, (1, F, F, 3), dtype=torch.uint8)
        timed("vae_enc_F", lambda: SO.encode_img(vsd, imgF))
        zF = torch.randn(1, 4, F // 8, F // 8) * 0.2
        timed("vae_dec_F", lambda: SO.decode_img_latents(vsd, zF))
    n_unet = (50 - start_step) * (2 if guidance != 0.0 else 1)
    per_frame = t["transformer"] + n_unet * t["unet_step_b1"] + t["vae_enc_512"] + t["vae_dec_512"] + t["vae_enc_F"] + t["vae_dec_F"]
    return {"value": 1.0 / per_frame, "unit": "frames/s", "cores": cores, "kind": "port", "cpu": cpu_model(),
            "sample": "oracle (plain torch fp32) on %d host threads: 1 %sTransformer fwd %.3fs, 1 batch-1 UNet step %.2fs (mean of 2; x%d UNet sample-calls per frame "
                      "extrapolated: every step is the same call%s), VAE enc + dec at 512x512 %.2fs + %.2fs and at %dx%d %.2fs + %.2fs timed directly"
                      % (cores, "text-conditioned " if text else "", t["transformer"], t["unet_step_b1"], n_unet,
                         ", two samples per step under guidance" if guidance != 0.0 else "", t["vae_enc_512"], t["vae_dec_512"], F, F, t["vae_enc_F"], t["vae_dec_F"])}


def self_launch(args):
    """--gpus N>1 without a launcher: start `torch.distributed.run` as a CHILD (this parent has made no HIP / torch.cuda
    call), pass the flags through, return its exit code.  Rank 0 of the children prints the JSON line on the shared stdout."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL / cross-process device memory on this driver
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.run(cmd, env=env).returncode


MFMA_FAMILIES = {"conv3x3": "3x3 convolutions: conv_halo_kernel<BN,PP> + igemm_kernel<BN, conv modes> (UNet resnets / samplers, every VAE conv)",
                 "gemm": "dense GEMMs: igemm_kernel<BN,dense> + gemm_pp_kernel<BN> (+ split-K reduce): attention projections, GEGLU feed-forward, 1x1 convs, time MLP",
                 "attention": "attn_kernel<D,QB,NST,BC>: fused softmax(QK^T)V, self (4096/1024/256/64 keys) and cross (77 keys)"}
HBM_FAMILIES = {"groupnorm": "gn_stats / gn_apply / gn_small", "layernorm": "ln_stats (LayerNorm itself is folded into the consuming GEMM)",
                "eltwise": "layout / DDIM step / image pre-post kernels", "softmax": "VAE mid-block row softmax"}


def roofline_pass(args, sd_utils, step, denoise, C, model=None):
    """Instrumented pass on rank 0: ONE stream group's clips alone, hipEvent brackets recorded by the library on the launch
    stream around every launch of each kernel family (svg_prof_*).  `roofline` is the family that takes the most time;
    `roofline.by_family` lists every family against its own bound."""
    import torch
    from sd_video_gen_amd import _lib
    ctx = sd_utils.ctx
    ctx.prof_reset()
    ctx.prof_enable(True)
    step(gather=False)          # rank-0-only pass: no collective
    torch.cuda.synchronize()
    rep = ctx.prof_report()
    ctx.prof_enable(False)
    n_grp = max(1, C // args.streams)
    out = {}
    outer = rep.pop("unet_step", None)          # outer bracket around whole UNet calls: not a family (its time is inside the others)
    fam = {k: {"calls": v["calls"], "ms": round(v["ms"], 3), "tflops": round(v["flops"] / max(v["ms"], 1e-9) / 1e9, 1),
               "alg_gb_per_s": round(v["bytes"] / max(v["ms"], 1e-9) / 1e6, 1)} for k, v in rep.items()}
    tot_ms = sum(v["ms"] for v in rep.values())
    if not denoise:
        dom = rep["xf_gemm"]
        ach = dom["bytes"] / (dom["ms"] * 1e-3) / 1e9
        fwd_ms = rep["xf_gemm"]["ms"] + rep.get("xf_misc", {"ms": 0})["ms"]
        per_fwd = dom["calls"] <= 4        # the layer-walking launch: one kernel per forward (csrc/xf_walk.hip)
        out["roofline"] = {"bound": "hbm", "kernel": ("xf_walk_kernel<MT> (the whole forward in one launch: f32 weight stream of the latent Transformer, "
                                                      "v_mfma_f32_16x16x4_f32)") if per_fwd else
                           "xf_gemm_kernel<MT> (f32 weight stream of the latent Transformer, v_mfma_f32_16x16x4_f32)",
                           "achieved": ach, "peak": PEAK_HBM / 1e9, "unit": "GB/s", "frac": ach / (PEAK_HBM / 1e9), "traffic": None,
                           "launches": dom["calls"], "avg_launch_ms": dom["ms"] / dom["calls"],
                           "algorithmic_bytes_per_launch": dom["bytes"] / dom["calls"],
                           "whole_forward_gb_per_s": dom["bytes"] / (fwd_ms * 1e-3) / 1e9,
                           "rows_per_forward": n_grp * 6,
                           "f32_mfma_tflops": dom["flops"] / (dom["ms"] * 1e-3) / 1e12,
                           "note": "f32 MFMA peaks at 157 TFLOP/s: above ~50 rows (2*M/4 FLOP/B against 157e12/6.3e12) the stream is MFMA-bound, not HBM-bound"}
        if model is not None:
            # the HBM-bound regime proper: 8 clips x 6 tokens = 48 rows per forward (wall time of back-to-back forwards, torch events)
            n_par = sum(p.numel() for p in model.parameters())
            D = model.d_lat
            X = torch.randn(8, 6, D, device="cuda")
            mask = model.get_tgt_mask(6).cuda()
            pe0 = torch.zeros(8, dtype=torch.int32, device="cuda")
            with torch.no_grad():
                for _ in range(3):
                    model(X, X, mask, pe_row=pe0)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    model(X, X, mask, pe_row=pe0)
                e1.record()
                torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            out["roofline"]["small_batch_forward"] = {"rows": 48, "ms": ms, "gb_per_s": n_par * 4 / (ms * 1e-3) / 1e9,
                                                      "frac_of_6.3_TB_per_s": n_par * 4 / (ms * 1e-3) / 6.3e12}
        out["families"] = fam
        return out
    pmc = None
    for name in sorted(os.listdir(os.path.join(ROOT, "profiles")), reverse=True):
        if name.endswith("pmc_summary.json"):
            with open(os.path.join(ROOT, "profiles", name)) as f:
                cand = json.load(f)
            # HBM counters cannot be read from inside this process: they come from committed rocprofv3 --pmc passes of the same
            # step (tools/pmc_step.sh) and are used only when they were taken from THIS build of the kernels
            if cand.get("src_hash") == _lib.source_hash():
                pmc = (name, cand)
                break
    by_family = {}
    for k, v in rep.items():
        if k in MFMA_FAMILIES:
            ach = v["flops"] / (v["ms"] * 1e-3) / 1e12
            e = {"bound": "mfma", "kernel": MFMA_FAMILIES[k], "achieved": ach, "peak": PEAK_BF16 / 1e12, "unit": "TFLOP/s", "frac": ach / (PEAK_BF16 / 1e12),
                 "algorithmic_flop_per_launch": v["flops"] / v["calls"]}
        elif k in HBM_FAMILIES:
            ach = v["bytes"] / (v["ms"] * 1e-3) / 1e9
            e = {"bound": "hbm", "kernel": HBM_FAMILIES[k], "achieved": ach, "peak": PEAK_HBM / 1e9, "unit": "GB/s", "frac": ach / (PEAK_HBM / 1e9)}
        else:
            continue
        e.update(launches=v["calls"], ms=v["ms"], share_of_kernel_time=v["ms"] / tot_ms, avg_launch_ms=v["ms"] / v["calls"],
                 algorithmic_bytes_per_launch=v["bytes"] / v["calls"], traffic=None)
        if e["bound"] == "mfma":      # the other ceiling, for the short-K shapes whose operands bound them before the matrix pipe does
            e["algorithmic_gb_per_s"] = v["bytes"] / (v["ms"] * 1e-3) / 1e9
            e["frac_of_hbm_peak"] = e["algorithmic_gb_per_s"] / (PEAK_HBM / 1e9)
        if pmc and k in pmc[1]["families"]:
            # per launch in the sense of `achieved`: one bracketed call (a GEMM and its split-K reduce are two kernels, one launch here)
            pf = pmc[1]["families"][k]
            e["traffic"] = (pf["fetch_bytes"] + pf["write_bytes"]) / v["calls"]
            e["traffic_kernels"] = pf["launches"]
            e["traffic_source"] = "profiles/%s (%s)" % (pmc[0], pmc[1].get("note", ""))
        by_family[k] = e
    dom_name = max((k for k in by_family if by_family[k]["bound"] == "mfma"), key=lambda k: by_family[k]["ms"])
    per_clip = 2 if args.guidance != 0.0 else 1          # UNet samples per clip and call (the CFG pair is algorithmic at guidance 7.5)
    fflop = frame_flop(sd_utils.config.FRAME_SIZE, 50 - args.start_step, per_clip)
    out["roofline"] = dict(by_family[dom_name], family=dom_name, by_family=by_family,
                           instrumented_pass="one stream group (%d clips) run alone with hipEvent brackets around every launch" % n_grp,
                           algorithmic_flop_per_frame=fflop,
                           whole_frame_frac_of_mfma_peak=fflop * n_grp * args.pred_frames / (tot_ms * 1e-3) / PEAK_BF16)
    if outer and outer["calls"]:
        t_step = outer["ms"] / outer["calls"] * 1e-3
        nb = n_grp * per_clip
        out["roofline"]["unet_step"] = {"calls": outer["calls"], "ms_per_call": t_step * 1e3, "samples": nb,
                                        "achieved": UNET_FLOP * nb / t_step / 1e12, "peak": PEAK_BF16 / 1e12, "unit": "TFLOP/s",
                                        "frac": UNET_FLOP * nb / t_step / PEAK_BF16,
                                        "definition": "803.27e9 FLOP x samples / time of one UNet call + scheduler step (SURVEY 8d); "
                                                      "peak = the 16-bit MFMA peak also under --dtype fp8 (5.0e15 there would be: frac / 2)"}
    out["families"] = fam
    return out


def fp8_same_box(args, cfg, local_rank, build_model, clips, seeds, kw_s, cls_emb, fps_fp16, streams):
    """extras.fp8_same_box: the SAME workload, clips and process, with the UNet's resnet / upsampler 3x3 convs on MX-fp8
    operands (SDUtils(fp8=True): e4m3 x e4m3 + E8M0 block scales, fp16 storage elsewhere) — two warm-up and two timed steps after the
    headline has been computed.  The headline stays fp16 (the reference's autocast arithmetic); fp8 is narrower than the reference and is
    BASELINE configs[4]'s arithmetic only.  Parity of that arithmetic at the loop level: tests/test_configs_gpu.py [fp8] legs."""
    import torch
    from sd_video_gen_amd import _lib
    from sd_video_gen_amd.predict import sample_clips, sample_clips_streams
    from sd_video_gen_amd.sd_utils import SDUtils
    workers = []
    for i in range(args.streams):                       # on the headline's own streams (see GPU_MAX_HW_QUEUES in main)
        c8 = _lib.Context(local_rank)
        torch.manual_seed(0)
        workers.append((build_model(c8), SDUtils(weights="synthetic", seed=0, verbose=False, ctx=c8, fp8=True, dtype="fp16"), streams[i]))

    def step8():
        return sample_clips_streams(workers, clips, args.pred_frames, seeds, cls_list=cls_emb, **kw_s)
    n_warm, n_timed = 2, 2
    for _ in range(n_warm):
        step8()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n_timed):
        out = step8()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n_timed
    assert torch.isfinite(out).all()
    # one stream group alone under the library's hipEvent brackets: the UNet call of the fp8 build
    m0, sdu0, _ = workers[0]
    n1 = max(1, clips.shape[0] // args.streams)
    kw1 = dict(kw_s)
    emb = kw1.get("text_embeddings")
    n = clips.shape[0]
    if emb is not None and emb.shape[0] == 2 * n and n > 1:
        kw1["text_embeddings"] = torch.cat([emb[:n1], emb[n:n + n1]])
    sdu0.ctx.prof_reset(); sdu0.ctx.prof_enable(True)
    sample_clips(m0, sdu0, clips[:n1], 1, seeds=seeds[:n1], cls_list=(cls_emb[:n1] if cls_emb is not None else None), **kw1)
    torch.cuda.synchronize()
    rep = sdu0.ctx.prof_report(); sdu0.ctx.prof_enable(False)
    outer = rep.get("unet_step")
    fps8 = clips.shape[0] * args.pred_frames / dt
    rec = {"frames_per_s": fps8, "ms_per_step": dt * 1e3, "steps": n_timed, "warmup": n_warm, "vs_fp16_headline": fps8 / fps_fp16,
           "dtype": "fp8 (MX e4m3 x e4m3, E8M0 scale per 32 channels: the UNet's resnet + upsampler 3x3 convs) + " + sdu0.ctx.model_dtype(_lib.SVG_UNET) + " storage elsewhere",
           "note": "same process, clips and stream groups as the headline; NOT the headline (narrower than the reference's fp16 autocast)"}
    if outer and outer["calls"]:
        rec["unet_step"] = {"ms_per_call": outer["ms"] / outer["calls"], "samples": n1 * (2 if args.guidance != 0.0 else 1), "calls": outer["calls"]}
    if "conv3x3" in rep:
        rec["conv3x3_family"] = {"ms": rep["conv3x3"]["ms"], "tflops": rep["conv3x3"]["flops"] / max(rep["conv3x3"]["ms"], 1e-9) / 1e9}
    del workers
    return rec


def train_bench(args, rank, local_rank, world, dist):
    """Training-step line.  The reference's trainer is single-GPU (no DDP): with N ranks every rank trains its own replica
    (DESIGN.md section 6 'replicas only'), value = iterations of all ranks / max-over-ranks time."""
    import statistics
    import torch
    from sd_video_gen_amd import config as svg_config
    from sd_video_gen_amd import _lib
    from sd_video_gen_amd.transformer import Transformer
    svg_config.set_args(["--dataset", "synthetic-ball", "--config", args.config])
    cfg = svg_config.load_config(args.config)
    dev = torch.device("cuda", local_rank)
    first = lambda v: v[0] if isinstance(v, (list, tuple)) else v
    torch.manual_seed(0)
    model = Transformer(num_tokens=0, dim_model=first(cfg.DIM_MODEL), num_heads=first(cfg.NUM_HEADS), num_encoder_layers=first(cfg.NUM_ENCODER_LAYERS),
                        num_decoder_layers=first(cfg.NUM_DECODER_LAYERS), dropout_p=first(cfg.DROPOUT_P)).train()
    n_par = sum(p.numel() for p in model.parameters())
    B, F = first(cfg.BATCH_SIZE), first(cfg.FRAMES_TO_PREDICT)
    T = first(cfg.FRAMES_PER_CLIP) + F + 1
    feat = cfg.FRAME_SIZE // 8
    D = 4 * feat * feat
    g = torch.Generator().manual_seed(1 + rank)
    new_batch = torch.cat([2.0 * torch.ones(B, 1, D), 0.8 * torch.randn(B, T - 1, D, generator=g)], dim=1).to(dev)   # resident before timing
    w = dict(w_mse=float(bool(first(getattr(cfg, "USE_MSE", False)))), w_l1=float(bool(first(getattr(cfg, "USE_L1", False)))),
             w_gdl=float(bool(first(getattr(cfg, "USE_GDL", False)))) * float(first(getattr(cfg, "LAMBDA_GDL", 1))),
             alpha=float(first(getattr(cfg, "ALPHA", 1))),
             w_con=float(bool(first(getattr(cfg, "USE_CONTRASTIVE", False)))) * float(first(getattr(cfg, "LAMBDA_CONTRASTIVE", 0.0))))
    tc = _lib.TrainCfg(frames_to_predict=F, feat_h=feat, feat_w=feat, w_mse=w["w_mse"], w_l1=w["w_l1"], w_gdl=w["w_gdl"], gdl_alpha=w["alpha"],
                       w_contrastive=w["w_con"], temperature=0.07, dropout_p=float(first(cfg.DROPOUT_P)), seed=0)
    lr = float(first(cfg.LR))

    side = torch.cuda.Stream()        # a capturable stream: the library replays the step as one hipGraph there

    def step(i):
        tc.seed = i + 1
        with torch.cuda.stream(side):
            terms = model.training_loss(tc, new_batch)      # reads the loss back like loss.item() in the reference's loop
            model.adam_step(lr)
        return terms

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    steps, warm = max(args.steps, 20) if args.steps == 2 else args.steps, max(args.warmup, 3)
    for i in range(warm):
        step(i)
    sync()
    per = []
    t0 = time.perf_counter()
    for i in range(steps):
        t1 = time.perf_counter()
        terms = step(warm + i)
        torch.cuda.synchronize()
        per.append(time.perf_counter() - t1)
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    assert all(v == v for v in terms.values())
    med = statistics.median(per)
    step_bytes = 10.0 * n_par * 4          # W read by forward and by dX, dW written, Adam: p, g, m, v read and p, m, v written
    rec = {"metric": "training iterations/sec, latent Transformer (forward + criterion + backward + Adam)", "value": world * steps / dt,
           "unit": "iterations/s", "n_gpus": world, "steps": steps, "warmup": warm, "ms_per_step": dt / steps * 1e3, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "%s training step: %d parameters, batch %d x %d tokens (5+%d frames + SOS), D_lat %d, dropout %.2f, loss %s"
                                  % (args.config, n_par, B, T, F, D, float(first(cfg.DROPOUT_P)), {k: v for k, v in w.items() if v}),
                      "parallelism": "replicas only (the reference trainer is single-GPU)", "weights": "seeded random init"},
           "roofline": {"bound": "hbm", "kernel": "whole step: xf_gemm (forward) + xf_gemm_nn (dX) + xf_gemm_tn (dW, db) + adam_kernel",
                        "achieved": step_bytes / med / 1e9, "peak": PEAK_HBM / 1e9, "unit": "GB/s", "frac": step_bytes / med / PEAK_HBM, "traffic": None,
                        "algorithmic_bytes_per_step": step_bytes, "median_ms_per_step": med * 1e3, "mean_ms_per_step": dt / steps * 1e3,
                        "note": "median over per-step wall times (each step synchronises to read its loss); the mean includes this pool's "
                                "periodic ~75 ms stalls of many-launch sequences (profiles/README.md)"}}
    if rank == 0 and not args.no_cpu_baseline:
        from oracle import train_oracle as TR
        torch.set_num_threads(host_cores())
        sd = TR.leaf_state({k: v.detach().cpu() for k, v in model.state_dict().items()})
        opt = torch.optim.Adam(TR.params_of(sd), lr=lr)
        nb = new_batch.cpu()
        t1 = time.perf_counter()
        total, _ = TR.loss(sd, first(cfg.NUM_HEADS), nb, F, feat, w_mse=w["w_mse"], w_l1=w["w_l1"], w_gdl=w["w_gdl"], alpha=w["alpha"],
                           w_contrastive=w["w_con"])
        opt.zero_grad(); total.backward(); opt.step()
        t_cpu = time.perf_counter() - t1
        rec["cpu_baseline"] = {"value": 1.0 / t_cpu, "unit": "iterations/s", "cores": host_cores(), "kind": "port", "cpu": cpu_model(),
                               "sample": "one full step of the training oracle (torch autograd over the explicit-op forward + torch.optim.Adam), no dropout"}
    if rank == 0:
        print(json.dumps(rec))
    if world > 1:
        dist.destroy_process_group()
    return 0


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    # HIP multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4, assigned at a stream's first use): two stream
    # groups that land on ONE queue lose their overlap (measured: the same fp16 pair 19.0 frames/s on one pair of streams, 17.8 on another;
    # with 8 queues both 19.0 — profiles/r05_fp8_inproc_probe.txt).  The headline's two streams are the first two of the process and never
    # aliased; the later passes of this process (fp8_same_box) would.  Must be set before the HIP runtime initialises.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with --nproc-per-node equal to --gpus, or unset "
                         "WORLD_SIZE and let bench.py spawn the ranks)" % (args.gpus, world))
    assert torch.cuda.is_available(), "bench.py needs the GPU: the product path has no CPU fallback"
    # rehearsal knobs (single-GPU box): SVG_DEVICE_OVERRIDE pins every rank to one device, SVG_DIST_BACKEND=gloo
    # replaces RCCL (two ranks cannot share a GPU under RCCL); the driver's multi-GPU runs use neither.
    if os.environ.get("SVG_DEVICE_OVERRIDE") is not None:
        local_rank = int(os.environ["SVG_DEVICE_OVERRIDE"])
    backend = os.environ.get("SVG_DIST_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        assert dist.get_world_size() == args.gpus, "process group has %d ranks, --gpus %d" % (dist.get_world_size(), args.gpus)

    from sd_video_gen_amd import config as svg_config, sharding
    from sd_video_gen_amd import _lib
    from sd_video_gen_amd.predict import sample_clips, sample_clips_streams, bouncing_ball_clips
    from sd_video_gen_amd.sd_utils import SDUtils
    from sd_video_gen_amd.transformer import Transformer

    if args.train:
        return train_bench(args, rank, local_rank, world, dist)
    denoise = not args.no_denoise
    wl = WORKLOADS[args.workload]
    argv = ["--dataset", "synthetic-ball", "--config", args.config, "--pred_frames", str(args.pred_frames),
            "--denoise_start_step", str(args.start_step)] + (["--denoise", "True"] if denoise else [])
    svg_config.set_args(argv)
    cfg = svg_config.load_config(args.config)
    dev = torch.device("cuda", local_rank)
    torch.manual_seed(0)
    fp8 = args.dtype == "fp8"
    sd_dtype = "fp16" if args.dtype in ("fp16", "fp8") else "bf16"      # fp8: e4m3 operands where they pay, fp16 storage elsewhere
    if os.environ.get("SVG_FP8_BASE"):
        sd_dtype = os.environ["SVG_FP8_BASE"]
        if sd_dtype not in ("fp16", "bf16"):
            raise SystemExit("bench.py: SVG_FP8_BASE=%r is not one of fp16, bf16" % sd_dtype)
    guidance = args.guidance if denoise else 0.0

    def build_model(ctx=None):
        torch.manual_seed(0)
        kw = dict(num_tokens=0, dim_model=cfg.DIM_MODEL[0], num_heads=cfg.NUM_HEADS[0], num_encoder_layers=cfg.NUM_ENCODER_LAYERS[0],
                  num_decoder_layers=cfg.NUM_DECODER_LAYERS[0], dropout_p=cfg.DROPOUT_P[0])
        if args.text:
            from sd_video_gen_amd.transformer_text import Transformer as TextTransformer
            m = TextTransformer(st_weights="synthetic", **kw).eval()
        else:
            m = Transformer(**kw).eval()
        return m.use_context(ctx) if ctx is not None else m

    if rank == 0:
        print("[bench.py] seeded SYNTHETIC weights of the exact architectures (no checkpoints offline): outputs are not images", file=sys.stderr)
    sd_utils = SDUtils(weights="synthetic", seed=0, verbose=False, fp8=fp8, dtype=sd_dtype)     # stdout carries the ONE JSON line only
    model = build_model()
    F = cfg.FRAME_SIZE
    C = args.clips
    n_global = C * world
    a, b = sharding.shard_range(n_global, rank, world)
    clips = bouncing_ball_clips(b - a, F, 5, seed=a, device=dev)          # resident in HBM before timing
    seeds = sharding.clip_seeds(1234, a, b)
    emb, cls_emb = None, None
    if denoise and args.text:
        # configs[4]: one UCF-101 class per clip; the class name conditions the Transformer (MiniLM sentence embedding, once per clip)
        # and its prompt the UNet (CLIP, [uncond(C); cond(C)] rows; evaluation/predict_fvd2_denoise.py:203,227-229) — both constant
        # over a clip's frames, so computed before the timed region like the '' embedding of configs[2] (hoisted, exact)
        names = ["ApplyEyeMakeup", "Archery", "BabyCrawling", "Basketball", "BenchPress", "Biking", "Bowling", "WallPushups"]
        cls = [names[(a + i) % len(names)] for i in range(b - a)]
        emb = sd_utils.encode_text(["a person doing " + c for c in cls])
        cls_emb = model.encode_classes(cls).to(dev)
    elif denoise:
        emb = sd_utils.encode_text([""])

    workers = [(model, sd_utils, torch.cuda.Stream())]
    for _ in range(1, args.streams):
        c2 = _lib.Context(local_rank)
        torch.manual_seed(0)
        sdu2 = SDUtils(weights="synthetic", seed=0, verbose=False, ctx=c2, fp8=fp8, dtype=sd_dtype)
        workers.append((build_model(c2), sdu2, torch.cuda.Stream()))
    kw_s = dict(denoise=denoise, start_step=args.start_step, text_embeddings=emb, guidance_scale=guidance)

    def step(gather=True):
        if args.streams > 1 and gather:
            lat = sample_clips_streams(workers, clips, args.pred_frames, seeds, cls_list=cls_emb, **kw_s)
        else:
            # single-stream form; the rank-0 instrumented pass (gather=False) times one stream group's share of the clips
            n = clips.shape[0]
            n1 = n if gather else max(1, n // args.streams)
            kw1 = dict(kw_s)
            if emb is not None and emb.shape[0] == 2 * n and n > 1:
                kw1["text_embeddings"] = torch.cat([emb[:n1], emb[n:n + n1]])
            lat = sample_clips(model, sd_utils, clips[:n1], args.pred_frames, seeds=seeds[:n1],
                               cls_list=(cls_emb[:n1] if cls_emb is not None else None), **kw1)
        return sharding.gather_clips(lat, n_global) if gather else lat

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    assert out.shape[0] == n_global and torch.isfinite(out).all()
    frames = n_global * args.pred_frames * args.steps
    fps = frames / dt

    n_steps = 50 - args.start_step
    metric = ("generated frames/sec at %d DDIM denoise steps, 512x512" % n_steps) if denoise else \
        "generated frames/sec, 64x64 no-denoise (latent Transformer only)"
    # what actually ran, read back from the library (the requested string is not evidence)
    ran_dtype = sd_utils.ctx.model_dtype(_lib.SVG_UNET) if denoise else "f32"
    if denoise and fp8:
        ran_dtype = ("fp8 (MX e4m3: the 16 x 16 level's resnet 3x3 convs — the placement a guided DDIM loop keeps under 1e-1, $SVG_FP8_SITES_GUIDED) + "
                     if guidance != 0.0 else "fp8 (MX e4m3: the resnets' 3x3 convs) + ") + ran_dtype
    line = {"metric": metric, "value": fps, "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1000.0 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": ran_dtype, "data": "synthetic",
            "ranks_seen": (dist.get_world_size() if world > 1 else 1),
            "config": {"workload": ("configs[%d]: %s F=%d, %s; --denoise --denoise_start_step %d, guidance_scale %g; SD-v1.4 UNet at 64x64 latents, VAE enc/dec at 512x512"
                                    % (wl["idx"], args.config, F, wl["what"], args.start_step, guidance)
                                    + ("; MX block-scaled fp8 arithmetic (the dtype configs[4] names) on this workload" if fp8 and args.workload != "cfg4" else ""))
                       if denoise else "%s F=%d no --denoise (latent Transformer only)" % (args.config, F),
                       "clips_per_gpu": C, "streams_per_gpu": args.streams, "pred_frames": args.pred_frames, "global_clips": n_global, "parallelism": "clip-sharded dp%d" % world,
                       "weights": "seeded random init (SD v1.4 architecture, %s)" % args.config}}

    # The job is over for the other ranks: every collective of the run (gathers, the timing all-reduce) is behind this barrier.  Rank 0's
    # instrumented pass, fp8 pass and CPU baseline are single-process work (no collective) and may take minutes: nobody waits for them
    # inside an RCCL barrier under its watchdog.  `ranks_seen` was read while the group was alive.
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    # The headline above is complete.  What follows on rank 0 is optional single-process work; none of it may void the line
    # (BENCH_r05: an exception in extras.fp8_same_box, raised before the print, lost a finished measurement): each pass is guarded,
    # a failure is recorded under its own key, and the line is printed from `finally` whatever happens.
    def guarded(key, fn, into=None):
        try:
            res = fn()
            if into is None:
                line.update(res)
            else:
                line.setdefault(into, {})[key] = res
        except Exception as e:      # noqa: BLE001 — recorded, never fatal
            import traceback
            err = {"error": "%s: %s" % (type(e).__name__, e), "where": traceback.format_exc().strip().splitlines()[-3:]}
            if into is None:
                line[key] = err
            else:
                line.setdefault(into, {})[key] = err
            print("bench.py: optional pass %r failed: %s" % (key, err["error"]), file=sys.stderr)

    try:
        if rank == 0 and not args.no_roofline:
            guarded("roofline", lambda: roofline_pass(args, sd_utils, step, denoise, C, model))
        if rank == 0 and not args.no_cpu_baseline:
            guarded("cpu_baseline", lambda: {"cpu_baseline": cpu_baseline(args.config, args.start_step, denoise, guidance, args.text)})
        if rank == 0 and denoise and args.dtype == "fp16" and not args.no_fp8_extra:
            if os.environ.get("SVG_BENCH_FAIL_EXTRA"):      # test hook (tests/test_host_cpu.py): the failure path of an optional pass
                guarded("fp8_same_box", lambda: (_ for _ in ()).throw(RuntimeError("SVG_BENCH_FAIL_EXTRA")), into="extras")
            else:
                guarded("fp8_same_box", lambda: fp8_same_box(args, cfg, local_rank, build_model, clips, seeds, kw_s, cls_emb, fps,
                                                             [w[2] for w in workers]), into="extras")
    finally:
        if rank == 0:
            print(json.dumps(line), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
