#!/usr/bin/env python3
"""Per-call-site time table of the SD side on one GPU: `clips` clips in lock step, a few DDIM steps of the full-size UNet
and the 512x512 VAE passes, hipEvent brackets per launch (svg_prof_enable(ctx, 2)).  Writes gpurun_out/shape_profile.json
and prints the signatures sorted by time with their TFLOP/s and algorithmic GB/s.

    python tools/shape_profile.py [--clips 28] [--steps 2] [--no-vae]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clips", type=int, default=28)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--no-vae", action="store_true")
    ap.add_argument("--fp8", action="store_true", help="MX fp8 3x3 convs (SDUtils(fp8=True))")
    ap.add_argument("--dtype", default="fp16", choices=["fp16", "bf16"])
    ap.add_argument("--grep", default="", help="only print call sites containing this substring")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "shape_profile.json"))
    a = ap.parse_args()
    import torch
    from sd_video_gen_amd import config as svg_config
    from sd_video_gen_amd.sd_utils import SDUtils
    svg_config.set_args(["--dataset", "synthetic-ball", "--config", "1_16_kitti_L1_64", "--denoise", "1"])
    sdu = SDUtils(weights="synthetic", verbose=False, fp8=a.fp8, dtype=a.dtype)
    ctx = sdu.ctx
    C = a.clips
    g = torch.Generator(device="cuda").manual_seed(0)
    z = torch.randn((C, 4, 64, 64), generator=g, device="cuda") * 0.8
    emb1 = sdu.encode_text([""])
    emb = torch.cat([emb1[:1].repeat(C, 1, 1), emb1[1:].repeat(C, 1, 1)])
    noise = torch.randn((C, 4, 64, 64), generator=g, device="cuda")
    S = 50 - a.steps

    def work():
        den = ctx.ddim_loop(z, emb, num_steps=50, start_step=S, guidance=0.0, noise=noise)
        if not a.no_vae:
            img = ctx.vae_decode(den)
            eps = torch.randn((C, 4, 64, 64), generator=g, device="cuda")
            ctx.vae_encode(img, eps=eps)
        return den
    work()                      # warm-up (arena growth, KV cache)
    torch.cuda.synchronize()
    ctx.prof_reset()
    ctx.prof_enable(True, detail=True)
    work()
    torch.cuda.synchronize()
    rep = ctx.prof_report()
    ctx.prof_enable(False)
    fam = {k: v for k, v in rep.items() if not k.startswith("@")}
    det = {k[1:]: v for k, v in rep.items() if k.startswith("@")}
    tot = sum(v["ms"] for v in fam.values())
    print("families (%d clips, %d DDIM steps%s): total %.2f ms" % (C, a.steps, "" if a.no_vae else " + VAE dec/enc @512", tot))
    for k, v in sorted(fam.items(), key=lambda kv: -kv[1]["ms"]):
        print("  %-10s %6d calls %9.3f ms  %5.1f%%  %7.1f TFLOP/s" % (k, v["calls"], v["ms"], 100 * v["ms"] / tot, v["flops"] / max(v["ms"], 1e-9) / 1e9))
    print("call sites:")
    for k, v in [kv for kv in sorted(det.items(), key=lambda kv: -kv[1]["ms"]) if a.grep in kv[0]][:70]:
        print("  %9.3f ms %5.1f%% %4d x %8.1f us  %7.1f TF/s %6.0f GB/s  %s" % (
            v["ms"], 100 * v["ms"] / tot, v["calls"], 1000 * v["ms"] / v["calls"], v["flops"] / max(v["ms"], 1e-9) / 1e9,
            v["bytes"] / max(v["ms"], 1e-9) / 1e6, k))
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump({"clips": C, "steps": a.steps, "vae": not a.no_vae, "families": fam, "sites": det}, f, indent=1)


if __name__ == "__main__":
    main()
