#!/usr/bin/env python3
"""Per-site sensitivity of the MX-fp8 conv placement: which resnet convs may run e4m3 x e4m3 under guidance 7.5?

For each placement ($SVG_FP8_SITES: bit i down_blocks.i, 4 mid, 5 + i up_blocks.i, 9 upsamplers, 10 conv1, 11 conv2) on the full-size
seeded UNet of the fixtures (contractive regime), against the SAME library in fp16 storage (1e-2 from the fp32 oracle under guidance, so
anything above a few 1e-2 here is the fp8 placement's own):
  call      rel-L2 of one batch-2 UNet call [uncond; cond], t = 500
  guided    rel-L2 of u + 7.5 (c - u) of that call (what the scheduler consumes: evaluation/predict_fvd2_denoise.py:227-229)
  loop50    rel-L2 of the latent after the 50-step DDIM loop at guidance 7.5 (free-running)
  ms/call   UNet call at batch 56 (the bench's 28 clips x [uncond; cond]) with the kernels' own eligibility rule, hipEvent-timed
usage: python tools/fp8_sites.py [out.json]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sd_video_gen_amd import _lib, sd_layout  # noqa: E402

ALL = 0xFFF
C1, C2 = 1 << 10, 1 << 11
BLK = dict(down64=1 << 0, down32=1 << 1, down16=1 << 2, mid=1 << 4, up8=1 << 5, up16=1 << 6, up32=1 << 7, up64=1 << 8, ups=1 << 9)
PLACEMENTS = [
    ("fp16 (no fp8)", None),
    ("all eligible convs (round 4)", ALL),
    ("all but up_blocks.3 (64^2 up path)", ALL & ~BLK["up64"]),
    ("all but up_blocks.3 and up_blocks.2", ALL & ~BLK["up64"] & ~BLK["up32"]),
    ("all but the 64^2 level (down.0, up.3)", ALL & ~BLK["up64"] & ~BLK["down64"]),
    ("down path + mid only", C1 | C2 | BLK["down64"] | BLK["down32"] | BLK["down16"] | BLK["mid"]),
    ("up path only", C1 | C2 | BLK["up8"] | BLK["up16"] | BLK["up32"] | BLK["up64"] | BLK["ups"]),
    ("conv1 only (all blocks)", ALL & ~C2),
    ("conv2 only (all blocks)", ALL & ~C1),
    ("conv1 only, not up_blocks.3", ALL & ~C2 & ~BLK["up64"]),
    ("16^2 + 32^2 levels only", C1 | C2 | BLK["down32"] | BLK["down16"] | BLK["up16"] | BLK["up32"] | BLK["ups"]),
    ("16^2 level only", C1 | C2 | BLK["down16"] | BLK["up16"]),
]


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r05_fp8_sites.json")
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    usd = dict(sd_layout.seeded_weights(sd_layout.unet_shapes(), 31))      # the fixtures' seeded UNet (seed 31) ...
    for k in ("conv_out.weight", "conv_out.bias"):                            # ... in the non-chaotic regime (conv_out x 0.1, as
        usd[k] = usd[k] * 0.1                                                 # tests/test_configs_gpu.py's contractive fixtures)
    c = sd_layout.SD_UNET
    ctx = _lib.Context(0)
    g = torch.Generator().manual_seed(11)
    z = (torch.randn(1, 4, 64, 64, generator=g) * 0.8)
    emb = torch.randn((2, 77, 768), generator=torch.Generator().manual_seed(124))     # [uncond; cond], distinct rows
    noise = torch.randn(1, 4, 64, 64, generator=g)
    x2 = torch.cat([z, z]).cuda()
    t2 = torch.tensor([500.0, 500.0]).cuda()
    B = 56
    xb = torch.randn(B, 4, 64, 64, generator=g).cuda()
    tb = torch.full((B,), 500.0).cuda()
    eb = emb.repeat(B // 2, 1, 1).cuda()
    rows, base = [], {}
    for name, mask in PLACEMENTS:
        for k in ("SVG_FP8_SITES", "SVG_HALO_MIN"):
            os.environ.pop(k, None)
        if mask is not None:
            os.environ["SVG_FP8_SITES"] = str(mask)
        os.environ["SVG_HALO_MIN"] = "1"          # the accuracy legs run batch 2: force the e4m3 kernel (same arithmetic at any batch)
        _lib.env_refresh()
        ctx.configure(_lib.SVG_UNET, block_out=list(c["block_out"]), layers=2, heads=8, ctx_dim=768, groups=32, attn=list(c["attn"]),
                      fp8=int(mask is not None), f16=1)
        ctx.load_state_dict(_lib.SVG_UNET, usd)
        ctx.finalize(_lib.SVG_UNET)
        ctx.prof_enable(True, detail=True); ctx.prof_reset()
        e = ctx.unet_forward(x2, t2, emb.cuda()).cpu()
        torch.cuda.synchronize()
        n8 = sum(v["calls"] for k, v in ctx.prof_report().items() if "conv_fp8" in k)
        ctx.prof_enable(False)
        loop = ctx.ddim_loop(z.cuda(), emb.cuda(), num_steps=50, start_step=0, guidance=7.5, noise=noise.cuda()).cpu()
        os.environ.pop("SVG_HALO_MIN")              # timing: the kernels' own eligibility rule, as the bench runs them
        _lib.env_refresh()
        for _ in range(2):
            ctx.unet_forward(xb, tb, eb)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(5):
            ctx.unet_forward(xb, tb, eb)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        guided = e[:1] + 7.5 * (e[1:] - e[:1])
        if mask is None:
            base = dict(e=e, guided=guided, loop=loop, ms=ms)
        row = dict(placement=name, mask=mask, fp8_conv_launches=n8, call=rel(e, base["e"]), guided=rel(guided, base["guided"]),
                   loop50=rel(loop, base["loop"]), ms_per_call_b56=ms, speedup=base["ms"] / ms)
        rows.append(row)
        print("%-44s mask %-6s fp8 convs %2d | call %.2e  guided %.2e  loop50 %.2e | %.2f ms / batch-56 call (x%.3f)"
              % (name, mask, n8, row["call"], row["guided"], row["loop50"], ms, row["speedup"]), flush=True)
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    with open(out_path, "w") as f:
        json.dump(dict(what="MX-fp8 conv placement vs the fp16 path of the same library; contractive seeded UNet, guidance 7.5", rows=rows), f, indent=1)


if __name__ == "__main__":
    main()
