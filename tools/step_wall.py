#!/usr/bin/env python3
"""GPU tool: wall time per DDIM step of one 28-clip group on ONE stream, hipGraph replay against direct launches, next to the
summed kernel time of the same step (hipEvent brackets of the library's profiler).  VERDICT r02 #5: single-stream step wall time
within 2 % of summed kernel time.   python tools/step_wall.py [clips] [dtype]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sd_video_gen_amd import _lib, sd_layout  # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 28
    dtype = sys.argv[2] if len(sys.argv) > 2 else "bf16"
    ctx = _lib.Context(0)
    c = sd_layout.SD_UNET
    ctx.configure(_lib.SVG_UNET, block_out=list(c["block_out"]), layers=2, heads=8, ctx_dim=768, groups=32, attn=list(c["attn"]), f16=int(dtype == "fp16"))
    ctx.load_state_dict(_lib.SVG_UNET, sd_layout.seeded_weights(sd_layout.unet_shapes(c), 2))
    ctx.finalize(_lib.SVG_UNET)
    z = torch.randn(N, 4, 64, 64, device="cuda") * 0.3
    e = torch.randn(1, 77, 768, device="cuda")
    emb = torch.cat([e, e]).repeat_interleave(N, 0) if False else torch.cat([e.repeat(N, 1, 1), e.repeat(N, 1, 1)])
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    out = {"clips": N, "dtype": dtype}
    steps = 20
    for mode in ("0", "1", "0", "1"):
        os.environ["SVG_DDIM_GRAPH"] = mode
        _lib.env_refresh()
        with torch.cuda.stream(side):
            ctx.ddim_loop(z, emb, num_steps=50, start_step=50 - steps, guidance=0.0, noise=torch.zeros_like(z))   # warm
            side.synchronize()
            t0 = time.perf_counter()
            ctx.ddim_loop(z, emb, num_steps=50, start_step=50 - steps, guidance=0.0, noise=torch.zeros_like(z))
            side.synchronize()
            dt = (time.perf_counter() - t0) / steps * 1e3
        out.setdefault("graph_ms_per_step" if mode == "1" else "direct_ms_per_step", []).append(round(dt, 3))
    # summed kernel time of a step (event brackets; graph off under the profiler)
    ctx.prof_reset()
    ctx.prof_enable(True)
    with torch.cuda.stream(side):
        ctx.ddim_loop(z, emb, num_steps=50, start_step=50 - steps, guidance=0.0, noise=torch.zeros_like(z))
        side.synchronize()
    rep = ctx.prof_report()
    ctx.prof_enable(False)
    outer = rep.pop("unet_step")
    out["bracketed_step_ms"] = round(outer["ms"] / outer["calls"], 3)
    out["sum_of_family_brackets_ms_per_step"] = round(sum(v["ms"] for v in rep.values()) / steps, 3)
    out["launches_per_step"] = sum(v["calls"] for v in rep.values()) // steps
    print(json.dumps(out))


if __name__ == "__main__":
    main()
