#!/usr/bin/env python3
"""GPU experiment: where the time of the short-K dense GEMMs goes.  For the 64x64-level shapes of a 28-clip UNet step, times the
launch under tile-width overrides (SVG_GEMM_BN) and compile-time-free ablations (SVG_GEMM_DBG: 1 no stores, 2 no MFMA, 3 no DMA),
each in its own process (the switches are read once).   python tools/gemm_exp.py            (parent)
                                                        python tools/gemm_exp.py child     (one configuration, env-driven)"""
import json
import math
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SHAPES = [(114688, 320, 320, 1), (114688, 640, 320, 0), (28672, 640, 640, 1), (28672, 1280, 640, 0), (7168, 1280, 1280, 1)]


def child():
    import torch
    from sd_video_gen_amd import _lib
    ctx = _lib.Context(0)
    bf = torch.bfloat16
    s = torch.cuda.current_stream().cuda_stream
    out = {}
    for (M, N, K, res) in SHAPES:
        A = torch.randn(M, K, device="cuda").to(bf)
        W = (torch.randn(N, K, device="cuda") / math.sqrt(K)).to(bf)
        bias = torch.randn(N, device="cuda")
        R = torch.randn(M, N, device="cuda").to(bf) if res else None
        C = torch.empty(M, N, device="cuda", dtype=bf)
        fn = lambda: ctx.check(ctx.lib.svg_op_gemm(ctx.h, A.data_ptr(), W.data_ptr(), bias.data_ptr(), R.data_ptr() if res else None, C.data_ptr(), M, N, K, 0, 0, s), "gemm")
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out["%dx%dx%d%s" % (M, N, K, "+res" if res else "")] = round(e0.elapsed_time(e1) / 20 * 1e3, 1)
    print(json.dumps(out))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        return child()
    rows = []
    for bn in ("", "64", "128", "160"):
        for dbg in ("0", "1", "2", "3"):
            env = dict(os.environ, SVG_GEMM_DBG=dbg)
            if bn:
                env["SVG_GEMM_BN"] = bn
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            rec = {"bn": bn or "auto", "dbg": dbg, "us": json.loads(line[-1]) if line else r.stderr[-300:]}
            rows.append(rec)
            print(json.dumps(rec), flush=True)
    with open(os.path.join(ROOT, "gpurun_out", "r03_gemm_exp.json"), "w") as f:
        json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
