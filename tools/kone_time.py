#!/usr/bin/env python3
"""Time one GEMM problem through the C ABI (hipEvent brackets inside the library), cold-ish and warm: 
usage: kone_time.py gemmres|gemm M N K [reps]   (gemmres: bias + residual, as the network's C -> C projections)"""
import math, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sd_video_gen_amd import _lib
ctx = _lib.Context(0)
s = torch.cuda.current_stream().cuda_stream
f16 = torch.float16
kind = sys.argv[1]
M, N, K = map(int, sys.argv[2:5])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 20
A = torch.randn(M, K, device="cuda").to(f16)
W = (torch.randn(N, K, device="cuda") / math.sqrt(K)).to(f16)
bias = torch.randn(N, device="cuda")
res = torch.randn(M, N, device="cuda").to(f16) if kind == "gemmres" else None
out = torch.empty(M, N, device="cuda", dtype=f16)
flush = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
def run():
    ctx.check(ctx.lib.svg_op_gemm_f16(ctx.h, A.data_ptr(), W.data_ptr(), bias.data_ptr(), res.data_ptr() if res is not None else None, out.data_ptr(), M, N, K, 0, 0, s), "gemm")
run(); torch.cuda.synchronize()
for mode in ("warm", "cold"):
    ctx.prof_reset(); ctx.prof_enable(True)
    for _ in range(reps):
        if mode == "cold":
            flush.zero_()          # 512 MiB through the caches: the operands come from HBM, as in the network
        run()
    torch.cuda.synchronize()
    r = ctx.prof_report()["gemm"]; ctx.prof_enable(False)
    ms = r["ms"] / reps
    by = (M * K + M * N * (2 if res is not None else 1) + N * K) * 2
    print("%s %d x %d x %d %s: %.4f ms  %.0f TF/s  %.0f GB/s of the operands" % (kind, M, N, K, mode, ms, 2.0 * M * N * K / ms / 1e9, by / ms / 1e6))
