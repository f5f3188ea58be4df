#!/usr/bin/env python3
"""Launch one conv / gemm shape N times (for rocprofv3 --pmc runs).  usage: kone.py conv B H Cin Cout mode | gemm M N K"""
import math, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sd_video_gen_amd import _lib
ctx = _lib.Context(0)
s = torch.cuda.current_stream().cuda_stream
bf = torch.bfloat16
if sys.argv[1] == "conv":
    B, H, Cin, Cout, mode = map(int, sys.argv[2:7])
    x = torch.randn(B, H, H, Cin, device="cuda").to(bf)
    w = torch.randn(Cout, Cin, 3, 3, device="cuda") / math.sqrt(9 * Cin)
    Ho = H // 2 if mode in (1, 2) else (2 * H if mode == 3 else H)
    out = torch.empty(B, Ho, Ho, Cout, device="cuda", dtype=bf)
    for _ in range(6):
        ctx.check(ctx.lib.svg_op_conv3x3(ctx.h, x.data_ptr(), w.data_ptr(), None, out.data_ptr(), B, H, H, Cin, Cout, mode, s), "conv")
elif sys.argv[1] == "attn":
    B, Sq, Skv, d = map(int, sys.argv[2:6])
    C = 8 * d
    Sp = (Skv + 7) // 8 * 8
    q = torch.randn(B, Sq, C, device="cuda").to(bf)
    k = torch.randn(B, Skv, C, device="cuda").to(bf)
    vt = torch.randn(B, C, Sp, device="cuda").to(bf)
    o = torch.empty_like(q)
    for _ in range(6):
        ctx.check(ctx.lib.svg_op_attention(ctx.h, q.data_ptr(), k.data_ptr(), vt.data_ptr(), o.data_ptr(), B, 8, Sq, Skv, d, C, C, Sp, C, Sq * C, Skv * C, C * Sp, Sq * C, 1 / math.sqrt(d), s), "attn")
else:
    M, N, K = map(int, sys.argv[2:5])
    A = torch.randn(M, K, device="cuda").to(bf)
    W = (torch.randn(N, K, device="cuda") / math.sqrt(K)).to(bf)
    out = torch.empty(M, N, device="cuda", dtype=bf)
    for _ in range(6):
        ctx.check(ctx.lib.svg_op_gemm(ctx.h, A.data_ptr(), W.data_ptr(), None, None, out.data_ptr(), M, N, K, 0, 0, s), "gemm")
torch.cuda.synchronize()
