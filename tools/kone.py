#!/usr/bin/env python3
"""Launch one conv / gemm / attention / fused-FF problem six times (for rocprofv3 --pmc runs).
usage: kone.py conv B H Cin Cout mode | gemm M N K | gemmres M N K (bias + residual, as in the network) | attn B Sq Skv d | ff M"""
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
elif sys.argv[1] == "ff":
    M = int(sys.argv[2]); C, Fh = 320, 1280
    x = torch.randn(M, C, device="cuda").to(bf)
    gamma = torch.ones(C, device="cuda"); beta = torch.zeros(C, device="cuda")
    w1 = torch.randn(2 * Fh, C, device="cuda") / math.sqrt(C); b1 = torch.zeros(2 * Fh, device="cuda")
    w2 = torch.randn(C, Fh, device="cuda") / math.sqrt(Fh); b2 = torch.zeros(C, device="cuda")
    res = torch.randn(M, C, device="cuda").to(bf)
    out = torch.empty_like(x)
    for _ in range(6):
        ctx.check(ctx.lib.svg_op_ff_fused(ctx.h, x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(),
                                          res.data_ptr(), out.data_ptr(), M, C, s), "ff")
elif sys.argv[1] == "gemmres":
    M, N, K = map(int, sys.argv[2:5])
    A = torch.randn(M, K, device="cuda").to(bf)
    W = (torch.randn(N, K, device="cuda") / math.sqrt(K)).to(bf)
    bias = torch.randn(N, device="cuda"); res = torch.randn(M, N, device="cuda").to(bf)
    out = torch.empty(M, N, device="cuda", dtype=bf)
    for _ in range(6):
        ctx.check(ctx.lib.svg_op_gemm(ctx.h, A.data_ptr(), W.data_ptr(), bias.data_ptr(), res.data_ptr(), out.data_ptr(), M, N, K, 0, 0, s), "gemm")
else:
    M, N, K = map(int, sys.argv[2:5])
    A = torch.randn(M, K, device="cuda").to(bf)
    W = (torch.randn(N, K, device="cuda") / math.sqrt(K)).to(bf)
    out = torch.empty(M, N, device="cuda", dtype=bf)
    for _ in range(6):
        ctx.check(ctx.lib.svg_op_gemm(ctx.h, A.data_ptr(), W.data_ptr(), None, None, out.data_ptr(), M, N, K, 0, 0, s), "gemm")
torch.cuda.synchronize()
