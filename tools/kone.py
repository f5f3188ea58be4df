#!/usr/bin/env python3
"""Launch one conv / gemm / attention / fused-FF problem six times (for rocprofv3 --pmc runs).
usage: kone.py conv B H Cin Cout mode | convmx B H Cin Cout (MX-fp8 conv, fp16 in / out) | gemm M N K | gemmres M N K (bias + residual, as in the network) |
       attn B Sq Skv d | ff M | walk B (the latent Transformer forward of 1_16_kitti_L1_64 as one launch, B clips x 6 tokens)"""
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
elif sys.argv[1] == "convmx":
    B, H, Cin, Cout = map(int, sys.argv[2:6])
    f16 = torch.float16
    x = torch.randn(B, H, H, Cin, device="cuda").to(f16)
    w = torch.randn(Cout, Cin, 3, 3, device="cuda") / math.sqrt(9 * Cin)
    out = torch.empty(B, H, H, Cout, device="cuda", dtype=f16)
    for _ in range(6):
        ctx.check(ctx.lib.svg_op_conv3x3_mx_f16(ctx.h, x.data_ptr(), w.data_ptr(), None, None, out.data_ptr(), None, None, B, H, H, Cin, Cout, 0, s), "convmx")
elif sys.argv[1] == "walk":
    B = int(sys.argv[2])
    from sd_video_gen_amd import config as svg_config
    from sd_video_gen_amd.transformer import Transformer
    svg_config.set_args(["--dataset", "ball", "--config", "model_10_26"])
    torch.manual_seed(0)
    m = Transformer(dim_model=2048, num_heads=8, num_encoder_layers=4, num_decoder_layers=8).eval()
    X = torch.randn(B, 6, 256).cuda()
    mask = m.get_tgt_mask(6).cuda()
    pe0 = torch.zeros(B, dtype=torch.int32)
    os.environ["SVG_XF_WALK_ROWS"] = "176"
    _lib.env_refresh()
    for _ in range(6):
        m(X, X, mask, pe_row=pe0)
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
