#!/usr/bin/env python3
"""Per-shape kernel timings through the op-level C ABI (hipEvent brackets inside the library).
usage: python tools/kbench.py [conv|gemm|attn|gn|xf|train] [--b B]"""
import argparse
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sd_video_gen_amd import _lib  # noqa: E402


def stream():
    return torch.cuda.current_stream().cuda_stream


def timeit(ctx, fam, fn, reps=5):
    fn()
    torch.cuda.synchronize()
    ctx.prof_reset()
    ctx.prof_enable(True)
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    rep = ctx.prof_report()
    ctx.prof_enable(False)
    r = rep[fam]
    return r["ms"] / reps, r["flops"] / reps, r["bytes"] / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="*", default=["conv", "gemm", "attn", "gn", "xf"])
    ap.add_argument("--b", type=int, default=8)
    a = ap.parse_args()
    ctx = _lib.Context(0)
    B = a.b
    bf = torch.bfloat16
    if "conv" in a.what:
        print("== conv3x3 (B=%d): H Cin Cout mode  ms  TFLOP/s" % B)
        shapes = [(64, 320, 320, 0), (64, 640, 320, 0), (64, 960, 320, 0), (32, 640, 640, 0), (32, 320, 640, 0), (32, 1280, 640, 0),
                  (32, 1920, 640, 0), (16, 1280, 1280, 0), (16, 640, 1280, 0), (16, 2560, 1280, 0), (8, 1280, 1280, 0), (8, 2560, 1280, 0),
                  (64, 320, 320, 1), (32, 640, 640, 1), (16, 1280, 1280, 1), (8, 1280, 1280, 3), (16, 1280, 1280, 3), (32, 640, 640, 3),
                  (64, 8, 320, 0), (64, 320, 4, 0),
                  (512, 8, 128, 0), (512, 128, 128, 0), (256, 128, 256, 0), (256, 256, 256, 0), (128, 256, 512, 0), (128, 512, 512, 0),
                  (64, 512, 512, 0), (256, 256, 256, 3), (128, 512, 512, 3), (512, 128, 128, 2), (512, 128, 4, 0)]
        for (H, Cin, Cout, mode) in shapes:
            b = B if H <= 64 else max(1, B // 4)
            x = torch.randn(b, H, H, Cin, device="cuda").to(bf)
            w = torch.randn(Cout, Cin, 3, 3, device="cuda") / math.sqrt(9 * Cin)
            Ho = H // 2 if mode in (1, 2) else (2 * H if mode == 3 else H)
            out = torch.empty(b, Ho, Ho, Cout, device="cuda", dtype=bf)
            ms, fl, _ = timeit(ctx, "conv3x3", lambda: ctx.check(ctx.lib.svg_op_conv3x3(
                ctx.h, x.data_ptr(), w.data_ptr(), None, out.data_ptr(), b, H, H, Cin, Cout, mode, stream()), "conv"))
            print("b=%d %4d %5d %5d %d  %8.3f ms  %7.1f TF" % (b, H, Cin, Cout, mode, ms, fl / ms / 1e9))
    if "convmx" in a.what:
        print("== MX fp8 conv3x3 vs fp16 (B=%d): H Cin Cout res  fp8 ms TF | fp16 ms TF" % B)
        f16 = torch.float16
        for (H, Cin, Cout) in [(64, 320, 320), (64, 640, 320), (64, 960, 320), (32, 640, 640), (32, 1280, 640), (16, 1280, 1280), (16, 2560, 1280)]:
            x = torch.randn(B, H, H, Cin, device="cuda").to(f16)
            w = torch.randn(Cout, Cin, 3, 3, device="cuda") / math.sqrt(9 * Cin)
            res = torch.randn(B, H, H, Cout, device="cuda").to(f16)
            out = torch.empty(B, H, H, Cout, device="cuda", dtype=f16)
            for r in (None, res):
                ms8, fl, _ = timeit(ctx, "conv3x3", lambda: ctx.check(ctx.lib.svg_op_conv3x3_mx_f16(
                    ctx.h, x.data_ptr(), w.data_ptr(), None, r.data_ptr() if r is not None else None, out.data_ptr(), None, None, B, H, H, Cin, Cout, 0, stream()), "convmx"))
                ms16 = float("nan")
                if r is None:
                    ms16, _, _ = timeit(ctx, "conv3x3", lambda: ctx.check(ctx.lib.svg_op_conv3x3_f16(
                        ctx.h, x.data_ptr(), w.data_ptr(), None, out.data_ptr(), B, H, H, Cin, Cout, 0, stream()), "conv"))
                print("b=%d %4d %5d %5d res%d  %8.3f ms %7.1f TF | %8.3f ms %7.1f TF" % (B, H, Cin, Cout, r is not None, ms8, fl / ms8 / 1e9, ms16, fl / ms16 / 1e9))
    if "gemm" in a.what:
        print("== gemm (B=%d): M N K act  ms  TFLOP/s" % B)
        shapes = []
        for hw, C in ((4096, 320), (1024, 640), (256, 1280), (64, 1280)):
            M = hw * B
            shapes += [(M, C, C, 0), (M, 2 * C, C, 0), (M, 8 * C, C, 3), (M, C, 4 * C, 0)]
        shapes += [(77 * B, 320, 768, 0), (77 * B, 1280, 768, 0), (B, 1280, 320, 1), (B, 31360, 1280, 0), (8192 * B, 512, 512, 0)]
        for (M, N, K, act) in shapes:
            A = torch.randn(M, K, device="cuda").to(bf)
            W = (torch.randn(N, K, device="cuda") / math.sqrt(K)).to(bf)
            out = torch.empty(M, N // 2 if act == 3 else N, device="cuda", dtype=bf)
            bias = torch.randn(N, device="cuda")                      # as in the network: bias everywhere, residual on the C -> C projections
            res = torch.randn(M, N, device="cuda").to(bf) if (act == 0 and N <= K) else None
            ms, fl, _ = timeit(ctx, "gemm", lambda: ctx.check(ctx.lib.svg_op_gemm(
                ctx.h, A.data_ptr(), W.data_ptr(), bias.data_ptr(), res.data_ptr() if res is not None else None, out.data_ptr(), M, N, K, act, 0, stream()), "gemm"))
            print("%7d %6d %6d %d  %8.3f ms  %7.1f TF" % (M, N, K, act, ms, fl / ms / 1e9))
    if "fp8" in a.what:
        print("== MX fp8 GEMM vs bf16 GEMM (B=%d): M N K  fp8 ms (TF, incl. nothing else)  quantise-A ms  bf16 ms (TF)" % B)
        for (hw, N, K) in ((256, 1280, 5120), (256, 1280, 1280), (256, 2560, 1280), (1024, 640, 2560), (1024, 1280, 640), (1024, 640, 640), (4096, 320, 1280), (4096, 640, 640)):
            M = hw * B
            A = torch.randn(M, K, device="cuda").to(bf)
            W = (torch.randn(N, K, device="cuda") / math.sqrt(K)).to(bf)
            out = torch.empty(M, N, device="cuda", dtype=bf)

            def run8():
                ctx.check(ctx.lib.svg_op_gemm_fp8(ctx.h, A.data_ptr(), W.data_ptr(), None, None, out.data_ptr(), M, N, K, 0, 0, stream()), "fp8")
            run8(); torch.cuda.synchronize()
            ctx.prof_reset(); ctx.prof_enable(True, detail=True)
            for _ in range(5):
                run8()
            torch.cuda.synchronize()
            rep = ctx.prof_report(); ctx.prof_enable(False)
            g8 = [v for k, v in rep.items() if k.startswith("@gemm|fp8")][0]
            qa = [v for k, v in rep.items() if k.startswith("@eltwise|quant_mx_rows%d_" % M)][0]
            ms16, fl, _ = timeit(ctx, "gemm", lambda: ctx.check(ctx.lib.svg_op_gemm(
                ctx.h, A.data_ptr(), W.data_ptr(), None, None, out.data_ptr(), M, N, K, 0, 0, stream()), "gemm"))
            ms8 = g8["ms"] / g8["calls"]
            print("%7d %6d %6d  %8.3f ms %7.1f TF   quant %7.3f ms   bf16 %8.3f ms %7.1f TF" % (M, N, K, ms8, fl / ms8 / 1e9, qa["ms"] / qa["calls"], ms16, fl / ms16 / 1e9))
    if "ff" in a.what:
        print("== fused GEGLU feed-forward C=320 (B=%d): M  ms  TFLOP/s" % B)
        C, Fh = 320, 1280
        for M in (4096 * B,):
            x = torch.randn(M, C, device="cuda").to(bf)
            gamma = torch.ones(C, device="cuda"); beta = torch.zeros(C, device="cuda")
            w1 = torch.randn(2 * Fh, C, device="cuda") / math.sqrt(C); b1 = torch.zeros(2 * Fh, device="cuda")
            w2 = torch.randn(C, Fh, device="cuda") / math.sqrt(Fh); b2 = torch.zeros(C, device="cuda")
            res = torch.randn(M, C, device="cuda").to(bf)
            out = torch.empty_like(x)
            ms, fl, _ = timeit(ctx, "gemm", lambda: ctx.check(ctx.lib.svg_op_ff_fused(
                ctx.h, x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), res.data_ptr(),
                out.data_ptr(), M, C, stream()), "ff"))
            print("%7d  %8.3f ms  %7.1f TF" % (M, ms, fl / ms / 1e9))
    if "attn" in a.what:
        print("== attention (B=%d): Sq Skv d  ms  TFLOP/s" % B)
        for (Sq, Skv, d) in ((4096, 4096, 40), (1024, 1024, 80), (256, 256, 160), (64, 64, 160), (4096, 77, 40), (1024, 77, 80), (256, 77, 160)):
            C = 8 * d
            Sp = (Skv + 7) // 8 * 8
            q = torch.randn(B, Sq, C, device="cuda").to(bf)
            k = torch.randn(B, Skv, C, device="cuda").to(bf)
            vt = torch.randn(B, C, Sp, device="cuda").to(bf)
            o = torch.empty_like(q)
            ms, fl, _ = timeit(ctx, "attention", lambda: ctx.check(ctx.lib.svg_op_attention(
                ctx.h, q.data_ptr(), k.data_ptr(), vt.data_ptr(), o.data_ptr(), B, 8, Sq, Skv, d, C, C, Sp, C, Sq * C, Skv * C, C * Sp, Sq * C,
                1 / math.sqrt(d), stream()), "attn"))
            print("%5d %5d %4d  %8.3f ms  %7.1f TF" % (Sq, Skv, d, ms, fl / ms / 1e9))
    if "gn" in a.what:
        print("== groupnorm+silu (B=%d): HW C  ms  GB/s (read+write once)" % B)
        for (HW, C) in ((4096, 320), (4096, 640), (4096, 960), (1024, 640), (1024, 1920), (256, 1280), (256, 2560), (64, 1280), (64, 2560),
                        (262144, 128), (65536, 256), (16384, 512)):
            b = B if HW <= 4096 else max(1, B // 4)
            x = torch.randn(b, HW, C, device="cuda").to(bf)
            g = torch.ones(C, device="cuda")
            o = torch.empty_like(x)
            ms, _, _ = timeit(ctx, "groupnorm", lambda: ctx.check(ctx.lib.svg_op_groupnorm(
                ctx.h, x.data_ptr(), g.data_ptr(), g.data_ptr(), o.data_ptr(), b, HW, C, 32, 1e-5, 1, stream()), "gn"))
            print("b=%d %7d %5d  %8.3f ms  %7.1f GB/s" % (b, HW, C, ms, 3 * x.numel() * 2 / ms / 1e6))
    if "xf" in a.what:
        print("== xf_gemm: M N K  ms  GB/s")
        for (M, N, K) in ((6, 2048, 2048), (6, 6144, 2048), (48, 6144, 2048), (48, 2048, 2048), (6, 2048, 256), (6, 256, 2048), (48, 4096, 2048),
                          (64, 2048, 2048), (64, 6144, 2048), (168, 2048, 2048), (168, 6144, 2048), (336, 2048, 2048), (336, 6144, 2048)):
            X = torch.randn(M, K, device="cuda")
            W = torch.randn(N, K, device="cuda")
            Y = torch.empty(M, N, device="cuda")
            ms, _, by = timeit(ctx, "xf_gemm", lambda: ctx.check(ctx.lib.svg_op_xf_gemm(
                ctx.h, X.data_ptr(), W.data_ptr(), None, Y.data_ptr(), M, N, K, 0, stream()), "xf"))
            print("%3d %5d %5d  %8.4f ms  %7.1f GB/s  %6.1f TFLOP/s (f32 MFMA peak 157)" % (M, N, K, ms, N * K * 4 / ms / 1e6, 2.0 * M * N * K / ms / 1e9))
    if "train" in a.what:
        # one optimisation step of the latent Transformer at the reference's training configuration (config 1_16_kitti_L1_64:
        # d=2048, 4+8 layers, batch 8 x (5+5 frames + SOS)), wall clock with torch events around 10 steps
        from sd_video_gen_amd import config as svg_config
        from sd_video_gen_amd.transformer import Transformer
        svg_config.set_args(["--dataset", "kitti", "--config", "1_16_kitti_L1_64"])
        cfgy = svg_config.parse_config_args()[0]
        torch.manual_seed(0)
        m = Transformer(dim_model=cfgy.DIM_MODEL[0], num_heads=cfgy.NUM_HEADS[0], num_encoder_layers=cfgy.NUM_ENCODER_LAYERS[0],
                        num_decoder_layers=cfgy.NUM_DECODER_LAYERS[0], dropout_p=cfgy.DROPOUT_P[0]).use_context(ctx)
        n_par = sum(p.numel() for p in m.parameters())
        for Bt in (8, 16, 32):
            T = cfgy.FRAMES_PER_CLIP[0] + cfgy.FRAMES_TO_PREDICT[0] + 1
            if Bt * T > 336 + 32:
                continue
            nb = torch.cat([2.0 * torch.ones(Bt, 1, 256), torch.randn(Bt, T - 1, 256)], dim=1).cuda()
            cfg = _lib.TrainCfg(frames_to_predict=cfgy.FRAMES_TO_PREDICT[0], feat_h=8, feat_w=8, w_mse=0.0, w_l1=1.0, w_gdl=0.0, gdl_alpha=1.0,
                                w_contrastive=0.0, temperature=0.07, dropout_p=cfgy.DROPOUT_P[0], seed=1)
            m.train()
            for _ in range(2):
                m.training_loss(cfg, nb); m.adam_step(cfgy.LR[0])
            torch.cuda.synchronize()
            import time
            for part in ("fwd+loss (eval)", "fwd+bwd", "adam", "step"):
                ts = []
                for i in range(30):
                    cfg.seed = 10 + i
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                    if part == "fwd+loss (eval)":
                        m.training_loss(cfg, nb, backward=False)
                    elif part == "fwd+bwd":
                        m.training_loss(cfg, nb)
                    elif part == "adam":
                        m.adam_step(cfgy.LR[0])
                    else:
                        m.training_loss(cfg, nb); m.adam_step(cfgy.LR[0])
                    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
                ts.sort()
                med, mean = ts[len(ts) // 2], sum(ts) / len(ts)
                # HBM bytes a step has to move: W read by forward and by dX, dW written, Adam reads p,g,m,v and writes p,m,v
                gb = {"fwd+loss (eval)": 1, "fwd+bwd": 3, "adam": 7, "step": 10}[part] * n_par * 4 / 1e9
                print("train B=%2d (%3d rows) %-16s median %7.3f ms (%5.0f GB/s of %5.2f GB algorithmic)   mean %7.3f ms, %d of 30 calls > 3x median"
                      % (Bt, Bt * T, part, med, gb / med * 1e3, gb, mean, sum(1 for t in ts if t > 3 * med)))

if __name__ == "__main__":
    main()
