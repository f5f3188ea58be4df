#!/usr/bin/env python3
"""svg_op_conv3x3_gn (conv whose epilogue emits the GroupNorm column sums + the GroupNorm that consumes them) from two threads on two contexts
at once: is the conv output / the normalised output bit-reproducible?"""
import ctypes, math, os, sys, threading
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sd_video_gen_amd import _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
ctxs = [_lib.Context(0), _lib.Context(0)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
f16 = torch.float16
for (B, H, W, Cin, Cout) in [(2, 128, 128, 64, 64), (2, 128, 128, 128, 64), (2, 64, 64, 128, 64), (2, 128, 128, 64, 128)]:
    g = torch.Generator(device="cuda").manual_seed(B + H + Cin)
    x = torch.randn(B, H, W, Cin, device="cuda", generator=g).to(f16)
    w = torch.randn(Cout, Cin, 3, 3, device="cuda", generator=g) / math.sqrt(9 * Cin)
    b = torch.randn(Cout, device="cuda", generator=g) * 0.5
    gamma = 1 + 0.2 * torch.randn(Cout, device="cuda", generator=g)
    beta = 0.3 * torch.randn(Cout, device="cuda", generator=g)
    def call(t):
        c = ctxs[t]
        conv = torch.empty(B, H, W, Cout, device="cuda", dtype=f16); out = torch.empty_like(conv)
        used = ctypes.c_int(-1)
        c.check(c.lib.svg_op_conv3x3_gn_f16(c.h, x.data_ptr(), w.data_ptr(), b.data_ptr(), gamma.data_ptr(), beta.data_ptr(), conv.data_ptr(), out.data_ptr(),
                                            B, H, W, Cin, Cout, 32, 1e-5, 1, ctypes.byref(used), torch.cuda.current_stream().cuda_stream), "conv_gn")
        return conv, out, used.value
    ref = []
    for t in range(2):
        with torch.cuda.stream(streams[t]):
            ref.append(call(t)); streams[t].synchronize()
    bad = [[0, 0], [0, 0]]; worst = 0.0
    for _ in range(N):
        outs = [None, None]
        def run(t):
            with torch.cuda.stream(streams[t]):
                outs[t] = call(t); streams[t].synchronize()
        ths = [threading.Thread(target=run, args=(t,)) for t in range(2)]
        [q.start() for q in ths]; [q.join() for q in ths]
        for t in range(2):
            for k in range(2):
                if not torch.equal(outs[t][k], ref[t][k]):
                    bad[t][k] += 1
                    d = (outs[t][k].float() - ref[t][k].float()).abs()
                    worst = max(worst, float(d.max()))
                    if bad[t][k] <= 3:
                        nz = (d > 0)
                        per_sample = nz.flatten(1).sum(1).tolist()
                        ch = nz.permute(3, 0, 1, 2).flatten(1).sum(1)          # per channel
                        groups = ch.reshape(32, -1).sum(1)
                        print("   thread %d output %d: %d elements differ; per sample %s; groups hit %s; channels hit %d of %d" % (
                            t, k, int(nz.sum()), per_sample, [int(i) for i in groups.nonzero().flatten()], int((ch > 0).sum()), ch.numel()))
    print("B%d %dx%d %d->%d (epilogue stats used: %d): [conv, gn] outputs differing per thread over %d paired calls: %s, max |diff| %.3e" % (B, H, W, Cin, Cout, ref[0][2], N, bad, worst))
