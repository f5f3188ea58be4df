#!/usr/bin/env python3
"""What does a CU-masked stream (hipExtStreamCreateWithCUMask) do to one conv launch, and how do the mask's bits map to the chip?
One 3x3 conv (28 x 64 x 64, 320 -> 320, bf16) timed on: the unmasked stream; masks of 128 CUs chosen as the LOW bits, the EVEN bits, the bits
with (i mod 8) < 4 (XCDs 0-3 if the bits go round-robin over the XCDs) and (i div 32) < 4 (XCDs 0-3 if they are blocked per XCD); then two
complementary masks at once on two threads / contexts.
usage (GPU box): python tools/cumask_probe.py"""
import ctypes
import math
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sd_video_gen_amd import _lib  # noqa: E402

hip = ctypes.CDLL("libamdhip64.so")


def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)()
    for i in bits:
        words[i >> 5] |= 1 << (i & 31)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


def conv_problem(ctx, B=28, H=64, Cin=320, Cout=320):
    bf = torch.bfloat16
    x = torch.randn(B, H, H, Cin, device="cuda").to(bf)
    w = torch.randn(Cout, Cin, 3, 3, device="cuda") / math.sqrt(9 * Cin)
    out = torch.empty(B, H, H, Cout, device="cuda", dtype=bf)
    fl = 2.0 * B * H * H * Cin * 9 * Cout

    def run(st):
        ctx.check(ctx.lib.svg_op_conv3x3(ctx.h, x.data_ptr(), w.data_ptr(), None, out.data_ptr(), B, H, H, Cin, Cout, 0, st.cuda_stream), "conv")
    return run, fl


def wall(run, st, reps=40):
    with torch.cuda.stream(st):
        for _ in range(5):
            run(st)
        st.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            run(st)
        st.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    torch.cuda.set_device(0)
    ctx = _lib.Context(0)
    run, fl = conv_problem(ctx)
    masks = {
        "all 256": range(256),
        "low 128 bits": range(128),
        "even bits": range(0, 256, 2),
        "(i mod 8) < 4": [i for i in range(256) if i % 8 < 4],
        "(i div 32) < 4": [i for i in range(256) if (i // 32) < 4],
        "(i div 16) even": [i for i in range(256) if (i // 16) % 2 == 0],
    }
    print("one stream: ms per launch (TFLOP/s)")
    st0 = torch.cuda.Stream()
    ms = wall(run, st0)
    print("  %-18s %.3f ms  %7.1f TF" % ("torch stream", ms, fl / ms / 1e9))
    for name, bits in masks.items():
        ms = wall(run, masked_stream(bits))
        print("  %-18s %.3f ms  %7.1f TF" % (name, ms, fl / ms / 1e9))

    print("two contexts / threads at once: ms per launch of each, aggregate TFLOP/s")
    ctx2 = _lib.Context(0)
    run2, _ = conv_problem(ctx2)
    pairs = {
        "unmasked + unmasked": (None, None),
        "low 128 | high 128": (range(128), range(128, 256)),
        "even | odd": (range(0, 256, 2), range(1, 256, 2)),
        "(i mod 8) < 4 | >= 4": ([i for i in range(256) if i % 8 < 4], [i for i in range(256) if i % 8 >= 4]),
        "(i div 32) < 4 | >= 4": ([i for i in range(256) if i // 32 < 4], [i for i in range(256) if i // 32 >= 4]),
    }
    for name, (ma, mb) in pairs.items():
        sa = torch.cuda.Stream() if ma is None else masked_stream(ma)
        sb = torch.cuda.Stream() if mb is None else masked_stream(mb)
        res = [0.0, 0.0]

        def go(i, r, s):
            torch.cuda.set_device(0)
            res[i] = wall(r, s, reps=80)
        ts = [threading.Thread(target=go, args=(0, run, sa)), threading.Thread(target=go, args=(1, run2, sb))]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        print("  %-24s %.3f / %.3f ms   %7.1f TF" % (name, res[0], res[1], fl / res[0] / 1e9 + fl / res[1] / 1e9))


if __name__ == "__main__":
    main()
