#!/usr/bin/env python3
"""Victim calls of the library with the LDS of every CU poisoned in between (tools/probe/lds_poison.hip): the poison kernel runs on the SAME
stream before every victim call (so the victim's first kernels inherit it) and, in a second pass, from another thread concurrently (so
kernels in the middle of a call inherit it too).  Any dependence on LDS a kernel never wrote shows as a large difference.
usage: python tools/lds_poison_test.py [full]"""
import ctypes, os, sys, threading
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_streams_gpu as T
from oracle import sd_oracle as SO
P = ctypes.CDLL(os.path.join(ROOT, "tools", "probe", "bin", "liblds_poison.so"))
P.lds_poison.argtypes = [ctypes.c_uint32, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
full = len(sys.argv) > 1
T._set_cfg()
if full:
    vsd, usd, arch = SO.seeded_weights(SO.vae_shapes(), 32), SO.seeded_weights(SO.unet_shapes(), 31), None
else:
    (vsd, usd), arch = T._small_nets(), {"vae": T.VCFG, "unet": T.UCFG}
m, sdu, st = T._worker(vsd, usd, arch)
c = sdu.ctx
g = torch.Generator(device="cuda").manual_seed(1)
emb = sdu.encode_text([""]); emb4 = emb.repeat_interleave(2, 0)
big = 64 if full else 16
x16 = torch.randn(2, 4, big, big, device="cuda", generator=g) * 0.2
x8 = torch.randn(2, 4, 8, 8, device="cuda", generator=g) * 0.2
img128 = torch.randint(0, 256, (2, big * 8, big * 8, 3), dtype=torch.uint8, device="cuda", generator=g)
img64 = torch.randint(0, 256, (10, 64, 64, 3), dtype=torch.uint8, device="cuda", generator=g)
e16 = torch.randn(2, 4, big, big, device="cuda", generator=g); e8 = torch.randn(10, 4, 8, 8, device="cuda", generator=g)
tt = torch.tensor([500.0, 20.0], device="cuda")
X = torch.randn(2, 6, 256, device="cuda", generator=g); pe = torch.zeros(2, dtype=torch.int32, device="cuda")
from sd_video_gen_amd.predict import predict
victims = {
    "vae_decode -> 64x64 u8": lambda: c.vae_decode(x16, out_hw=(64, 64)),
    "vae_decode float": lambda: c.vae_decode(x16, return_float=True)[1],
    "vae_decode 8x8 latent": lambda: c.vae_decode(x8),
    "vae_encode big": lambda: c.vae_encode(img128, eps=e16),
    "vae_encode 64x64 x10": lambda: c.vae_encode(img64, eps=e8),
    "unet_forward": lambda: c.unet_forward(x16, tt, emb[:1].repeat(2, 1, 1)),
    "ddim_loop 3 steps": lambda: c.ddim_loop(x16, emb4, num_steps=50, start_step=47, guidance=0.0, noise=e16),
    "transformer predict": lambda: predict(m, X, pe_row=pe),
}
PATTERNS = [0x7F7FFFFF, 0x7BFF7BFF, 0x7F800000, 0xFFFFFFFF, 0x3F800000]   # f32 max, fp16 max pair, +inf, NaN, 1.0
stop = False
def aggressor(sp):
    k = 0
    while not stop:
        P.lds_poison(PATTERNS[k % len(PATTERNS)], 64 * 1024, 1024, 64, ctypes.c_void_p(sp.cuda_stream)); k += 1
        sp.synchronize()
with torch.cuda.stream(st):
    for vn, vf in victims.items():
        ref = vf(); st.synchronize()
        bad1, w1 = 0, 0.0
        for pat in PATTERNS:
            P.lds_poison(pat, 160 * 1024, 512, 16, ctypes.c_void_p(st.cuda_stream))
            out = vf(); st.synchronize()
            if not torch.equal(out, ref):
                bad1 += 1; d = (out.float() - ref.float()).abs(); w1 = max(w1, float(d[torch.isfinite(d)].max()) if torch.isfinite(d).any() else float("inf")) if torch.isfinite(out.float()).all() else float("nan")
        stop = False
        sp = torch.cuda.Stream()
        th = threading.Thread(target=aggressor, args=(sp,)); th.start()
        bad2, w2 = 0, 0.0
        for _ in range(12):
            out = vf(); st.synchronize()
            if not torch.equal(out, ref):
                bad2 += 1; d = (out.float() - ref.float()).abs(); w2 = max(w2, float(d.max())) if torch.isfinite(d).all() else float("nan")
        stop = True; th.join()
        print("%-26s poison in front: %d of %d differ (max %.3e) | poison concurrently: %d of 12 differ (max %.3e)" % (vn, bad1, len(PATTERNS), w1, bad2, w2))
