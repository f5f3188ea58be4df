#!/usr/bin/env python3
"""Which library call stops being bit-reproducible when ANOTHER context works on the GPU at the same time?  A victim context repeats one call
(VAE encode / decode, UNet forward, DDIM loop, Transformer forward) N times while an aggressor thread keeps a second context busy; every
repetition is compared bit for bit with the first (taken with the GPU otherwise idle).  usage: python tools/stress_ops.py [N] [full]"""
import os, sys, threading
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_streams_gpu as T
from oracle import sd_oracle as SO

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
full = len(sys.argv) > 2
T._set_cfg()
if full:
    vsd, usd, arch = SO.seeded_weights(SO.vae_shapes(), 32), SO.seeded_weights(SO.unet_shapes(), 31), None
else:
    (vsd, usd), arch = T._small_nets(), {"vae": T.VCFG, "unet": T.UCFG}
mV, sV, stV = T._worker(vsd, usd, arch)
mA, sA, stA = T._worker(vsd, usd, arch)
cV, cA = sV.ctx, sA.ctx
g = torch.Generator(device="cuda").manual_seed(1)
nb = 2 if full else 10
img = torch.randint(0, 256, (nb, 64, 64, 3), dtype=torch.uint8, device="cuda", generator=g)
eps = torch.randn(nb, 4, 8, 8, device="cuda", generator=g)
big = 512 if full else 128
img2 = torch.randint(0, 256, (2, big, big, 3), dtype=torch.uint8, device="cuda", generator=g)
eps2 = torch.randn(2, 4, big // 8, big // 8, device="cuda", generator=g)
z2 = torch.randn(2, 4, big // 8, big // 8, device="cuda", generator=g) * 0.2
emb = sV.encode_text([""])
emb4 = emb.repeat_interleave(2, 0)
tt = torch.tensor([500.0, 20.0], device="cuda")
X = torch.randn(2, 6, 256, device="cuda", generator=g)
pe = torch.zeros(2, dtype=torch.int32, device="cuda")
from sd_video_gen_amd.predict import predict
calls = {
    "vae_encode 64x64 x%d" % nb: lambda c, m: c.vae_encode(img, eps=eps),
    "vae_encode %dx%d x2" % (big, big): lambda c, m: c.vae_encode(img2, eps=eps2),
    "vae_decode %dx%d x2" % (big, big): lambda c, m: c.vae_decode(z2),
    "vae_decode -> 64x64": lambda c, m: c.vae_decode(z2, out_hw=(64, 64)),
    "unet_forward x2": lambda c, m: c.unet_forward(z2, tt, emb[:1].repeat(2, 1, 1)),
    "ddim_loop 3 steps x2": lambda c, m: c.ddim_loop(z2, emb4, num_steps=50, start_step=47, guidance=0.0, noise=eps2),
    "transformer predict 12 rows": lambda c, m: predict(m, X, pe_row=pe),
}
stop = False
def aggressor():
    with torch.cuda.stream(stA):
        while not stop:
            cA.vae_encode(img2, eps=eps2); cA.vae_decode(z2); cA.unet_forward(z2, tt, emb[:1].repeat(2, 1, 1))
            stA.synchronize()
for name, fn in calls.items():
    with torch.cuda.stream(stV):
        ref = fn(cV, mV); stV.synchronize()
        quiet = sum(int(not torch.equal(fn(cV, mV), ref)) for _ in range(N)); stV.synchronize()
    stop = False
    th = threading.Thread(target=aggressor); th.start()
    bad, worst = 0, 0.0
    with torch.cuda.stream(stV):
        for _ in range(N):
            out = fn(cV, mV); stV.synchronize()
            if not torch.equal(out, ref):
                bad += 1; worst = max(worst, float((out.float() - ref.float()).abs().max()))
    stop = True; th.join()
    print("%-32s alone: %d of %d differ | beside another context: %d of %d differ (max |diff| %.3e)" % (name, quiet, N, bad, N, worst))
