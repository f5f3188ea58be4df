#!/usr/bin/env python3
"""Does a library call's result depend on what ran BEFORE it (stale LDS / workspace contents)?  One context, one stream: the victim call on
fixed inputs is preceded by a 'polluter' call on varying data; its outputs must not change.  usage: python tools/stale_state.py [extra env via SVG_*]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_streams_gpu as T

T._set_cfg()
(vsd, usd), arch = T._small_nets(), {"vae": T.VCFG, "unet": T.UCFG}
m, sdu, st = T._worker(vsd, usd, arch)
c = sdu.ctx
g = torch.Generator(device="cuda").manual_seed(1)
emb = sdu.encode_text([""])
emb4 = emb.repeat_interleave(2, 0)
x16 = torch.randn(2, 4, 16, 16, device="cuda", generator=g) * 0.2
x8 = torch.randn(2, 4, 8, 8, device="cuda", generator=g) * 0.2
img128 = torch.randint(0, 256, (2, 128, 128, 3), dtype=torch.uint8, device="cuda", generator=g)
img64 = torch.randint(0, 256, (10, 64, 64, 3), dtype=torch.uint8, device="cuda", generator=g)
e16 = torch.randn(2, 4, 16, 16, device="cuda", generator=g)
e8 = torch.randn(10, 4, 8, 8, device="cuda", generator=g)
tt = torch.tensor([500.0, 20.0], device="cuda")
victims = {
    "vae_decode 16x16 -> 64x64 u8": lambda: c.vae_decode(x16, out_hw=(64, 64)),
    "vae_decode 16x16 float": lambda: c.vae_decode(x16, return_float=True)[1],
    "vae_decode 8x8": lambda: c.vae_decode(x8),
    "vae_encode 128x128": lambda: c.vae_encode(img128, eps=e16),
    "vae_encode 64x64 x10": lambda: c.vae_encode(img64, eps=e8),
    "unet_forward": lambda: c.unet_forward(x16, tt, emb[:1].repeat(2, 1, 1)),
}
def polluters(k):
    s = 0.2 + 0.37 * k
    z = torch.randn(2, 4, 16, 16, device="cuda", generator=g) * s
    n = torch.randn(2, 4, 16, 16, device="cuda", generator=g)
    return {
        "ddim_loop": lambda: c.ddim_loop(z, emb4, num_steps=50, start_step=47, guidance=0.0, noise=n),
        "unet_forward": lambda: c.unet_forward(z, tt, emb[:1].repeat(2, 1, 1)),
        "vae_encode128": lambda: c.vae_encode((img128.float() * (0.3 + 0.1 * k)).clamp(0, 255).to(torch.uint8), eps=n),
    }
with torch.cuda.stream(st):
    for vn, vf in victims.items():
        ref = vf(); st.synchronize()
        for pn in ("ddim_loop", "unet_forward", "vae_encode128"):
            bad, worst = 0, 0.0
            for k in range(6):
                polluters(k)[pn](); out = vf(); st.synchronize()
                if not torch.equal(out, ref):
                    bad += 1; worst = max(worst, float((out.float() - ref.float()).abs().max()))
            print("%-30s after %-14s: %d of 6 differ (max |diff| %.3e)" % (vn, pn, bad, worst))
