#!/usr/bin/env python3
"""Minimal threaded scenarios: two contexts, two threads, the same short call sequence in each, repeated; outputs compared with a quiet reference."""
import os, sys, threading
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_streams_gpu as T
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
T._set_cfg()
(vsd, usd), arch = T._small_nets(), {"vae": T.VCFG, "unet": T.UCFG}
W = [T._worker(vsd, usd, arch) for _ in range(2)]
g = torch.Generator(device="cuda").manual_seed(1)
emb = W[0][1].encode_text([""]); emb4 = emb.repeat_interleave(2, 0)
z = [torch.randn(2, 4, 16, 16, device="cuda", generator=g) * 0.2 for _ in range(2)]
n = [torch.randn(2, 4, 16, 16, device="cuda", generator=g) for _ in range(2)]
img = [torch.randint(0, 256, (2, 64, 64, 3), dtype=torch.uint8, device="cuda", generator=g) for _ in range(2)]
e128 = [torch.randn(2, 4, 16, 16, device="cuda", generator=g) for _ in range(2)]
def seq(name, t):
    c = W[t][1].ctx
    if name == "ddim->decode":
        den = c.ddim_loop(z[t], emb4, num_steps=50, start_step=47, guidance=0.0, noise=n[t]); return [den, c.vae_decode(den, out_hw=(64, 64))]
    if name == "ddim,sync,decode":
        den = c.ddim_loop(z[t], emb4, num_steps=50, start_step=47, guidance=0.0, noise=n[t]); torch.cuda.current_stream().synchronize(); return [den, c.vae_decode(den, out_hw=(64, 64))]
    if name == "decode only":
        return [c.vae_decode(z[t], out_hw=(64, 64))]
    if name == "encode128->ddim->decode":
        r = c.vae_encode(img[t], H=128, W=128, eps=e128[t]); den = c.ddim_loop(r, emb4, num_steps=50, start_step=47, guidance=0.0, noise=n[t]); return [r, den, c.vae_decode(den, out_hw=(64, 64))]
    if name == "unet->decode":
        e = c.unet_forward(z[t], torch.tensor([500.0, 20.0], device="cuda"), emb[:1].repeat(2, 1, 1)); return [e, c.vae_decode(z[t], out_hw=(64, 64))]
for name in (os.environ.get("STRESS_ONLY", "").split(";") if os.environ.get("STRESS_ONLY") else ("decode only", "unet->decode", "ddim->decode", "ddim,sync,decode", "encode128->ddim->decode")):
    ref = []
    for t in range(2):
        with torch.cuda.stream(W[t][2]):
            ref.append([o.clone() for o in seq(name, t)]); W[t][2].synchronize()
    bad = [[0] * len(ref[0]) for _ in range(2)]
    for _ in range(N):
        outs = [None, None]
        def run(t):
            with torch.cuda.stream(W[t][2]):
                outs[t] = seq(name, t); W[t][2].synchronize()
        ths = [threading.Thread(target=run, args=(t,)) for t in range(2)]
        [x.start() for x in ths]; [x.join() for x in ths]
        for t in range(2):
            for k, (o, r) in enumerate(zip(outs[t], ref[t])):
                bad[t][k] += int(not torch.equal(o, r))
    print("%-26s outputs differing over %d threaded repetitions, per thread and output: %s" % (name, N, bad))
