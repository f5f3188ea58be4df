#!/usr/bin/env python3
"""As stress_streams.py, but every library call of both worker threads is recorded (a clone of its first output tensor) and a repetition that
differs from the first run reports the FIRST call, per group, whose output differs — and whether that call's INPUTS already differed."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_streams_gpu as T
from sd_video_gen_amd.predict import sample_clips_streams, bouncing_ball_clips
from sd_video_gen_amd import _lib

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
T._set_cfg()
vsd, usd = T._small_nets()
workers = [T._worker(vsd, usd, {"vae": T.VCFG, "unet": T.UCFG}) for _ in range(2)]
clips = bouncing_ball_clips(4, 64, 5, seed=9).cuda()
seeds = [21, 22, 23, 24]
emb = workers[0][1].encode_text([""])
kw = dict(denoise=True, start_step=47, text_embeddings=emb, res=128)
NAMES = ["vae_encode", "vae_decode", "ddim_loop", "transformer_forward", "resize_bilinear_f32"]
log = {}
def wrap(ctx, gi):
    for n in NAMES:
        orig = getattr(ctx, n)
        def f(*a, _o=orig, _n=n, **k):
            out = _o(*a, **k)
            first = out[0] if isinstance(out, tuple) else out
            ins = [x.detach().clone() for x in list(a) + list(k.values()) if isinstance(x, torch.Tensor)]
            log.setdefault(gi, []).append((_n, first.detach().clone(), ins))
            return out
        setattr(ctx, n, f)
for gi, w in enumerate(workers):
    wrap(w[1].ctx, gi)
def run():
    log.clear()
    out = sample_clips_streams(workers, clips, 2, seeds, **kw)
    torch.cuda.synchronize()
    return out, {g: list(v) for g, v in log.items()}
run()                      # (the first call also runs the planning pass: its call log is twice as long)
ref, ref_log = run()
bad = 0
for i in range(N):
    out, lg = run()
    if torch.equal(out, ref):
        continue
    bad += 1
    for g in lg:
        for j, ((n, o, ins), (rn, ro, rins)) in enumerate(zip(lg[g], ref_log[g])):
            if not torch.equal(o, ro):
                in_diff = [k for k, (x, y) in enumerate(zip(ins, rins)) if not torch.equal(x, y)]
                d = (o.float() - ro.float()).abs()
                print("run %d group %d: first differing call #%d %s (output shape %s, %d elements differ, max %.3e); inputs differing: %s" % (
                    i, g, j, n, tuple(o.shape), int((d > 0).sum()), float(d.max()), in_diff or "none"))
                for k in in_diff:
                    x, y = ins[k].float(), rins[k].float()
                    rows = (x != y).flatten(1).any(1).nonzero().flatten().tolist() if x.dim() > 1 else []
                    print("    input %d: shape %s, rows that differ %s; bad rows: mean %.3f std %.3f min %.3f max %.3f, finite %s" % (
                        k, tuple(x.shape), rows, float(x[rows].mean()), float(x[rows].std()), float(x[rows].min()), float(x[rows].max()), bool(torch.isfinite(x).all())))
                    # is the bad content some OTHER valid tensor of the reference run?
                    for g2 in ref_log:
                        for j2, (n2, o2, ins2) in enumerate(ref_log[g2]):
                            for k2, t2 in enumerate(ins2):
                                if t2.shape == x.shape and t2.dtype == ins[k].dtype and torch.equal(t2, ins[k]):
                                    print("    == reference run's group %d call #%d %s input %d" % (g2, j2, n2, k2))
                break
print("%d of %d repetitions differ" % (bad, N))
