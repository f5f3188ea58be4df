#!/usr/bin/env python3
"""Run-to-run determinism of predict.sample_clips_streams at reduced width (the setting of tests/test_streams_gpu.py): N repetitions of the
same threaded call, max |difference| against the first.  usage: python tools/stress_streams.py [N]   (try SVG_XF_WALK=0)"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_streams_gpu as T
from sd_video_gen_amd.predict import sample_clips_streams, bouncing_ball_clips

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10
FULL = os.environ.get("STRESS_FULL")          # full-size SD v1.4 networks, 512 x 512 round trip, STRESS_FULL clips per group
if FULL:
    from oracle import sd_oracle as SO
    T._set_cfg(True, "1_16_kitti_L1_64")
    usd, vsd = SO.seeded_weights(SO.unet_shapes(), 31), SO.seeded_weights(SO.vae_shapes(), 32)
    if os.environ.get("STRESS_FP8"):                       # MX-fp8 convs (what extras.fp8_same_box runs)
        os.environ["SVG_UNET_FP8"] = "1"
        os.environ["SVG_HALO_MIN"] = "1"
    workers = [T._worker(vsd, usd, None, seed=7, d_model=256, layers=(2, 2)) for _ in range(2)]
    nc = 2 * int(FULL)
    clips = bouncing_ball_clips(nc, 64, 5, seed=9).cuda()
    seeds = list(range(21, 21 + nc))
    emb = workers[0][1].encode_text([""])
    kw = dict(denoise=True, start_step=47, text_embeddings=emb, guidance_scale=float(os.environ.get("STRESS_GUIDANCE", "0")))
else:
    T._set_cfg()
    vsd, usd = T._small_nets()
    workers = [T._worker(vsd, usd, {"vae": T.VCFG, "unet": T.UCFG}) for _ in range(2)]
    clips = bouncing_ball_clips(4, 64, 5, seed=9).cuda()
    seeds = [21, 22, 23, 24]
    emb = workers[0][1].encode_text([""])
    kw = dict(denoise=True, start_step=47, text_embeddings=emb, res=128)
ref = sample_clips_streams(workers, clips, 2, seeds, **kw)
torch.cuda.synchronize()
bad = 0
for i in range(N):
    out = sample_clips_streams(workers, clips, 2, seeds, **kw)
    torch.cuda.synchronize()
    d = (out - ref).abs()
    if float(d.max()) != 0.0:
        bad += 1
        fr = d.amax(dim=(0, 2))
        print("run %d: max |diff| %.3e, %d elements differ, per frame position %s, per clip %s" % (i, float(d.max()), int((d > 0).sum()), [float("%.1e" % v) for v in fr],
              [float("%.1e" % v) for v in d.amax(dim=(1, 2))]))
print("%d of %d repetitions differ from the first" % (bad, N))
