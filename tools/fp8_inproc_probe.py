#!/usr/bin/env python3
"""Why is the in-process fp8 pass of bench.py slower than `bench.py --dtype fp8` in its own process?  Times fp16 and fp8 worker pairs
in one process, in both orders, with and without the other pair alive."""
import os, sys, time, gc
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sd_video_gen_amd import _lib, config as svg_config
from sd_video_gen_amd.predict import sample_clips_streams, bouncing_ball_clips
from sd_video_gen_amd.sd_utils import SDUtils
from sd_video_gen_amd.transformer import Transformer

svg_config.set_args(["--dataset", "synthetic-ball", "--config", "1_16_kitti_L1_64", "--pred_frames", "1", "--denoise_start_step", "0", "--denoise", "True"])
cfg = svg_config.load_config("1_16_kitti_L1_64")
C = 56
clips = bouncing_ball_clips(C, cfg.FRAME_SIZE, 5, seed=0, device=torch.device("cuda", 0))
seeds = list(range(1234, 1234 + C))

def build(fp8, streams=None):
    ws = []
    for i in range(2):
        c = _lib.Context(0)
        torch.manual_seed(0)
        m = Transformer(num_tokens=0, dim_model=cfg.DIM_MODEL[0], num_heads=cfg.NUM_HEADS[0], num_encoder_layers=cfg.NUM_ENCODER_LAYERS[0],
                        num_decoder_layers=cfg.NUM_DECODER_LAYERS[0], dropout_p=cfg.DROPOUT_P[0]).eval().use_context(c)
        ws.append((m, SDUtils(weights="synthetic", seed=0, verbose=False, ctx=c, fp8=fp8, dtype="fp16"), streams[i] if streams else torch.cuda.Stream()))
    return ws

def timeit(ws, tag, n=2):
    emb = ws[0][1].encode_text([""])
    kw = dict(denoise=True, start_step=0, text_embeddings=emb, guidance_scale=0.0)
    sample_clips_streams(ws, clips, 1, seeds, **kw)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        sample_clips_streams(ws, clips, 1, seeds, **kw)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print("%-46s %.1f ms/step  %.2f frames/s   (device memory in use %.1f GB)" % (tag, dt * 1e3, C / dt, (torch.cuda.mem_get_info()[1] - torch.cuda.mem_get_info()[0]) / 1e9), flush=True)

if len(sys.argv) > 1 and sys.argv[1] == "fp8first":
    w8 = build(True)
    timeit(w8, "fp8 pair, built FIRST, alone")
    w16 = build(False)
    timeit(w16, "fp16 pair, built second")
    timeit(w8, "fp8 pair again")
    w8b = build(True)
    timeit(w8b, "a second fp8 pair, built third")
elif len(sys.argv) > 1 and sys.argv[1] == "prio":
    w16 = build(False)
    timeit(w16, "fp16 pair on new streams (s0, s1), default priority")
    lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
    print("priority range", lo, hi)
    for tag, pr in (("(high, normal)", (-1, 0)), ("(high, high)", (-1, -1)), ("(normal, normal) new", (0, 0))):
        st = [torch.cuda.Stream(priority=p) for p in pr]
        timeit([(w16[i][0], w16[i][1], st[i]) for i in range(2)], "fp16 pair on streams of priority " + tag)
    timeit(w16, "fp16 pair back on (s0, s1)")
elif len(sys.argv) > 1 and sys.argv[1] == "cross":
    w16 = build(False)
    timeit(w16, "fp16 pair on new streams (s0, s1)")
    w8 = build(True)
    timeit(w8, "fp8 pair on new streams (s2, s3)")
    sA = [w[2] for w in w16]; sB = [w[2] for w in w8]
    w8x = [(w8[i][0], w8[i][1], sA[i]) for i in range(2)]
    w16x = [(w16[i][0], w16[i][1], sB[i]) for i in range(2)]
    timeit(w8x, "the SAME fp8 pair on (s0, s1)")
    timeit(w16x, "the SAME fp16 pair on (s2, s3)")
    timeit(w8, "fp8 pair back on (s2, s3)")
    timeit(w16, "fp16 pair back on (s0, s1)")
elif len(sys.argv) > 1 and sys.argv[1] == "samestreams":
    w16 = build(False)
    timeit(w16, "fp16 pair, alone")
    w8 = build(True, streams=[w[2] for w in w16])
    timeit(w8, "fp8 pair on the fp16 pair's two streams")
    w8n = build(True)
    timeit(w8n, "fp8 pair on two new streams")
    w8m = build(True)
    timeit(w8m, "fp8 pair on two more new streams")
else:
    w16 = build(False)
    timeit(w16, "fp16 pair, alone")
    w8 = build(True)
    timeit(w8, "fp8 pair, fp16 pair alive")
    timeit(w16, "fp16 pair, fp8 pair alive")
    for m, s, _ in w16:
        s.ctx.close()
    del w16; gc.collect(); torch.cuda.empty_cache()
    timeit(w8, "fp8 pair, fp16 pair closed")
