"""Stand-alone check of the layer-walking launch (csrc/xf_walk.hip) against the per-GEMM kernels — prints, does not assert.
usage: python tools/xf_walk_check.py [d_model] [heads] [enc] [dec] [B] [d_lat]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sd_video_gen_amd import _lib, config as svg_config  # noqa: E402
from sd_video_gen_amd.transformer import Transformer  # noqa: E402

d, heads, enc, dec, B, d_lat = [int(a) for a in (sys.argv[1:] + ["256", "8", "1", "1", "1", "256"][len(sys.argv) - 1:])][:6]
svg_config.set_args(["--dataset", "ball", "--config", "model_10_26"])
torch.manual_seed(0)
m = Transformer(dim_model=d, num_heads=heads, num_encoder_layers=enc, num_decoder_layers=dec).eval()
X = torch.randn(B, 6, d_lat).cuda()
mask = m.get_tgt_mask(6).cuda()
pe0 = torch.zeros(B, dtype=torch.int32)


def run(walk, small=False):
    os.environ["SVG_XF_WALK"] = "1" if walk else "0"
    os.environ["SVG_XF_WALK_SMALL"] = "1" if small else "0"          # the small-row form (at most 8 rows), forced on / off
    _lib.env_refresh()
    out = m(X, X, mask, pe_row=pe0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        out = m(X, X, mask, pe_row=pe0)
    e1.record()
    torch.cuda.synchronize()
    return out.cpu(), e0.elapsed_time(e1) / 20


print("per-GEMM ...", flush=True)
old, t_old = run(False)
print("walk ...", flush=True)
new, t_new = run(True)
err = ((new - old).norm() / old.norm()).item()
line = "d=%d heads=%d enc=%d dec=%d B=%d d_lat=%d: rel-L2 walk vs per-GEMM %.3e | per-GEMM %.3f ms, walk %.3f ms" % (d, heads, enc, dec, B, d_lat, err, t_old, t_new)
if B * 6 <= 8 and d % 256 == 0 and d_lat % 256 == 0:
    sm, t_sm = run(True, small=True)
    line += ", small-row walk %.3f ms (rel-L2 vs per-GEMM %.3e)" % (t_sm, ((sm - old).norm() / old.norm()).item())
print(line, flush=True)
