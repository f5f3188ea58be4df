"""CPU: the I3D / FVD oracle (oracle/i3d_oracle.py) pinned against outputs of the LIVE reference modules evaluation/pytorch_i3d.py
and evaluation/fvd_2.py captured in tests/golden/i3d_fvd.pt (oracle/gen_golden_i3d.py)."""
import os
import sys

import torch

from conftest import rel_l2

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import i3d_oracle as IO  # noqa: E402

GOLD = torch.load(os.path.join(ROOT, "tests", "golden", "i3d_fvd.pt"), weights_only=False)


def test_i3d_shapes_and_param_count():
    sh = IO.i3d_shapes()
    assert sum(int(torch.tensor(v).prod()) for v in sh.values()) == 12_711_881 - 57          # minus the 57 num_batches_tracked scalars
    assert sh["Conv3d_1a_7x7.conv3d.weight"] == (64, 3, 7, 7, 7) and sh["logits.conv3d.weight"] == (400, 1024, 1, 1, 1)
    assert sh["Mixed_4f.b2b.conv3d.weight"] == (128, 32, 3, 3, 3)


def test_preprocess_matches_reference():
    x = IO.preprocess(GOLD["video"])
    assert x.shape == (2, 3, 16, 224, 224)
    assert rel_l2(x[:, :, ::5, ::37, ::41], GOLD["pre_slice"]) < 1e-6
    mean, std, lo, hi = GOLD["pre_stats"]
    assert abs(float(x.mean()) - mean) < 1e-6 and abs(float(x.std()) - std) < 1e-6 and float(x.min()) == lo and float(x.max()) == hi


def test_i3d_forward_matches_reference():
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    sd = IO.seeded_i3d_weights(GOLD["w_seed"])
    x = IO.preprocess(GOLD["video"])
    with torch.no_grad():
        f = IO.i3d_forward(sd, x, upto="Mixed_3c")
        assert rel_l2(f[:, ::17, :, ::5, ::6], GOLD["mixed3c_slice"]) < 2e-6
        logits = IO.i3d_forward(sd, x)
    assert logits.shape == (2, 400) and rel_l2(logits, GOLD["logits"]) < 5e-6


def test_frechet_distance_matches_reference():
    e1, e2, e3 = IO.fvd_test_embeddings(GOLD["emb_seed"])
    for a, b, key in ((e1, e2, "fd_12"), (e1, e3, "fd_13"), (e3, e3[:300], "fd_33")):
        got = float(IO.frechet_distance(a.clone(), b.clone()))
        assert abs(got - GOLD[key]) <= 2e-4 * abs(GOLD[key]), (key, got, GOLD[key])
    assert abs(float(IO.frechet_distance(e1.clone(), e1.clone()))) < 0.5 and abs(GOLD["fd_11"]) < 0.5          # identical sets: ~0 against traces of ~3000 (f32 SVD noise, 0.1 in the reference too)
