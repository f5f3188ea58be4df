"""GPU: the FVD evaluation in the library — I3D forward (implicit-GEMM 3-D convolutions on f32 MFMA), the uint8 -> I3D input
preprocessing and the Fréchet distance (Jacobi eigen-decomposition, f64) — against the golden outputs of the LIVE reference modules
(tests/golden/i3d_fvd.pt: evaluation/pytorch_i3d.py, evaluation/fvd_2.py) and the CPU oracle on the same seeded weights."""
import os
import sys

import pytest
import torch

from conftest import margin, rel_l2

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import i3d_oracle as IO  # noqa: E402
from sd_video_gen_amd import fvd  # noqa: E402

pytestmark = pytest.mark.gpu
GOLD = torch.load(os.path.join(ROOT, "tests", "golden", "i3d_fvd.pt"), weights_only=False)


@pytest.fixture(scope="module")
def i3d(ctx):
    assert fvd.i3d_shapes() == IO.i3d_shapes()
    m = fvd.I3D(IO.seeded_i3d_weights(GOLD["w_seed"]), ctx)
    assert m.n_params == 12_711_881 - 57
    return m


def test_preprocess_against_reference_golden(ctx):
    x = ctx.fvd_preprocess(GOLD["video"]).cpu()
    assert x.shape == (2, 3, 16, 224, 224)
    margin("fvd preprocess vs the reference's (bilinear resize, crop, [-1,1])", rel_l2(x[:, :, ::5, ::37, ::41], GOLD["pre_slice"]), 2e-6)
    assert rel_l2(x, IO.preprocess(GOLD["video"])) < 2e-6


def test_i3d_logits_against_reference_golden(ctx, i3d):
    logits = fvd.get_fvd_logits(torch.cat([GOLD["video"]] * 8), i3d).cpu()          # 16 clips: one get_logits batch
    assert logits.shape == (16, 400) and torch.isfinite(logits).all()
    margin("I3D logits (12.7 M parameters, 16 x 224 x 224 clips) vs the live reference module", rel_l2(logits[:2], GOLD["logits"]), 2e-5)
    assert torch.equal(logits[:2], logits[14:16])                                    # rows of a batch are independent
    # the float-input entry point on the oracle's preprocessing
    lg2 = i3d(IO.preprocess(GOLD["video"]).cuda()).cpu()
    assert rel_l2(lg2, GOLD["logits"]) < 2e-5


def test_reference_module_paths(ctx, i3d):
    """a user of the reference imports `evaluation.fvd_2` / `evaluation.pytorch_i3d` (predict_text.py:14-15 there): same names here,
    same call sequence (InceptionI3d(400, in_channels=3) -> load_state_dict -> eval; preprocess -> get_logits), same numbers"""
    from evaluation import fvd_2
    from evaluation.pytorch_i3d import InceptionI3d
    videos = torch.cat([GOLD["video"]] * 8)
    want = fvd.get_fvd_logits(videos, i3d).cpu()
    m = InceptionI3d(400, in_channels=3).to("cuda")
    with pytest.raises(RuntimeError):
        fvd_2.get_logits(m, fvd_2.preprocess(videos), "cuda")                      # no weights yet
    m.load_state_dict(IO.seeded_i3d_weights(GOLD["w_seed"]))
    m.eval()
    got = fvd_2.get_logits(m, fvd_2.preprocess(videos.numpy()), "cuda").cpu()      # the reference hands preprocess() a numpy array
    assert torch.equal(got, want)
    assert torch.equal(fvd_2.get_fvd_logits(videos.numpy(), m, "cuda").cpu(), want)
    e1, e2, _ = IO.fvd_test_embeddings(GOLD["emb_seed"])
    assert abs(fvd_2.frechet_distance(e1, e2) - GOLD["fd_12"]) <= 3e-3 * abs(GOLD["fd_12"])


def test_frechet_distance_against_reference_golden(ctx):
    e1, e2, e3 = IO.fvd_test_embeddings(GOLD["emb_seed"])
    for a, b, key in ((e1, e2, "fd_12"), (e1, e3, "fd_13"), (e3, e3[:300], "fd_33")):
        got = ctx.frechet_distance(a.cuda(), b.cuda())
        # the reference evaluates in f32 (SVD + a difference of traces of ~3000): its own value carries ~1e-3 of noise
        margin("frechet distance %s vs the reference's value (f32 SVD there, f64 Jacobi here)" % key, abs(got - GOLD[key]) / abs(GOLD[key]), 3e-3, unit="rel")
        ref64 = float(IO.frechet_distance(a.double(), b.double()))                 # the same formula (the pinned oracle) in float64
        margin("frechet distance %s vs the oracle in float64" % key, abs(got - ref64) / abs(ref64), 1e-6, unit="rel")
    assert abs(ctx.frechet_distance(e1.cuda(), e1.cuda())) < 1e-6 * 3000          # identical sets: exactly symmetric arithmetic -> ~0
    # against the float64 closed form on a small full-rank case
    g = torch.Generator().manual_seed(3)
    a = torch.randn(500, 16, generator=g)
    b = torch.randn(400, 16, generator=g) * 1.7 + 0.2
    ref = float(IO.frechet_distance(a.double(), b.double()))
    assert abs(ctx.frechet_distance(a.cuda(), b.cuda()) - ref) < 1e-5 * abs(ref)


def test_predict_text_main_end_to_end(tmp_path, monkeypatch):
    """`python -m prediction.predict_text` on a made-up UCF-101 tree of pre-extracted frames: class names -> MiniLM -> text-conditioned
    Transformer -> generated frames -> I3D logits of 16 real and 16 generated clips -> Fréchet distance (prediction/predict_text.py:76-321)."""
    import numpy as np
    from sd_video_gen_amd import predict_text as PT
    import test_boundary_gpu as TB                                # tiny SD networks in a diffusers-format directory
    from oracle import sd_oracle as SO
    TB.write_diffusers_dir(str(tmp_path / "sd"), SO.seeded_weights(SO.vae_shapes(TB.VCFG), 3), SO.seeded_weights(SO.unet_shapes(TB.UCFG), 4))
    monkeypatch.setenv("SVG_SD_WEIGHTS", str(tmp_path / "sd"))
    monkeypatch.setenv("SVG_ALLOW_SYNTHETIC_WEIGHTS", "1")        # Transformer / MiniLM / I3D: seeded parameters (no checkpoints offline)
    monkeypatch.setenv("SVG_FVD_CLIPS", "16")
    monkeypatch.chdir(tmp_path)
    root = tmp_path / "data" / "UCF-101" / "UCF-101-wallpushups"
    lab = tmp_path / "data" / "UCF101TrainTestSplits-RecognitionTask" / "ucfTrainTestlist"
    os.makedirs(lab)
    rng = np.random.default_rng(1)
    lines = []
    for cls, n in (("WallPushups", 2), ("PlayingGuitar", 1)):
        os.makedirs(root / cls)
        for v in range(n):
            name = "v_%s_g01_c%02d" % (cls, v + 1)
            base = rng.integers(0, 256, (1, 30, 40, 3))
            np.save(root / cls / (name + ".npy"), np.clip(base + rng.integers(-20, 20, (200, 30, 40, 3)), 0, 255).astype(np.uint8))
            lines.append("%s/%s.avi" % (cls, name))
    (lab / "testlist01.txt").write_text("\n".join(lines) + "\n")
    argv = ["--dataset", "ucf-wallpushups", "--config", "model_10_26", "--mode", "test", "--pred_frames", "16", "--save_output", "True"]         # 16 generated frames per clip: what I3D takes
    torch.manual_seed(11)                                         # the Transformer's initial parameters stand for a checkpoint
    val = PT.main(argv)
    # 200 frames at 25 -> 3 fps: 24 frames per video -> 9 clips of 16 each, 27 in all; 16 sampled: one group of real, one of generated clips
    assert val is not None and np.isfinite(val) and val > 0
    assert os.path.isdir(tmp_path / "outputs_pred" / "model_10_26_0_test")
    torch.manual_seed(11)
    assert PT.main(argv) == val                                   # seeded end to end
    with pytest.raises(ValueError, match="Invalid dataset name"):
        PT.main(["--dataset", "ucf-cooking", "--config", "model_10_26"])
