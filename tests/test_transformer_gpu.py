"""GPU parity of the HIP latent-Transformer path (through the C ABI) against (a) the committed golden
outputs of the live reference and (b) the CPU oracle on seeded inputs.  f32 path (f32-input MFMA):
tolerance 2e-5 rel-L2 — north_star asks 1e-3 on predicted latents."""
import os
import sys

import pytest
import torch

from conftest import rel_l2

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import transformer_oracle as TO  # noqa: E402

pytestmark = pytest.mark.gpu
GOLD = os.path.join(ROOT, "tests", "golden")
TOL = 2e-5


def gold(name):
    return torch.load(os.path.join(GOLD, name), weights_only=False)


def build(cfg, kw, sd=None, seed=None):
    from sd_video_gen_amd import config as svg_config
    from sd_video_gen_amd.transformer import Transformer
    svg_config.set_args(["--dataset", "ball", "--config", cfg])
    if seed is not None:
        torch.manual_seed(seed)
    m = Transformer(**kw).eval()
    if sd is not None:
        m.load_state_dict(sd)
    return m


def test_tiny_against_reference_golden(ctx):
    g = gold("transformer_tiny.pt")
    m = build("model_10_26", dict(dim_model=32, num_heads=4, num_encoder_layers=1, num_decoder_layers=2), g["state_dict"])
    dev = lambda t: t.cuda()
    out6 = m(dev(g["X6"]), dev(g["X6"]), dev(m.get_tgt_mask(6)))
    assert out6.shape == (6, 1, 256)
    assert rel_l2(out6.cpu(), g["out6"]) < TOL
    x5 = dev(g["X5"])
    assert rel_l2(m(x5, x5, dev(m.get_tgt_mask(5))).cpu(), g["out5"]) < TOL
    xb = dev(g["Xb"])
    assert rel_l2(m(xb, xb, dev(m.get_tgt_mask(5))).cpu(), g["outb"]) < TOL          # PE(b) per batch row
    assert rel_l2(m(x5, dev(g["X6"]), None).cpu(), g["out_nomask"]) < TOL           # Ts != Tt, no mask
    # pe_row override: every row PE(0) == running the rows one at a time
    rows = torch.cat([m(xb[i:i + 1], xb[i:i + 1], dev(m.get_tgt_mask(5))) for i in range(3)], dim=1)
    batched = m(xb, xb, dev(m.get_tgt_mask(5)), pe_row=torch.zeros(3, dtype=torch.int32))
    assert rel_l2(batched.cpu(), rows.cpu()) < 1e-6


def test_key_padding_masks_against_reference_golden(ctx):
    """src_pad_mask / tgt_pad_mask (models/transformer.py:64) through svg_transformer_forward_padded vs the live reference's outputs"""
    g = gold("transformer_pad.pt")
    m = build("model_10_26", dict(dim_model=32, num_heads=4, num_encoder_layers=1, num_decoder_layers=2), gold("transformer_tiny.pt")["state_dict"])
    src, tgt, m5 = g["src"].cuda(), g["tgt"].cuda(), m.get_tgt_mask(5).cuda()
    sp = m.create_pad_mask(g["src_ids"], 0)
    tp = m.create_pad_mask(g["tgt_ids"], 0)
    assert rel_l2(m(src, tgt, m5, sp, tp).cpu(), g["out_both"]) < TOL
    assert rel_l2(m(src, tgt, m5, sp.cuda(), None).cpu(), g["out_src"]) < TOL
    assert rel_l2(m(src, tgt, None, None, tp).cpu(), g["out_tgt"]) < TOL
    assert rel_l2(m(src, tgt, m5, g["float_src_pad"], None).cpu(), g["out_float_src"]) < TOL          # float mask: added to the scores
    assert rel_l2(m(src, tgt, m5).cpu(), g["out_both"]) > 1e-2
    with pytest.raises(ValueError):
        m(src, tgt, m5, sp[:, :3], None)
    # a batch larger than one weight-stream chunk (336 rows / 6 tokens = 56 batch rows): masks follow their rows through the chunks
    B = 60
    gS = torch.Generator().manual_seed(5)
    S = torch.randn(B, 6, 256, generator=gS)
    pad = torch.rand(B, 6, generator=gS) > 0.7
    pad[:, 0] = False
    pe0 = torch.zeros(B, dtype=torch.int32)
    out = m(S.cuda(), S.cuda(), m.get_tgt_mask(6).cuda(), pad, pad, pe_row=pe0).cpu()
    sd = {k: v.cpu() for k, v in m.state_dict().items()}
    for b in (0, 29, 57, 59):
        ref = TO.forward(sd, S[b:b + 1], S[b:b + 1], 4, TO.get_tgt_mask(6), src_pad_mask=pad[b:b + 1], tgt_pad_mask=pad[b:b + 1])
        assert rel_l2(out[:, b:b + 1], ref) < TOL


@pytest.mark.parametrize("cfg", ["config_test", "1_16_kitti_L1_64", "11_27_ucf_final"])
def test_full_size_against_reference_golden(ctx, cfg):
    spot = gold("transformer_spot.pt")[cfg]
    m = build(cfg, spot["kw"], seed=spot["seed"])
    from sd_video_gen_amd.predict import predict
    pred = predict(m, spot["X"].cuda())
    assert pred.shape == (spot["d_lat"],)
    assert rel_l2(pred.cpu(), spot["pred"]) < TOL
    assert m.n_params == spot["n_params"]


def test_clip_batched_matches_oracle(ctx):
    """B=11 clips x T=6 (> 64 rows: chunked inside the library), every clip on PE(0)."""
    torch.manual_seed(5)
    m = build("model_10_26", dict(dim_model=64, num_heads=4, num_encoder_layers=2, num_decoder_layers=2))
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    X = torch.randn(11, 6, 256)
    out = m(X.cuda(), X.cuda(), m.get_tgt_mask(6).cuda(), pe_row=torch.zeros(11, dtype=torch.int32)).cpu()
    for b in range(11):
        ref = TO.forward(sd, X[b:b + 1], X[b:b + 1], 4, TO.get_tgt_mask(6))
        assert rel_l2(out[:, b:b + 1], ref) < TOL
    # reference quirk path (PE row = batch row) at B=11
    out_q = m(X.cuda(), X.cuda(), m.get_tgt_mask(6).cuda()).cpu()
    assert rel_l2(out_q, TO.forward(sd, X, X, 4, TO.get_tgt_mask(6))) < TOL


def test_rollout_matches_reference_trace(ctx):
    g = gold("loop_trace.pt")
    t = gold("transformer_tiny.pt")
    m = build("model_10_26", dict(dim_model=32, num_heads=4, num_encoder_layers=1, num_decoder_layers=2), t["state_dict"])
    from sd_video_gen_amd.predict import rollout_latents
    all_latents, trace = rollout_latents(m, g["new_batch"].cuda(), pred_frames=4)
    assert trace == g["trace"]
    assert rel_l2(all_latents.cpu(), g["all_latents"]) < 5e-5


def test_errors(ctx):
    m = build("model_10_26", dict(dim_model=32, num_heads=4, num_encoder_layers=1, num_decoder_layers=1))
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 5, 256), torch.zeros(1, 5, 256))        # CPU tensors: no fallback
    with pytest.raises((RuntimeError, ValueError)):
        m(torch.zeros(1, 40, 256).cuda(), torch.zeros(1, 40, 256).cuda())   # T > 16 unsupported


def test_text_conditioned_variant(ctx):
    """a14 / BASELINE config 5 shape family: d = DIM_MODEL + 384 (head dim not a power of two), clip-batched."""
    from sd_video_gen_amd import config as svg_config
    from sd_video_gen_amd.transformer_text import Transformer as TextTransformer, predict as predict_text
    svg_config.set_args(["--dataset", "ucf", "--config", "model_10_26"])
    torch.manual_seed(4)
    m = TextTransformer(dim_model=128, num_heads=8, num_encoder_layers=2, num_decoder_layers=2, st_weights="synthetic").eval()
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    X = torch.randn(3, 6, 256)
    names = ["WallPushups", "PlayingGuitar", "WallPushups"]
    txt = m.encode_classes(names).cpu()
    out = m(X.cuda(), names, X.cuda(), m.get_tgt_mask(6).cuda()).cpu()
    assert out.shape == (6, 3, 256)
    assert rel_l2(out, TO.forward(sd, X, X, 8, TO.get_tgt_mask(6), txt=txt)) < TOL
    p = predict_text(m, X[:1].cuda(), names[:1]).cpu()
    assert rel_l2(p, TO.predict(sd, X[:1], 8, txt=txt[:1])) < TOL
    with pytest.raises((RuntimeError, ValueError)):
        m._ctx.transformer_forward(X.cuda(), X.cuda(), None, None)       # text model (in ITS context) called without a text embedding


# ---- the layer-walking launch (csrc/xf_walk.hip): one kernel for the whole forward ------------------------------------------------------
def _walk_model(d=256, heads=4, enc=2, dec=2, seed=11):
    torch.manual_seed(seed)
    return build("model_10_26", dict(dim_model=d, num_heads=heads, num_encoder_layers=enc, num_decoder_layers=dec))


def _with_walk(on, fn, small=True):
    """on: the layer-walking launch (False: the per-GEMM kernels); small: forwards of at most 8 rows take the small-row form of the walk
    (whole-K GEMM stages, csrc/xf_walk.hip: xf_walk_small_kernel) — small=False keeps them on the split-K walk"""
    from sd_video_gen_amd import _lib
    # ROWS: every accumulator height the kernel is built for (the default stops at 96 rows); SPLIT: larger batches in chunks
    keys = {"SVG_XF_WALK": "1", "SVG_XF_WALK_SPLIT": "1", "SVG_XF_WALK_ROWS": "176", "SVG_XF_WALK_SMALL": "1" if small else "0"}
    old = {k: os.environ.get(k) for k in keys}
    for k, v in keys.items():
        os.environ[k] = v if (on or k == "SVG_XF_WALK_SMALL") else "0"
    _lib.env_refresh()
    try:
        return fn()
    finally:
        for k in keys:
            if old[k] is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = old[k]
        _lib.env_refresh()


@pytest.mark.parametrize("B", [1, 3, 8, 11, 16, 28, 30, 60])
def test_walk_matches_oracle_and_the_per_gemm_kernels(ctx, B):
    """every accumulator height of the walk (1, 2, 3, 4, 6, 8, 11 tiles of 16 rows), one chunk (<= 176 rows) and several (B = 30, 60):
    against the CPU oracle on sampled batch rows and against the per-GEMM HIP path on all of them; bit-identical from run to run"""
    m = _walk_model()
    sd = {k: v.clone().cpu() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(B)
    X = torch.randn(B, 6, 256, generator=g)
    pe0 = torch.zeros(B, dtype=torch.int32)
    mask = m.get_tgt_mask(6).cuda()
    run = lambda: m(X.cuda(), X.cuda(), mask, pe_row=pe0).cpu()
    out = _with_walk(True, run)
    again = _with_walk(True, run)
    old = _with_walk(False, run)
    assert torch.equal(out, again)
    assert rel_l2(out, old) < 5e-6
    for b in sorted({0, B // 2, B - 1}):
        ref = TO.forward(sd, X[b:b + 1], X[b:b + 1], 4, TO.get_tgt_mask(6))
        assert rel_l2(out[:, b:b + 1], ref) < TOL


def test_walk_masks_lengths_and_pe_rows(ctx):
    """Ts != Tt without a mask, key-padding masks, the reference's PE-row-by-batch-row quirk, zero encoder / decoder-only depth"""
    m = _walk_model(enc=1, dec=2, seed=12)
    sd = {k: v.clone().cpu() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(3)
    S, T = torch.randn(5, 5, 256, generator=g), torch.randn(5, 6, 256, generator=g)
    out = _with_walk(True, lambda: m(S.cuda(), T.cuda(), None).cpu())
    assert rel_l2(out, TO.forward(sd, S, T, 4, None)) < TOL
    pad_s = torch.rand(5, 5, generator=g) > 0.6
    pad_t = torch.rand(5, 6, generator=g) > 0.6
    pad_s[:, 0] = False
    pad_t[:, 0] = False
    m6 = m.get_tgt_mask(6)
    out = _with_walk(True, lambda: m(S.cuda(), T.cuda(), m6.cuda(), pad_s, pad_t).cpu())
    assert rel_l2(out, TO.forward(sd, S, T, 4, TO.get_tgt_mask(6), src_pad_mask=pad_s, tgt_pad_mask=pad_t)) < TOL
    old = _with_walk(False, lambda: m(S.cuda(), T.cuda(), m6.cuda(), pad_s, pad_t).cpu())
    assert rel_l2(out, old) < 5e-6


def test_walk_text_conditioned(ctx):
    from sd_video_gen_amd import config as svg_config
    from sd_video_gen_amd.transformer_text import Transformer as TextTransformer
    svg_config.set_args(["--dataset", "ucf", "--config", "model_10_26"])
    torch.manual_seed(4)
    m = TextTransformer(dim_model=128, num_heads=8, num_encoder_layers=2, num_decoder_layers=2, st_weights="synthetic").eval()
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    X = torch.randn(9, 6, 256)
    names = ["WallPushups", "PlayingGuitar", "WallPushups"] * 3
    txt = m.encode_classes(names).cpu()
    run = lambda: m(X.cuda(), names, X.cuda(), m.get_tgt_mask(6).cuda()).cpu()
    out = _with_walk(True, run)
    assert rel_l2(out, TO.forward(sd, X, X, 8, TO.get_tgt_mask(6), txt=txt)) < TOL
    assert rel_l2(out, _with_walk(False, run)) < 5e-6


def test_walk_long_sequences_and_odd_head_dim(ctx):
    """the text loop's shape family (prediction/predict_text.py conditions on 16 frames + SOS = 17 tokens): 289 score pairs per (row, head)
    (several passes of the 16-pairs-per-pass score loop), head dim 80 (not a power of two), Ts != Tt with a causal mask"""
    from sd_video_gen_amd import config as svg_config
    from sd_video_gen_amd.transformer_text import Transformer as TextTransformer
    svg_config.set_args(["--dataset", "ucf", "--config", "model_10_26"])
    torch.manual_seed(9)
    m = TextTransformer(dim_model=256, num_heads=8, num_encoder_layers=1, num_decoder_layers=2, st_weights="synthetic").eval()
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(2)
    S, T = torch.randn(3, 17, 256, generator=g), torch.randn(3, 16, 256, generator=g)
    names = ["WallPushups", "PlayingGuitar", "Archery"]
    txt = m.encode_classes(names).cpu()
    run = lambda: m(S.cuda(), names, T.cuda(), m.get_tgt_mask(16).cuda()).cpu()
    out = _with_walk(True, run)
    assert out.shape == (16, 3, 256)
    assert rel_l2(out, TO.forward(sd, S, T, 8, TO.get_tgt_mask(16), txt=txt)) < TOL
    assert rel_l2(out, _with_walk(False, run)) < 5e-6


def test_walk_from_two_threads_on_two_streams(ctx):
    """the sampling loop runs its clip groups on worker threads with a stream each: layer-walking launches issued concurrently are
    ordered on the device (an event chain inside the library) and every one of them reproduces the single-threaded result bit for bit"""
    import threading
    m = _walk_model(seed=13)
    mask = m.get_tgt_mask(6).cuda()
    Xs = [torch.randn(b, 6, 256, generator=torch.Generator().manual_seed(40 + b)).cuda() for b in (2, 8)]
    pes = [torch.zeros(x.shape[0], dtype=torch.int32) for x in Xs]

    def body():
        want = [m(x, x, mask, pe_row=p).clone() for x, p in zip(Xs, pes)]
        torch.cuda.synchronize()
        bad = []

        def worker(i):
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                for it in range(150):
                    out = m(Xs[i], Xs[i], mask, pe_row=pes[i])
                    if it % 10 == 9:
                        s.synchronize()
                        if not torch.equal(out, want[i]):
                            bad.append((i, it))
                s.synchronize()

        th = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        return bad

    assert _with_walk(True, body) == []


def test_walk_deep_narrow_model_more_workgroups_than_tiles(ctx):
    """The shape class of `config_test` (d = 256, 8 heads, 6 + 6 layers, 128 x 128 frames: D_lat 1024) — the class the walk's first run
    aborted in: 156 stages; every GEMM has 4 ... 32 tiles of 128 x 128 for 256 workgroups, so most workgroups own no tile in
    any stage (their weight stream ends before it starts), a few own tiles in the feed-forward stages only, and the embedding / output
    projections have a different K (1024) than the layers (256).  Walk forced ON, against the per-GEMM kernels and the CPU oracle."""
    from sd_video_gen_amd import config as svg_config
    from sd_video_gen_amd.transformer import Transformer
    svg_config.set_args(["--dataset", "ball", "--config", "config_test"])          # FRAME_SIZE 128 -> D_lat = 4 * 16 * 16
    torch.manual_seed(21)
    m = Transformer(dim_model=256, num_heads=8, num_encoder_layers=6, num_decoder_layers=6).eval()
    assert m.d_lat == 1024
    sd = {k: v.clone().cpu() for k, v in m.state_dict().items()}
    mask = m.get_tgt_mask(6).cuda()
    for B in (1, 3, 9):
        g = torch.Generator().manual_seed(100 + B)
        X = torch.randn(B, 6, 1024, generator=g)
        pe0 = torch.zeros(B, dtype=torch.int32)
        run = lambda: m(X.cuda(), X.cuda(), mask, pe_row=pe0).cpu()
        out = _with_walk(True, run)
        assert torch.equal(out, _with_walk(True, run))
        assert rel_l2(out, _with_walk(False, run)) < 5e-6
        for b in sorted({0, B - 1}):
            assert rel_l2(out[:, b:b + 1], TO.forward(sd, X[b:b + 1], X[b:b + 1], 8, TO.get_tgt_mask(6))) < TOL


@pytest.mark.parametrize("B", [2, 1])
def test_walk_give_up_is_reported_once_and_the_device_falls_back(ctx, B):
    """a layer-walking launch that gives up at a device-wide barrier (test hook: the barrier waits for 8 workgroups more
    than the grid has; 20 ms wall-clock give-up time) NaN-fills its WHOLE output and turns the walk off for the device; the event is raised
    — once — where the result is consumed (svg_transformer_status after a stream sync) or by the next Transformer call, while a VAE call in
    between only logs it and runs.  The calls after that run the per-GEMM kernels and are correct.  svg_env_refresh re-arms the walk.
    B = 2: 12 rows, the split-K walk; B = 1: 6 rows, the small-row walk (xf_walk_small_kernel)."""
    from sd_video_gen_amd import _lib
    m = _walk_model(seed=14)
    sd = {k: v.clone().cpu() for k, v in m.state_dict().items()}
    X = torch.randn(B, 6, 256, generator=torch.Generator().manual_seed(3))
    pe0 = torch.zeros(B, dtype=torch.int32)
    mask = m.get_tgt_mask(6).cuda()
    run = lambda: m(X.cuda(), X.cuda(), mask, pe_row=pe0).cpu()
    good = _with_walk(True, run)
    mctx = m._ctx
    # a small VAE on the fixture context: a non-Transformer entry point on the same device
    from oracle import sd_oracle as SO
    vcfg = dict(block_out=(64, 128), layers=1, groups=32, latent=4)
    ctx.configure(_lib.SVG_VAE, block_out=list(vcfg["block_out"]), layers=1, groups=32, latent=4, f16=1)
    ctx.load_state_dict(_lib.SVG_VAE, SO.seeded_weights(SO.vae_shapes(vcfg), 5))
    ctx.finalize(_lib.SVG_VAE)
    img = torch.randint(0, 256, (1, 16, 16, 3), dtype=torch.uint8, device="cuda")
    keys = {"SVG_XF_WALK": "1", "SVG_XF_WALK_TEST_GIVEUP": "1", "SVG_XF_WALK_TIMEOUT_MS": "20"}
    old = {k: os.environ.get(k) for k in keys}
    os.environ.update(keys)
    _lib.env_refresh()
    try:
        bad = run()                                        # the launch itself succeeds; its result is poisoned
        assert torch.isnan(bad).all()
        z = ctx.vae_encode(img)                            # did nothing wrong: runs (the event is logged and stays pending)
        assert torch.isfinite(z).all()
        with pytest.raises(RuntimeError, match="gave up"):
            mctx.transformer_status()                      # where the forward's result is consumed
        mctx.transformer_status()                          # raised once
        os.environ["SVG_XF_WALK_TEST_GIVEUP"] = "0"        # (not refreshed: the library must already have fallen back by itself)
        after = run()                                      # no raise any more, per-GEMM kernels
        assert torch.isfinite(after).all() and rel_l2(after, good) < 5e-6
        assert rel_l2(after[:, :1], TO.forward(sd, X[:1], X[:1], 4, TO.get_tgt_mask(6))) < TOL
        # without a status query the NEXT Transformer call raises (round 4's contract)
        os.environ["SVG_XF_WALK_TEST_GIVEUP"] = "1"
        _lib.env_refresh()
        assert torch.isnan(run()).all()
        with pytest.raises(RuntimeError, match="gave up"):
            run()
    finally:
        for k in keys:
            if old[k] is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = old[k]
        _lib.env_refresh()
    assert torch.equal(_with_walk(True, run), good)        # re-armed by the refresh: the walk again, same bits


@pytest.mark.parametrize("shape", ["kitti", "config_test", "ucf"])
def test_walk_small_rows_single_clip(ctx, shape):
    """The small-row form of the walk (at most 8 rows = one clip): whole-K GEMM stages with the LayerNorm of their input and
    bias / ReLU / residual / embedding epilogue folded in — 5 + 8 instead of 9 + 16 stages per encoder / decoder layer.  Full-size models of
    three configs (d = 2048 4 + 8 layers D_lat 256; d = 256 6 + 6 layers D_lat 1024; d = 2048 D_lat 1024), one clip of 6 tokens, and 2 clips of
    4 tokens (8 rows, reference PE quirk: row = batch row), Ts != Tt with key-padding masks: against the split-K walk, the per-GEMM kernels
    and the CPU oracle; bit-identical from run to run."""
    from sd_video_gen_amd import config as svg_config
    from sd_video_gen_amd.transformer import Transformer
    cfgname, kw, heads = {"kitti": ("1_16_kitti_L1_64", dict(dim_model=2048, num_heads=8, num_encoder_layers=4, num_decoder_layers=8), 8),
                          "config_test": ("config_test", dict(dim_model=256, num_heads=8, num_encoder_layers=6, num_decoder_layers=6), 8),
                          "ucf": ("11_27_ucf_final", dict(dim_model=2048, num_heads=8, num_encoder_layers=2, num_decoder_layers=2), 8)}[shape]
    svg_config.set_args(["--dataset", "ball", "--config", cfgname])
    torch.manual_seed(31)
    m = Transformer(**kw).eval()
    D = m.d_lat
    sd = {k: v.clone().cpu() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(7)
    # (1) one clip, 6 tokens, causal mask: the single-clip sampling call
    X = torch.randn(1, 6, D, generator=g)
    mask = m.get_tgt_mask(6).cuda()
    pe0 = torch.zeros(1, dtype=torch.int32)
    run = lambda: m(X.cuda(), X.cuda(), mask, pe_row=pe0).cpu()
    out = _with_walk(True, run)
    assert torch.equal(out, _with_walk(True, run))
    assert rel_l2(out, _with_walk(True, run, small=False)) < 5e-6
    assert rel_l2(out, _with_walk(False, run)) < 5e-6
    assert rel_l2(out, TO.forward(sd, X, X, heads, TO.get_tgt_mask(6))) < TOL
    # (2) two clips of 4 tokens (8 rows), the reference's PE-row-by-batch-row quirk (no pe_row)
    X2 = torch.randn(2, 4, D, generator=g)
    m4 = m.get_tgt_mask(4).cuda()
    run2 = lambda: m(X2.cuda(), X2.cuda(), m4).cpu()
    out2 = _with_walk(True, run2)
    assert rel_l2(out2, _with_walk(False, run2)) < 5e-6
    assert rel_l2(out2, TO.forward(sd, X2, X2, heads, TO.get_tgt_mask(4))) < TOL
    # (3) Ts != Tt, key-padding masks on both sides
    S, T = torch.randn(1, 5, D, generator=g), torch.randn(1, 7, D, generator=g)
    ps = torch.tensor([[False, False, True, False, True]]); pt = torch.tensor([[False, False, False, True, False, False, True]])
    m7 = m.get_tgt_mask(7).cuda()
    run3 = lambda: m(S.cuda(), T.cuda(), m7, ps, pt).cpu()
    out3 = _with_walk(True, run3)
    assert rel_l2(out3, _with_walk(False, run3)) < 5e-6
    assert rel_l2(out3, TO.forward(sd, S, T, heads, TO.get_tgt_mask(7), src_pad_mask=ps, tgt_pad_mask=pt)) < TOL
    # timing of the single-clip call, printed (profiles/README.md)
    def timed(small, walk=True):
        def f():
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(3):
                m(X.cuda(), X.cuda(), mask, pe_row=pe0)
            e0.record()
            for _ in range(20):
                m(X.cuda(), X.cuda(), mask, pe_row=pe0)
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) / 20
        return _with_walk(walk, f, small=small)
    print("[walk] %s, 6 rows: small-row walk %.3f ms | split-K walk %.3f ms | per-GEMM kernels %.3f ms" % (shape, timed(True), timed(False), timed(True, walk=False)))
