"""CPU: host logic, C-ABI surface, and the N>1 path on gloo (world_size 2)."""
import ctypes
import os
import re
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_library_exports_every_declared_symbol():
    """dlopen works without a GPU; every function declared in include/svg_hip.h is exported and bound."""
    from sd_video_gen_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "svg_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(svg_\w+)\s*\(", hdr))
    assert len(declared) >= 25
    lib = _lib.load()
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), name
        assert name in _lib.SIGNATURES, "no ctypes signature for " + name
    assert set(_lib.SIGNATURES) == declared
    assert lib.svg_version().decode().startswith("svg_hip")


def test_committed_counter_summary_belongs_to_this_build():
    """bench.py fills `roofline.traffic` from the newest profiles/*pmc_summary.json only when its `src_hash` is the hash of the sources the
    library was built from (svg_version()); a kernel-source edit without re-taking tools/pmc_step.sh would silently turn the field into null.
    Checks library == sources (the Makefile's own hash) == committed summary."""
    import json
    import subprocess
    from sd_video_gen_amd import _lib
    csrc = os.path.join(ROOT, "sd-video-gen_amd", "csrc")
    db = subprocess.run(["make", "-pn", "-C", csrc], capture_output=True, text=True).stdout
    of_sources = re.search(r"^SRCHASH := (\w+)", db, flags=re.M).group(1)
    assert _lib.source_hash() == of_sources, "libsvg_hip.so is older than its sources: run __graft_entry__.build()"
    names = sorted(n for n in os.listdir(os.path.join(ROOT, "profiles")) if n.endswith("pmc_summary.json"))
    with open(os.path.join(ROOT, "profiles", names[-1])) as f:
        summary = json.load(f)
    assert summary["src_hash"] == of_sources, names[-1] + " was taken from another build of the kernels: re-run tools/pmc_step.sh"


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU behaviour")
def test_product_path_fails_loudly_without_gpu():
    from sd_video_gen_amd import _lib, config as svg_config
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.Context()
    h = ctypes.c_void_p()
    assert _lib.load().svg_create(0, ctypes.byref(h)) != 0
    assert len(_lib.load().svg_last_error(None)) > 0
    svg_config.set_args(["--dataset", "synthetic-ball", "--config", "model_10_26"])
    from sd_video_gen_amd.sd_utils import SDUtils
    with pytest.raises(RuntimeError):
        SDUtils()


def test_layout_tables_match_oracle_tables():
    from oracle import sd_oracle as SO
    from sd_video_gen_amd import sd_layout
    assert sd_layout.unet_shapes() == SO.unet_shapes() and sd_layout.vae_shapes() == SO.vae_shapes()
    assert sd_layout.count(sd_layout.unet_shapes()) == 859_520_964


def test_shard_range_and_seeds():
    from sd_video_gen_amd import sharding
    for n in (1, 5, 8, 16, 17):
        for ws in (1, 2, 3, 8):
            spans = [sharding.shard_range(n, r, ws) for r in range(ws)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(ws - 1))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1
            seeds = sum([sharding.clip_seeds(100, a, b) for a, b in spans], [])
            assert seeds == list(range(100, 100 + n))            # world-size invariant


def test_bouncing_ball_clips():
    from sd_video_gen_amd.predict import bouncing_ball_clips
    a = bouncing_ball_clips(3, 64, 5, seed=2)
    assert a.shape == (3, 5, 64, 64, 3) and a.dtype == torch.uint8
    assert torch.equal(a, bouncing_ball_clips(3, 64, 5, seed=2))
    assert torch.equal(a[1], bouncing_ball_clips(1, 64, 5, seed=3)[0])   # clip c depends only on seed + c
    assert set(a.unique().tolist()) == {0, 255}
    assert not torch.equal(a[0, 0], a[0, 4])


def _gather_worker(rank, ws, port, n_clips, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    from sd_video_gen_amd import sharding
    a, b = sharding.shard_range(n_clips, rank, ws)
    local = torch.stack([torch.full((3, 4), float(c)) for c in range(a, b)]) if b > a else torch.zeros(0, 3, 4)
    out = sharding.gather_clips(local, n_clips)
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_clips", [4, 5])
def test_gather_clips_gloo_world2(n_clips):
    """the one collective of the N>1 path (all-gather of the finished clips), ragged shards included."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + n_clips
    procs = [ctx.Process(target=_gather_worker, args=(r, 2, port, n_clips, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = torch.stack([torch.full((3, 4), float(c)) for c in range(n_clips)])
    for _, out in res:
        assert torch.equal(out, want)


def _fake_sampler(clips, seeds):
    """stands where sample_clips stands (the real one needs the GPU): outputs are pure functions of the clip seed"""
    lat = torch.stack([torch.full((6, 16), float(s)) + torch.arange(16.0) for s in seeds]) if seeds else torch.zeros(0, 6, 16)
    frames = torch.stack([torch.full((6, 8, 8, 3), s % 251, dtype=torch.uint8) for s in seeds]) if seeds else \
        torch.zeros(0, 6, 8, 8, 3, dtype=torch.uint8)
    assert clips.shape[0] == len(seeds)
    return lat, frames


def _run_sharded_worker(rank, ws, port, n_clips, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    from sd_video_gen_amd.predict import run_sharded
    calls = []
    real = dist.all_gather
    dist.all_gather = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    clips = torch.zeros(n_clips, 5, 8, 8, 3, dtype=torch.uint8)
    lat, frames = run_sharded(clips, _fake_sampler, base_seed=40)
    q.put((rank, lat, frames, len(calls)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_clips", [4, 5, 1])
def test_run_sharded_one_allgather_gloo_world2(n_clips):
    """predict.main's N>1 path (shard -> sample -> ONE all-gather of latents + frames), ragged and empty shards included;
    the result equals the world-size-1 run."""
    from sd_video_gen_amd.predict import run_sharded
    want_lat, want_frames = run_sharded(torch.zeros(n_clips, 5, 8, 8, 3, dtype=torch.uint8), _fake_sampler, base_seed=40)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000) + n_clips
    procs = [ctx.Process(target=_run_sharded_worker, args=(r, 2, port, n_clips, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for _, lat, frames, n_calls in res:
        assert n_calls == 1                                         # one collective ends the step
        assert lat.dtype == torch.float32 and frames.dtype == torch.uint8
        assert torch.equal(lat, want_lat) and torch.equal(frames, want_frames)


class _FakeCtx:
    device = torch.device("cpu")


class _FakeI3D:
    ctx, num_classes = _FakeCtx(), 400


def _fvd_worker(rank, ws, port, groups_per_rank, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    from sd_video_gen_amd import fvd, predict_text as PT
    fvd.get_fvd_logits = lambda g, i3d, device=None: g.reshape(g.shape[0], -1)[:, :400].float()      # stands where the I3D stands
    fvd.frechet_distance = lambda a, b, ctx=None: float((a.mean(0) - b.mean(0)).pow(2).sum())
    n_real, n_fake = groups_per_rank[rank]
    mk = lambda n, base: [torch.full((16, 2, 4, 25, 3), base + 10 * rank + i, dtype=torch.uint8) for i in range(n)]
    val, real, fake = PT.fvd_from_stacks(mk(n_real, 1), mk(n_fake, 3), _FakeI3D())
    q.put((rank, val, tuple(real.shape), tuple(fake.shape)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("groups", [((1, 1), (0, 0)), ((0, 0), (0, 0)), ((2, 1), (1, 1))])
def test_fvd_decision_is_collective_gloo_world2(groups):
    """a rank whose shard produced no complete group of 16 clips (31 clips on 2 ranks) must still enter the all_gathers
    of the logits; whether an FVD exists is decided on the GATHERED row counts, on every rank alike — no rank hangs, all agree."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000) + 7 * sum(sum(g) for g in groups)
    procs = [ctx.Process(target=_fvd_worker, args=(r, 2, port, groups, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n_real, n_fake = 16 * (groups[0][0] + groups[1][0]), 16 * (groups[0][1] + groups[1][1])
    assert res[0][1:] == res[1][1:]
    assert res[0][2] == (n_real, 400) and res[0][3] == (n_fake, 400)
    assert (res[0][1] is None) == (n_real == 0 or n_fake == 0)


def test_save_frames_bytes(tmp_path):
    """prediction/predict.py:201-223: one PNG per frame; predicted frames carry a 1-px red border (BGR [0,0,255]); frames are
    BGR in memory and written like cv2.imwrite (so the file holds RGB = reversed channels)."""
    import numpy as np
    from PIL import Image
    from sd_video_gen_amd.predict import save_frames
    g = np.random.default_rng(0)
    frames = g.integers(0, 256, size=(3, 8, 8, 3), dtype=np.uint8)
    save_frames(frames, [False, True, True], str(tmp_path / "o"))
    assert sorted(os.listdir(tmp_path / "o")) == ["0.png", "1.png", "2.png"]
    a0 = np.asarray(Image.open(tmp_path / "o" / "0.png"))
    assert a0.shape == (8, 8, 3) and np.array_equal(a0[..., ::-1], frames[0])
    a1 = np.asarray(Image.open(tmp_path / "o" / "1.png"))
    assert a1.shape == (10, 10, 3)
    assert np.array_equal(a1[1:-1, 1:-1, ::-1], frames[1])
    border = np.concatenate([a1[0], a1[-1], a1[:, 0], a1[:, -1]])
    assert (border == np.array([255, 0, 0], dtype=np.uint8)).all()   # RGB on disk = BGR (0,0,255) in memory


def test_text_model_loads_reference_checkpoint_layout():
    """a reference text-model checkpoint (state_dict of models/transformer_text.py) also holds the SentenceTransformer's
    ``sent_transformer.0.auto_model.*`` tensors (MiniLM): they fill the class-name encoder, everything else loads strictly, and a
    checkpoint saved here carries them again under the same names (loads in the reference under strict=True)."""
    from sd_video_gen_amd import config as svg_config, minilm, sd_layout
    from sd_video_gen_amd.transformer_text import Transformer
    svg_config.set_args(["--dataset", "ball", "--config", "model_10_26"])
    torch.manual_seed(0)
    m = Transformer(dim_model=32, num_heads=4, num_encoder_layers=1, num_decoder_layers=1)
    assert not m.sent_transformer.loaded and not any(k.startswith("sent_transformer.") for k in m.state_dict())
    sd = {k: v.clone() + 1.0 if v.is_floating_point() else v.clone() for k, v in m.state_dict().items()}
    tiny = dict(vocab=1200, d_model=384, heads=12, layers=1, ffn=64, max_pos=32)
    m.sent_transformer.cfg.update(tiny)
    bert = sd_layout.seeded_weights(minilm.bert_shapes(tiny), 9)
    sd.update({"sent_transformer.0.auto_model." + k: v for k, v in bert.items()})
    sd["sent_transformer.0.auto_model.embeddings.position_ids"] = torch.arange(32).unsqueeze(0)
    r = m.load_state_dict(sd)                                       # strict
    assert not r.missing_keys and not r.unexpected_keys
    assert torch.equal(m.out.bias, sd["out.bias"]) and m.sent_transformer.loaded
    out = m.state_dict()
    assert set(out) == set(sd) and all(torch.equal(out[k], sd[k]) for k in sd)          # round trip, MiniLM tensors included
    with pytest.raises(RuntimeError):
        m.load_state_dict({k: v for k, v in sd.items() if k != "out.bias"})
    with pytest.raises(FileNotFoundError):                          # an encoder without weights refuses, like a failed from_pretrained
        Transformer(dim_model=32, num_heads=4, num_encoder_layers=1, num_decoder_layers=1).encode_classes(["Archery"])
    # the checkpoint brought REAL MiniLM weights but no WordPiece vocabulary — hashed stand-in ids must not reach them
    assert not m.sent_transformer.synthetic and isinstance(m.sent_transformer.tokenizer, minilm.StandInWordPiece)
    with pytest.raises(FileNotFoundError, match="vocab"):
        m.encode_classes(["Archery"])


def test_minilm_vocab_is_separate_from_the_weights(tmp_path, monkeypatch):
    """the WordPiece vocabulary comes from vocab= / $SVG_MINILM_VOCAB / the weights directory, independently of the weights; seeded
    synthetic weights are the only ones the crc32 stand-in tokenizer may serve (no GPU: tokenisation only)."""
    from sd_video_gen_amd import minilm, sd_layout
    tiny = dict(vocab=1200, d_model=384, heads=12, layers=1, ffn=64, max_pos=32)
    vf = tmp_path / "vocab.txt"
    vf.write_text("\n".join(["[PAD]"] + ["[unused%d]" % i for i in range(99)] + ["[UNK]", "[CLS]", "[SEP]", "[MASK]", "apply", "eye", "make", "##up", "archery"]) + "\n")
    monkeypatch.delenv("SVG_MINILM_WEIGHTS", raising=False)
    monkeypatch.delenv("SVG_MINILM_VOCAB", raising=False)
    monkeypatch.delenv("SVG_MINILM_STANDIN_TOKENIZER", raising=False)
    real = sd_layout.seeded_weights(minilm.bert_shapes(tiny), 4)                 # handed over as a dict = "real" weights
    enc = minilm.SentenceEncoder(weights=dict(real), cfg=tiny)
    assert enc.loaded and not enc.synthetic and isinstance(enc.tokenizer, minilm.StandInWordPiece)
    with pytest.raises(FileNotFoundError, match="vocab"):
        enc.encode(["Apply Eye Makeup"])
    enc = minilm.SentenceEncoder(weights=dict(real), cfg=tiny, vocab=str(vf))
    ids, lens = enc.tokenizer(["Apply Eye Makeup", "Archery"])
    cls, sep = 101, 102
    assert ids[0].tolist() == [cls, 104, 105, 106, 107, sep] and ids[1].tolist() == [cls, 108, sep, 0, 0, 0] and lens.tolist() == [6, 3]
    monkeypatch.setenv("SVG_MINILM_VOCAB", str(vf))
    assert isinstance(minilm.SentenceEncoder(weights=dict(real), cfg=tiny).tokenizer, minilm._HFWordPiece)
    monkeypatch.delenv("SVG_MINILM_VOCAB")
    with pytest.raises(FileNotFoundError):
        minilm.SentenceEncoder(weights=dict(real), cfg=tiny, vocab=str(tmp_path / "missing.txt"))
    syn = minilm.SentenceEncoder(weights="synthetic", cfg=tiny, seed=1)
    assert syn.synthetic and isinstance(syn.tokenizer, minilm.StandInWordPiece)


def test_checkpoint_format_roundtrip(tmp_path):
    """checkpoints are plain `torch.save(model.state_dict())` files (trainer.py:469-480): tensors only, loadable with
    weights_only=True, keys as SURVEY appendix B — in both directions."""
    from sd_video_gen_amd import config as svg_config
    from sd_video_gen_amd.transformer import Transformer
    svg_config.set_args(["--dataset", "ball", "--config", "model_10_26"])
    torch.manual_seed(0)
    m = Transformer(dim_model=32, num_heads=4, num_encoder_layers=1, num_decoder_layers=2)
    p = str(tmp_path / "model_10_26_0_train.pt")
    torch.save(m.state_dict(), p)
    assert os.path.getsize(p) < 2 * sum(t.numel() * 4 for t in m.state_dict().values())     # no module pickled along
    sd = torch.load(p, map_location="cpu", weights_only=True)
    ref_keys = set(torch.load(os.path.join(ROOT, "tests", "golden", "transformer_tiny.pt"), weights_only=False)["state_dict"])
    assert set(sd) == ref_keys                                         # the reference module's own key set (G1)
    m2 = Transformer(dim_model=32, num_heads=4, num_encoder_layers=1, num_decoder_layers=2)
    m2.load_state_dict(sd)
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))


def test_bench_refuses_mismatched_world_size():
    """bench.py --gpus N under a launcher that started another number of ranks stops with a clear message (and never
    initialises the GPU in the parent when it spawns the ranks itself)."""
    import subprocess
    env = dict(os.environ, WORLD_SIZE="1", RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


def test_cli_surface_parses_like_the_reference():
    from sd_video_gen_amd import config as svg_config
    import trainers.trainer as tr
    svg_config.set_args(["--dataset", "ball", "--config", "config_test", "--debug", "True"])
    assert callable(tr.main)
    t = tr.Trainer(sd_utils=object())                       # the SD side is only needed once batches are encoded
    assert t.SOS_token.shape == (1, 1, t.config.FRAME_SIZE ** 2 // 64 * 4) and float(t.SOS_token[0, 0, 0]) == 2.0
    assert t.criterion(use_mse=True, use_L1=True) is None                              # trainer.py:107-109
    c = t.criterion(use_mse=False, use_L1=True, use_gdl=True, lambda_gdl=0.5, alpha=1, use_contrastive=True, lambda_contrastive=0.025)
    cfg = c.cfg(frames_to_predict=5, dropout_p=0.1, seed=7)
    assert (cfg.w_mse, cfg.w_l1, cfg.w_gdl, cfg.gdl_alpha) == (0.0, 1.0, 0.5, 1.0) and abs(cfg.w_contrastive - 0.025) < 1e-9
    assert cfg.feat_h == t.config.FRAME_SIZE // 8 and cfg.frames_to_predict == 5 and cfg.seed == 7
    import prediction.predict as pp
    assert callable(pp.predict) and callable(pp.main)
    import models.transformer, utils.config, utils.sd_utils   # noqa: F401,E401


def test_loop_oracle_matches_reference_trace_shapes():
    """the oracle's per-clip loop reproduces the reference's loop plumbing (G5) when the VAE is the identity-free part:
    all_latents has 4 + N frames and the window is 5."""
    from oracle import loop_oracle, sd_oracle as SO, transformer_oracle as TO
    from sd_video_gen_amd.predict import bouncing_ball_clips
    g = torch.load(os.path.join(ROOT, "tests", "golden", "transformer_tiny.pt"), weights_only=False)
    vcfg = dict(block_out=(64, 64, 64, 64), layers=1, groups=32, latent=4)
    vsd = SO.seeded_weights(SO.vae_shapes(vcfg), 4)
    clip = bouncing_ball_clips(1, 64, 5, seed=1)[0]
    gen = torch.Generator().manual_seed(0)
    noise = {"cond": torch.randn(5, 4, 8, 8, generator=gen)}
    out = loop_oracle.sample_clip(g["state_dict"], g["num_heads"], vsd, clip, 3, noise, vae_cfg=vcfg)
    assert out.shape == (1, 7, 256) and torch.isfinite(out).all()


def test_host_thread_cap_follows_the_cgroup_quota(tmp_path, monkeypatch):
    """_lib.fit_host_threads(): torch's intra-op pool is capped to a quarter of the CFS quota (cpu.max = "quota period") unless the user
    chose a thread count; an unlimited cgroup ("max") leaves the affinity count (profiles/r03_throttle_*.txt: why this exists)."""
    import torch
    from sd_video_gen_amd import _lib
    before = torch.get_num_threads()
    try:
        f = tmp_path / "cpu.max"
        f.write_text("1600000 100000\n")
        monkeypatch.setenv("SVG_CGROUP_CPU_MAX", str(f))
        monkeypatch.delenv("OMP_NUM_THREADS", raising=False)
        monkeypatch.delenv("SVG_HOST_THREADS", raising=False)
        monkeypatch.delenv("LOCAL_WORLD_SIZE", raising=False)
        monkeypatch.delenv("WORLD_SIZE", raising=False)
        import os
        n_aff = len(os.sched_getaffinity(0))
        assert _lib.host_cpu_quota() == (min(16, n_aff), True)
        torch.set_num_threads(max(before, 8))
        assert _lib.fit_host_threads() == min(max(before, 8), max(1, min(16, n_aff) // 4))
        monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")          # eight ranks share the node's quota
        torch.set_num_threads(8)
        assert _lib.fit_host_threads() == 1
        monkeypatch.delenv("LOCAL_WORLD_SIZE")
        monkeypatch.setenv("WORLD_SIZE", "64")              # a multi-node job: WORLD_SIZE says nothing about THIS node's ranks
        torch.set_num_threads(8)
        assert _lib.fit_host_threads() == min(8, max(1, min(16, n_aff) // 4))
        monkeypatch.delenv("WORLD_SIZE")
        f.write_text("max 100000\n")                       # no CFS quota: nothing throttles, the pool is left alone
        assert _lib.host_cpu_quota() == (n_aff, False)
        torch.set_num_threads(8)
        assert _lib.fit_host_threads() == 8
        monkeypatch.setenv("SVG_HOST_THREADS", "3")
        torch.set_num_threads(8)
        assert _lib.fit_host_threads() == 3
        monkeypatch.setenv("OMP_NUM_THREADS", "7")          # an explicit user choice wins: nothing is changed
        torch.set_num_threads(5)
        assert _lib.fit_host_threads() == 5
    finally:
        torch.set_num_threads(before)


def test_reference_module_paths_exist():
    """the reference's import paths resolve to the mirror (thin re-exports; nothing computes without the GPU)"""
    import importlib
    for mod, names in (("evaluation.fvd_2", ("preprocess", "get_fvd_logits", "get_logits", "frechet_distance", "load_i3d_pretrained", "all_gather")),
                       ("evaluation.pytorch_i3d", ("InceptionI3d",)),
                       ("prediction.predict_text", ("predict", "main", "find_classes", "splitClassNames")),
                       ("prediction.predict", ("predict", "main")),
                       ("models.transformer", ("Transformer",)), ("models.transformer_text", ("Transformer",)),
                       ("utils.sd_utils", ("SDUtils",)), ("utils.config", ("parse_config_args",))):
        m = importlib.import_module(mod)
        for n in names:
            assert hasattr(m, n), "%s.%s" % (mod, n)
