"""GPU: the data formats either side of the path and the ownership rules of the boundary.

  * local diffusers-format weight directories ($SVG_SD_WEIGHTS: <sub>/config.json + diffusion_pytorch_model.safetensors|.bin)
    stand where `from_pretrained('CompVis/stable-diffusion-v1-4', subfolder=...)` stands (utils/sd_utils.py:52-66);
  * `./checkpoints/<config>_<index>_<mode>.pt` (prediction/predict.py:50-52) through `main()`; PNG output with the border;
  * missing weights raise (as the reference's loaders do) unless synthetic weights are asked for;
  * two models on one library context do not compute with each other's weights.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import rel_l2

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import sd_oracle as SO, transformer_oracle as TO  # noqa: E402
from sd_video_gen_amd import _lib  # noqa: E402

pytestmark = pytest.mark.gpu

VCFG = dict(block_out=(64, 128, 128, 128), layers=1, groups=32, latent=4)
UCFG = dict(block_out=(64, 128), layers=1, heads=4, ctx_dim=768, groups=32, in_ch=4, out_ch=4, attn=(1, 0))


def write_diffusers_dir(root, vsd, usd, bin_format=False):
    """a directory laid out like a local clone of the SD repo: vae/ and unet/ with config.json + weights"""
    from safetensors.torch import save_file
    for sub, sd, cfg in (("vae", vsd, {"block_out_channels": list(VCFG["block_out"]), "layers_per_block": VCFG["layers"],
                                      "norm_num_groups": 32, "latent_channels": 4, "_class_name": "AutoencoderKL"}),
                         ("unet", usd, {"block_out_channels": list(UCFG["block_out"]), "layers_per_block": UCFG["layers"],
                                        "attention_head_dim": UCFG["heads"], "cross_attention_dim": 768, "norm_num_groups": 32,
                                        "in_channels": 4, "out_channels": 4, "_class_name": "UNet2DConditionModel",
                                        "down_block_types": ["CrossAttnDownBlock2D", "DownBlock2D"]})):
        os.makedirs(os.path.join(root, sub), exist_ok=True)
        with open(os.path.join(root, sub, "config.json"), "w") as f:
            json.dump(cfg, f)
        sd = {k: v.contiguous() for k, v in sd.items()}
        if bin_format:
            torch.save(sd, os.path.join(root, sub, "diffusion_pytorch_model.bin"))
        else:
            save_file(sd, os.path.join(root, sub, "diffusion_pytorch_model.safetensors"))


@pytest.fixture()
def tiny_nets():
    return SO.seeded_weights(SO.vae_shapes(VCFG), 3), SO.seeded_weights(SO.unet_shapes(UCFG), 4)


def _args(denoise=True, cfg="model_10_26"):
    from sd_video_gen_amd import config as svg_config
    svg_config.set_args(["--dataset", "synthetic-ball", "--config", cfg] + (["--denoise", "1"] if denoise else []))


def test_missing_weights_raise_unless_synthetic(monkeypatch):
    from sd_video_gen_amd.sd_utils import SDUtils
    monkeypatch.delenv("SVG_SD_WEIGHTS", raising=False)
    monkeypatch.delenv("SVG_ALLOW_SYNTHETIC_WEIGHTS", raising=False)
    _args()
    with pytest.raises(FileNotFoundError, match="vae"):
        SDUtils(verbose=False)
    arch = {"vae": VCFG, "unet": UCFG}
    sdu = SDUtils(weights="synthetic", arch=arch, verbose=False)
    assert sdu.vae_source == "synthetic" and sdu.unet_source == "synthetic"
    with pytest.raises(ValueError):
        SDUtils(weights={"vae": "random"}, arch=arch, verbose=False)
    del sdu
    monkeypatch.setenv("SVG_ALLOW_SYNTHETIC_WEIGHTS", "1")
    assert SDUtils(arch=arch, verbose=False).unet_source == "synthetic"


@pytest.mark.parametrize("bin_format", [False, True])
def test_local_diffusers_directory(tmp_path, monkeypatch, tiny_nets, bin_format):
    """$SVG_SD_WEIGHTS: architecture from config.json, tensors from .safetensors / .bin; results equal the same weights
    handed over as dicts, bit for bit."""
    from sd_video_gen_amd.sd_utils import SDUtils
    vsd, usd = tiny_nets
    write_diffusers_dir(str(tmp_path), vsd, usd, bin_format)
    _args()
    monkeypatch.delenv("SVG_SD_WEIGHTS", raising=False)
    a = SDUtils(weights={"vae": vsd, "unet": usd}, arch={"vae": VCFG, "unet": UCFG}, verbose=False, text_embeddings=torch.zeros(2, 77, 768))
    g = torch.Generator().manual_seed(0)
    img = torch.randint(0, 256, (2, 64, 64, 3), dtype=torch.uint8, generator=g)
    eps = torch.randn(2, 4, 8, 8, generator=g)
    x = torch.randn(2, 4, 16, 16, generator=g)
    emb = torch.randn(2, 77, 768, generator=g)
    za = a.encode_img(img, eps=eps.cuda())
    ea = a.unet(x.cuda(), 500, encoder_hidden_states=emb.cuda())["sample"]
    fa = a.decode_img_latents(za)
    del a
    monkeypatch.setenv("SVG_SD_WEIGHTS", str(tmp_path))
    with pytest.raises(FileNotFoundError, match="CLIP"):          # a directory with real-looking UNet weights but no text encoder
        SDUtils(verbose=False)
    b = SDUtils(verbose=False, text_embeddings=torch.zeros(2, 77, 768))   # no arch=, no weights=: everything from the directory
    assert b.vae_source.startswith("local:") and b.unet_source.startswith("local:")
    assert tuple(b.unet_arch["block_out"]) == UCFG["block_out"] and b.unet_arch["heads"] == 4 and tuple(b.unet_arch["attn"]) == (1, 0)
    assert b.vae.n_params == SO.count(SO.vae_shapes(VCFG)) and b.unet.n_params == SO.count(SO.unet_shapes(UCFG))
    assert torch.equal(b.encode_img(img, eps=eps.cuda()), za)
    assert torch.equal(b.unet(x.cuda(), 500, encoder_hidden_states=emb.cuda())["sample"], ea)
    assert np.array_equal(b.decode_img_latents(za), fa)
    # a directory that lacks a tensor names it
    os.remove(os.path.join(str(tmp_path), "unet", "diffusion_pytorch_model." + ("bin" if bin_format else "safetensors")))
    bad = dict(usd)
    bad.pop("mid_block.attentions.0.transformer_blocks.0.attn2.to_k.weight")
    write_diffusers_dir(str(tmp_path), vsd, bad, bin_format)
    del b
    with pytest.raises(ValueError, match="attn2.to_k"):
        SDUtils(verbose=False, text_embeddings=torch.zeros(2, 77, 768))


def test_main_checkpoint_and_png_output(tmp_path, monkeypatch, tiny_nets):
    """prediction/predict.py __main__: checkpoint ./checkpoints/<config>_<index>_<mode>.pt, outputs/<n>/<i>.png with the red
    border on predicted frames; a missing checkpoint raises like torch.load does in the reference."""
    from PIL import Image
    from sd_video_gen_amd import predict as P
    from sd_video_gen_amd import config as svg_config
    from sd_video_gen_amd.transformer import Transformer
    vsd, usd = tiny_nets
    write_diffusers_dir(str(tmp_path / "sd"), vsd, usd)
    monkeypatch.setenv("SVG_SD_WEIGHTS", str(tmp_path / "sd"))
    monkeypatch.delenv("SVG_ALLOW_SYNTHETIC_WEIGHTS", raising=False)
    monkeypatch.chdir(tmp_path)
    argv = ["--dataset", "synthetic-ball", "--config", "config_test", "--pred_frames", "2", "--index", "3", "--mode", "test",
            "--save_output", "True"]
    with pytest.raises(FileNotFoundError, match="config_test_3_test.pt"):
        P.main(argv)
    svg_config.set_args(argv)
    torch.manual_seed(21)
    m = Transformer(dim_model=256, num_heads=8, num_encoder_layers=6, num_decoder_layers=6)
    os.makedirs("checkpoints")
    torch.save(m.state_dict(), os.path.join("checkpoints", "config_test_3_test.pt"))          # trainer.py:469-480 layout
    lat = P.main(argv)
    assert lat.shape == (8, 6, 1024)                                                            # 8 synthetic clips, 4 + 2 frames
    # the checkpoint's weights are the ones that ran: clip 0 against the oracle
    noise = {"cond": torch.randn((5, 4, 16, 16), generator=torch.Generator(device="cuda").manual_seed(0), device="cuda").cpu()}
    from oracle import loop_oracle
    ref = loop_oracle.sample_clip({k: v for k, v in m.state_dict().items()}, 8, vsd, P.bouncing_ball_clips(1, 128, 5, seed=0)[0],
                                  2, noise, vae_cfg=VCFG)
    assert rel_l2(lat[:1].cpu(), ref) < 3e-2
    dirs = sorted(os.listdir("outputs"), key=int)
    assert len(dirs) == 8
    files = sorted(os.listdir(os.path.join("outputs", dirs[0])), key=lambda f: int(f[:-4]))
    assert files == ["%d.png" % i for i in range(6)]
    cond = np.asarray(Image.open(os.path.join("outputs", dirs[0], "0.png")))
    pred = np.asarray(Image.open(os.path.join("outputs", dirs[0], "5.png")))
    assert cond.shape == (128, 128, 3) and pred.shape == (130, 130, 3)
    assert (pred[0] == np.array([255, 0, 0], dtype=np.uint8)).all() and (pred[:, -1] == np.array([255, 0, 0], dtype=np.uint8)).all()


def test_main_reads_png_clip_folders(tmp_path, monkeypatch, tiny_nets):
    """`--dataset ball --folder <dir>`: the directory crawler of loaders/bouncing_ball_loader.py feeds main(); clips come out in
    crawl order with per-clip seeds, so the result equals sampling the same frames directly."""
    from PIL import Image
    from sd_video_gen_amd import predict as P, config as svg_config
    from sd_video_gen_amd.sd_utils import SDUtils
    from sd_video_gen_amd.transformer import Transformer
    vsd, usd = tiny_nets
    write_diffusers_dir(str(tmp_path / "sd"), vsd, usd)
    monkeypatch.setenv("SVG_SD_WEIGHTS", str(tmp_path / "sd"))
    monkeypatch.setenv("SVG_ALLOW_SYNTHETIC_WEIGHTS", "1")       # initial Transformer parameters instead of a checkpoint
    monkeypatch.chdir(tmp_path)
    clips = P.bouncing_ball_clips(3, 64, 5, seed=7).numpy()       # (3,5,64,64,3) BGR == RGB (grey)
    for c in range(3):
        for t in range(5):
            d = tmp_path / "data" / "test" / ("%04d" % (c + 1))
            os.makedirs(d, exist_ok=True)
            Image.fromarray(clips[c, t]).save(d / ("frame_%03d.png" % t))
    argv = ["--dataset", "ball", "--folder", str(tmp_path / "data"), "--config", "model_10_26", "--pred_frames", "2"]
    torch.manual_seed(3)
    lat = P.main(argv)
    assert lat.shape == (3, 6, 256)
    svg_config.set_args(argv)
    torch.manual_seed(3)
    sdu = SDUtils(verbose=False)
    m = Transformer(num_tokens=0, dim_model=256, num_heads=8, num_encoder_layers=6, num_decoder_layers=6, dropout_p=0.1).eval()
    direct = P.sample_clips(m, sdu, torch.from_numpy(clips).cuda(), 2, seeds=[0, 1, 2])
    assert torch.equal(direct, lat)
    with pytest.raises(ValueError, match="folder"):
        P.main(["--dataset", "kitti", "--config", "model_10_26"])
    with pytest.raises(ValueError, match="Invalid dataset name"):
        P.main(["--dataset", "nope", "--config", "model_10_26"])


def test_two_transformers_share_a_context():
    """A.forward, B.forward, A.forward on the default context: each call computes with its own module's weights
    (the Transformer slot of a context holds one model; the modules re-upload when the slot changed hands)."""
    from sd_video_gen_amd import config as svg_config
    from sd_video_gen_amd.transformer import Transformer
    from sd_video_gen_amd.transformer_text import Transformer as TextTransformer
    svg_config.set_args(["--dataset", "ball", "--config", "model_10_26"])
    torch.manual_seed(1)
    a = Transformer(dim_model=64, num_heads=4, num_encoder_layers=1, num_decoder_layers=1).eval()
    torch.manual_seed(2)
    b = Transformer(dim_model=64, num_heads=4, num_encoder_layers=1, num_decoder_layers=1).eval()
    torch.manual_seed(3)
    c = TextTransformer(dim_model=64, num_heads=8, num_encoder_layers=1, num_decoder_layers=1, st_weights="synthetic").eval()
    X = torch.randn(1, 5, 256)
    mask = a.get_tgt_mask(5)
    ra = TO.forward(a.state_dict(), X, X, 4, mask)
    rb = TO.forward(b.state_dict(), X, X, 4, mask)
    txt = c.encode_classes(["Archery"]).cpu()
    rc = TO.forward(c.state_dict(), X, X, 8, mask, txt=txt)
    assert rel_l2(ra, rb) > 1e-2
    xc, mc = X.cuda(), mask.cuda()
    for _ in range(2):
        assert rel_l2(a(xc, xc, mc).cpu(), ra) < 2e-5
        assert rel_l2(b(xc, xc, mc).cpu(), rb) < 2e-5
        assert rel_l2(c(xc, ["Archery"], xc, mc).cpu(), rc) < 2e-5
    assert rel_l2(a(xc, xc, mc).cpu(), ra) < 2e-5


def test_two_sdutils_do_not_share_weights(tiny_nets):
    """a second SDUtils while the first is alive gets its own library context; handing both the SAME context makes the
    first one fail loudly instead of computing with the second's weights."""
    from sd_video_gen_amd.sd_utils import SDUtils
    vsd, usd = tiny_nets
    vsd2 = SO.seeded_weights(SO.vae_shapes(VCFG), 13)
    _args(denoise=False)
    arch = {"vae": VCFG}
    g = torch.Generator().manual_seed(0)
    img = torch.randint(0, 256, (1, 64, 64, 3), dtype=torch.uint8, generator=g)
    eps = torch.randn(1, 4, 8, 8, generator=g)
    s1 = SDUtils(weights={"vae": vsd}, arch=arch, verbose=False)
    z1 = s1.encode_img(img, eps=eps.cuda())
    s2 = SDUtils(weights={"vae": vsd2}, arch=arch, verbose=False)
    assert s2.ctx is not s1.ctx
    z2 = s2.encode_img(img, eps=eps.cuda())
    assert torch.equal(s1.encode_img(img, eps=eps.cuda()), z1)                 # still its own weights
    assert rel_l2(z1.cpu(), SO.encode_img(vsd, img, eps, VCFG)) < 3e-2 and rel_l2(z2.cpu(), SO.encode_img(vsd2, img, eps, VCFG)) < 3e-2
    assert rel_l2(z1.cpu(), z2.cpu()) > 1e-2
    s3 = SDUtils(weights={"vae": vsd2}, arch=arch, verbose=False, ctx=s1.ctx)  # explicit sharing: the slot changes hands
    assert torch.equal(s3.encode_img(img, eps=eps.cuda()), z2)
    with pytest.raises(RuntimeError, match="another model"):
        s1.encode_img(img, eps=eps.cuda())


def test_error_codes_map_to_exceptions(ctx):
    """SVG_ERR_INVALID (-2) -> ValueError, SVG_ERR_RUNTIME (-1) -> RuntimeError; decided by the code, not the message."""
    x = torch.zeros(1, 4, 12, 12, device="cuda")
    ctx.configure(_lib.SVG_UNET, block_out=[64, 128], layers=1, heads=4, ctx_dim=64, groups=32, attn=[1, 0])
    with pytest.raises(ValueError, match="missing weight"):
        ctx.finalize(_lib.SVG_UNET)
    rc = ctx.lib.svg_unet_forward(ctx.h, x.data_ptr(), 1, 12, 12, x.data_ptr(), x.data_ptr(), 7, x.data_ptr(), None)
    assert rc == _lib.SVG_ERR_INVALID and b"finalize" in ctx.lib.svg_last_error(ctx.h)
    h = _lib.C.c_void_p()
    assert ctx.lib.svg_create(99, _lib.C.byref(h)) == _lib.SVG_ERR_INVALID


def test_bench_spawns_its_ranks():
    """`python bench.py --gpus 2` with no launcher: the script starts torch.distributed.run itself (before any GPU call in
    the parent) and rank 0 prints ONE JSON line with n_gpus 2.  Both ranks share the box's single GPU here (gloo + device
    override: the rehearsal knobs), tiny step (1 clip per rank, last DDIM step only).  An optional pass that FAILS on rank 0 after
    the headline (the $SVG_BENCH_FAIL_EXTRA hook stands in for extras.fp8_same_box raising, as in BENCH_r05) still leaves the
    line and exit code 0, with the failure recorded under its own key."""
    env = dict(os.environ, SVG_DIST_BACKEND="gloo", SVG_DEVICE_OVERRIDE="0", SVG_BENCH_FAIL_EXTRA="1")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--clips", "1",
                        "--streams", "1", "--start_step", "49", "--no-cpu-baseline", "--no-roofline"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_clips"] == 2 and d["value"] > 0 and d["scaling"] == "weak"
    assert "SVG_BENCH_FAIL_EXTRA" in d["extras"]["fp8_same_box"]["error"]


def test_rccl_gathers_device_tensors_in_a_one_rank_group():
    """The N > 1 path hands DEVICE tensors straight to `all_gather` under RCCL (sharding.gather_clips_packed: latents f32 + frames u8
    packed as bytes).  A one-GPU box cannot host two RCCL ranks, but a one-rank "nccl" group still goes through RCCL's init
    (`device_id=`), its uint8 all_gather and the packing / trimming code: run in a child process (a process group per pytest process
    would outlive the test)."""
    code = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from sd_video_gen_amd import sharding
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
assert dist.get_backend() == "nccl"
sharding._FORCE_COLLECTIVE = True
g = torch.Generator().manual_seed(0)
lat = torch.randn(3, 6, 256, generator=g).cuda()
frames = torch.randint(0, 256, (3, 2, 16, 16, 3), generator=g, dtype=torch.uint8).cuda()
a, b = sharding.gather_clips_packed([lat, frames], 3)
assert a.is_cuda and b.is_cuda and torch.equal(a, lat) and torch.equal(b, frames)
t = torch.ones(4, device="cuda")
dist.all_reduce(t)
assert float(t.sum()) == 4.0
dist.barrier()
dist.destroy_process_group()
print("rccl one-rank ok")
""" % ROOT
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29631", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "rccl one-rank ok" in r.stdout, (r.stdout + r.stderr)[-2000:]
