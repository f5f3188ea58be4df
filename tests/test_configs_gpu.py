"""GPU: BASELINE.json's configs at FULL size and FULL length against the CPU oracle.

The long oracle runs (50 DDIM steps; 8- and 16-frame autoregressive rollouts with the denoise round trip) were made once
with oracle/gen_golden_sd.py and are kept as small fixtures (tests/golden/sd_*.pt): these tests rebuild the same seeded
weights, clip and CPU-generator noise, run the HIP path through the C ABI, and compare.  Every measured error is reported
through conftest.margin (printed, and collected into gpurun_out/parity_margins.json -> profiles/).

  configs[1]  1_19_ball_complex_L1_64, F=64, 8 predicted frames, --denoise --denoise_start_step 25      sd_cfg1_rollout.pt
  configs[2]  1_16_kitti_L1_64, F=64, 50-step DDIM at 64x64 latents, 512x512 VAE passes                 sd_cfg2_frame.pt
  configs[3]  11_27_ucf_final, F=128, 16 predicted frames (start step 48: 2 of the 50 steps per frame)  sd_cfg3_rollout.pt
              and at full length (all 50 steps per frame, non-chaotic weights)                          sd_cfg3_full_contractive.pt
  configs[4]  11_27_ucf_text_final: d = 2432 text-conditioned Transformer; guidance_scale 7.5 => the batch-2 UNet call of
              evaluation/predict_fvd2_denoise.py:227-229 is genuinely needed (in-test oracle, a few UNet calls)
              and at the full DDIM length (4 frames x 50 steps, batch-2 calls, non-chaotic weights)      sd_cfg4_text_guided_contractive.pt
Every fixture test runs in both storage modes: fp16 (SDUtils' default: the reference's autocast arithmetic) and bf16.  Tolerances are
<= 3x what was measured on MI355X in that mode (f32 accumulation, against the fp32 oracle); free-running 50-step comparisons are
asserted on the non-chaotic weights (test_config2_free_running_contractive, test_config3_full_length_contractive).
"""
import os
import sys

import pytest
import torch

from conftest import margin, rel_l2, sd_tol

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import gen_golden_sd as GG, sd_oracle as SO, transformer_oracle as TO  # noqa: E402
from sd_video_gen_amd import _lib  # noqa: E402

pytestmark = pytest.mark.gpu
FORCED_TOL = 4.5e-2  # one DDIM step from the oracle's latent: measured 1.5e-2 worst (bf16 UNet call 1e-2, amplified by 1/sqrt(alpha_t))
# The free-running 50-step comparison is ASSERTED in the non-chaotic weight regime (test_config2_free_running_contractive);
# on the chaotic unscaled network the free-running distances are printed only (they measure conditioning, not arithmetic).
GOLD = os.path.join(ROOT, "tests", "golden")


def gold(name):
    p = os.path.join(GOLD, name)
    if not os.path.exists(p):
        pytest.skip("fixture %s not generated (python oracle/gen_golden_sd.py)" % name)
    return torch.load(p, weights_only=False)


@pytest.fixture(scope="module", autouse=True)
def _threads():
    n = torch.get_num_threads()
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    yield
    torch.set_num_threads(n)


@pytest.fixture(scope="module")
def nets():
    """the seeded full-size SD v1.4 networks of the fixtures (UNet seed 31, VAE seed 32)"""
    usd = SO.seeded_weights(SO.unet_shapes(), GG.UNET_SEED)
    vsd = SO.seeded_weights(SO.vae_shapes(), GG.VAE_SEED)
    return usd, vsd


def _sdu(cfg_name, nets, dtype="bf16"):
    """dtype 'fp8' = what `bench.py --dtype fp8` / `--workload cfg4` builds: SDUtils(fp8=True, dtype='fp16') — MX-fp8 (e4m3 + E8M0
    block scales) operands on the resnets' / upsamplers' 3x3 convs, fp16 storage everywhere else (BASELINE configs[4]'s arithmetic)."""
    from sd_video_gen_amd import config as svg_config
    from sd_video_gen_amd.sd_utils import SDUtils
    svg_config.set_args(["--dataset", "synthetic-ball", "--config", cfg_name, "--denoise", "1"])
    usd, vsd = nets
    fp8 = dtype == "fp8"
    return SDUtils(weights={"vae": vsd, "unet": usd, "text_encoder": "synthetic"}, verbose=False, dtype="fp16" if fp8 else dtype, fp8=fp8)


class _arith:
    """fp8 legs: the MX-fp8 conv kernel takes a conv only when it fills the chip (>= 192 workgroups: the bench's 28-clip groups do, the
    fixtures' single clip does not and would silently run fp16).  The kernel and its arithmetic do not depend on the batch, so the fp8
    legs lower the threshold ($SVG_HALO_MIN=1) — and check_fp8() proves from the library's own launch records that the e4m3 kernels ran."""

    def __init__(self, dtype):
        self.on = dtype == "fp8"

    def __enter__(self):
        if self.on:
            self.old = os.environ.get("SVG_HALO_MIN")
            os.environ["SVG_HALO_MIN"] = "1"
            _lib.env_refresh()
        return self

    def __exit__(self, *a):
        if self.on:
            if self.old is None:
                os.environ.pop("SVG_HALO_MIN", None)
            else:
                os.environ["SVG_HALO_MIN"] = self.old
            _lib.env_refresh()

    def check_fp8(self, c, batch):
        """one more UNet call of the same batch under the per-site profile: the resnets' convs must have launched conv_halo_fp8"""
        if not self.on:
            return
        c.prof_enable(True, detail=True)
        c.prof_reset()
        c.unet_forward(torch.zeros(batch, 4, 64, 64).cuda(), torch.full((batch,), 500.0).cuda(), torch.zeros(batch, 77, 768).cuda())
        torch.cuda.synchronize()
        rep = c.prof_report()
        c.prof_enable(False)
        n8 = sum(v["calls"] for k, v in rep.items() if "conv_fp8" in k)
        print("[parity] fp8 leg: %d of the UNet call's conv launches ran conv_halo_fp8 (e4m3 x e4m3, E8M0 block scales)" % n8)
        assert n8 >= 30, "the fp8 leg did not run the MX-fp8 conv kernel (%d launches)" % n8


def _rollout(cfg_name, g, nets, dtype="bf16"):
    from sd_video_gen_amd.predict import sample_clips, bouncing_ball_clips
    sdu = _sdu(cfg_name, nets, dtype)
    m, cfg = GG.build_transformer(cfg_name)
    clip = bouncing_ball_clips(1, cfg.FRAME_SIZE, 5, seed=GG.CLIP_SEED)
    lat = sample_clips(m, sdu, clip.cuda(), g["pred_frames"], denoise=True, start_step=g["start_step"], seeds=[GG.NOISE_SEED],
                       text_embeddings=GG.text_emb().cuda(), cpu_noise=True)
    assert lat.shape == g["all_latents"].shape and torch.isfinite(lat).all()
    return lat.cpu(), sdu


@pytest.mark.parametrize("dtype,tol_cond,tol_forced", [("fp16", 1.4e-3, 5.5e-3), ("bf16", 1.3e-3, FORCED_TOL)])   # measured fp16: 4.5e-4, 1.8e-3; bf16 (UNet; the VAE keeps fp16 storage since round 6): 4.2e-4, 1.45e-2
def test_config2_full_frame_50_steps(ctx, nets, dtype, tol_cond, tol_forced):
    """configs[2] end to end for one generated frame, and the 50-step DDIM loop step by step.

    What was measured on MI355X (profiles/r02_parity.md): the DDIM map of this seeded random-weight UNet is chaotic — a
    1e-3 (rel-L2) perturbation of the starting latent grows 16x in the first step and saturates near 0.55 within ten steps,
    HIP path against HIP path, exactly as the bf16-vs-fp32 difference does.  A free-running 50-step comparison therefore
    measures the conditioning of the (untrained) network, not the arithmetic.  The arithmetic is checked per step with
    teacher forcing: every one of the 50 steps starts from the ORACLE's latent z_k and must reproduce the oracle's z_{k+1};
    the free-running table is reported, its first step asserted, and its tail bounded by the saturation level."""
    g = gold("sd_cfg2_frame.pt")
    lat, sdu = _rollout("1_16_kitti_L1_64", g, nets, dtype)
    e_cond = rel_l2(lat[:, :4], g["all_latents"][:, :4])
    e_frame = rel_l2(lat[:, 4:], g["all_latents"][:, 4:])
    emb = GG.text_emb().cuda()
    hist_ref = g["hist"]
    assert g["hist_steps"] == list(range(51)) and hist_ref.shape == (51, 4, 64, 64)
    # ---- teacher-forced: all 50 steps in ONE batch-50 UNet call (per-sample timesteps 980, 960, ..., 0), then the scheduler
    c = sdu.unet.ctx
    ts = torch.tensor([980.0 - 20.0 * k for k in range(50)])
    eps = c.unet_forward(hist_ref[:50].cuda(), ts.cuda(), emb[:1].repeat(50, 1, 1))          # guidance 0: the uncond half
    forced = []
    for k in range(50):
        t = int(ts[k])
        z1 = c.ddim_step(hist_ref[k:k + 1].cuda(), eps[k:k + 1], t, t - 20).cpu()
        forced.append(rel_l2(z1, hist_ref[k + 1:k + 2]))
    print("[parity] teacher-forced DDIM step k -> k+1 (oracle z_k in), rel-L2 of z_{k+1}: " + "  ".join("%d:%.1e" % (k, e) for k, e in enumerate(forced)))
    # ---- free-running from the oracle's starting latent, and the growth of a 1e-3 perturbation through the same HIP loop
    hist = c.ddim_loop(g["lat0"].cuda(), emb, num_steps=50, start_step=0, guidance=0.0, return_hist=True).cpu()
    assert hist.shape[0] == 51
    ks = [0, 1, 2, 3, 5, 10, 15, 20, 30, 40, 50]
    table = [(k, rel_l2(hist[k], hist_ref[k])) for k in ks]
    print("[parity] free-running DDIM drift vs the fp32 oracle after k steps: " + "  ".join("k=%d: %.2e" % t for t in table))
    gp = torch.Generator().manual_seed(99)
    d = torch.randn(g["lat0"].shape, generator=gp)
    pert = g["lat0"] + 1e-3 * d * (g["lat0"].norm() / d.norm())
    hist_p = c.ddim_loop(pert.cuda(), emb, num_steps=50, start_step=0, guidance=0.0, return_hist=True).cpu()
    sens = [(k, rel_l2(hist_p[k], hist[k])) for k in ks]
    print("[parity] growth of a 1e-3 input perturbation through the same loop (HIP vs HIP): " + "  ".join("k=%d: %.2e" % t for t in sens))
    import json
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "ddim_drift.json"), "w") as f:
            json.dump({"teacher_forced_rel_l2_per_step": forced, "steps": ks, "free_running_rel_l2_vs_oracle": [t[1] for t in table],
                       "growth_of_1e-3_perturbation_hip_vs_hip": [t[1] for t in sens]}, f)
    except OSError:
        pass
    margin("cfg2 conditioning latents (VAE encode @64, %s)" % dtype, e_cond, tol_cond)
    margin("cfg2 DDIM step, teacher-forced, worst of the 50 steps (%s)" % dtype, max(forced), tol_forced)
    margin("cfg2 DDIM free-running, after the first step (%s)" % dtype, table[1][1], tol_forced)
    print("[parity] cfg2 on the CHAOTIC unscaled network (printed, not asserted): free-running after 50 steps %.3f, generated frame %.3f"
          % (table[-1][1], e_frame))


@pytest.mark.parametrize("dtype,tol_loop,tol_frame", [("fp16", 5e-3, 1.5e-2), ("bf16", 1.5e-2, 1.8e-2), ("fp8", 1.2e-1, 1.2e-1)])   # bf16 UNet + fp16 VAE (round 6) measured 4.9e-3 / 5.7e-3 (all-bf16 was 5e-3 / 1.9e-2)
def test_config2_free_running_contractive(ctx, nets, dtype, tol_loop, tol_frame):
    """configs[2] END TO END, free-running, at a real tolerance: the headline workload (one generated frame = VAE passes at 512 x 512 +
    all 50 DDIM steps of the full-size UNet, nothing teacher-forced) against the fp32 oracle, in the synthetic-weight regime whose
    DDIM map is not chaotic (oracle/gen_golden_sd.py: the seeded UNet with conv_out x 0.1; a 1e-3 perturbation grows < 3x over the
    50 steps — asserted below HIP against HIP — while the network still moves the final latent by 0.28 rel-L2).
    Tolerances: the loop's final latent <= 5e-3 in fp16 storage (the reference's arithmetic) and <= 3e-2 in bf16; the generated
    frame's latent additionally passes three uint8 quantisations (decode @512 -> encode @64), stated separately."""
    g = gold("sd_cfg2_contractive.pt")
    usd, vsd = nets
    cnets = (GG.contractive_unet(usd), vsd)
    with _arith(dtype) as ar:
        _config2_free_running(g, cnets, dtype, tol_loop, tol_frame, ar)


def _config2_free_running(g, cnets, dtype, tol_loop, tol_frame, ar):
    lat, sdu = _rollout("1_16_kitti_L1_64", g, cnets, dtype)
    ar.check_fp8(sdu.unet.ctx, 1)
    store = "fp16" if dtype == "fp8" else dtype
    # (the VAE keeps fp16 storage under a bf16 UNet: its decoder ends in a uint8 image — SDUtils)
    assert sdu.ctx.model_dtype(_lib.SVG_UNET) == store and sdu.ctx.model_dtype(_lib.SVG_VAE) == "fp16" and sdu.fp8 == (dtype == "fp8")
    e_frame = rel_l2(lat[:, 4:], g["all_latents"][:, 4:])
    emb = GG.text_emb().cuda()
    ks = g["hist_steps"]
    assert ks[-1] == 50
    c = sdu.unet.ctx
    hist = c.ddim_loop(g["lat0"].cuda(), emb, num_steps=50, start_step=0, guidance=0.0, return_hist=True).cpu()
    table = [(k, rel_l2(hist[k], g["hist"][i])) for i, k in enumerate(ks)]
    print("[parity] %s free-running DDIM (contractive regime) vs the fp32 oracle after k steps: " % dtype + "  ".join("k=%d: %.2e" % t for t in table))
    gp = torch.Generator().manual_seed(99)
    d = torch.randn(g["lat0"].shape, generator=gp)
    pert = g["lat0"] + 1e-3 * d * (g["lat0"].norm() / d.norm())
    hist_p = c.ddim_loop(pert.cuda(), emb, num_steps=50, start_step=0, guidance=0.0, return_hist=True).cpu()
    growth = max(rel_l2(hist_p[k], hist[k]) for k in range(1, 51)) / 1e-3
    if dtype == "fp16":   # (bf16's own rounding noise, 5e-3, is above the 1e-3 perturbation: the growth is a property of the map, measured in fp16)
        margin("cfg2 contractive regime (fp16): growth of a 1e-3 perturbation over the 50 steps, HIP vs HIP", growth, 3.0, unit="x")
    else:
        print("[parity] bf16: distance of the perturbed run / 1e-3 = %.2f (includes bf16's own 5e-3 rounding noise)" % growth)
    # the network matters in this regime: zeroing eps (the scheduler alone) lands far from the result
    ac = torch.cumprod(1 - torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000) ** 2, 0)
    z = g["lat0"].clone()
    for i in range(50):
        t = 980 - 20 * i
        z = (ac[t - 20] if t >= 20 else torch.tensor(1.0)).sqrt() * (z / ac[t].sqrt()).clamp(-1, 1)
    assert rel_l2(g["hist"][-1], z) > 0.2, "the regime must keep the UNet relevant to the result"
    margin("cfg2 contractive regime (%s): FREE-RUNNING latent after all 50 DDIM steps" % dtype, table[-1][1], tol_loop)
    margin("cfg2 contractive regime (%s): worst step of the free-running history" % dtype, max(t[1] for t in table), tol_loop)
    margin("cfg2 contractive regime (%s): generated frame latent (50 steps + VAE @512 + 3 uint8 round trips)" % dtype, e_frame, tol_frame)


@pytest.mark.parametrize("dtype,tol_all,tol_worst", [("fp16", 1e-2, 1.1e-2), ("bf16", 2.1e-2, 2.3e-2)])   # measured fp16: 3.3e-3, 3.7e-3; bf16 UNet + fp16 VAE (round 6): 6.8e-3, 7.7e-3 (all-bf16: 1.07e-2, 1.24e-2)
def test_config1_rollout_8_frames_start25(ctx, nets, dtype, tol_all, tol_worst):
    """configs[1]: 8 autoregressive frames, 25 DDIM steps each (200 UNet calls in the oracle fixture)."""
    g = gold("sd_cfg1_rollout.pt")
    lat, _ = _rollout("1_19_ball_complex_L1_64", g, nets, dtype)
    for k in range(g["pred_frames"]):
        print("[parity] cfg1 frame %d rel-L2 %.3e" % (k, rel_l2(lat[:, 4 + k], g["all_latents"][:, 4 + k])))
    # start step 25 enters the loop at t = 480, past the early steps (t >= 800, division by sqrt(alpha_t) <= 0.2) that make the
    # 50-step loop of test_config2_full_frame_50_steps chaotic: 25 free-running steps + three uint8 round trips per frame,
    # eight frames deep, stay at the single-call error (measured 1.07e-2 overall, 0.99e-2 .. 1.30e-2 per frame)
    margin("cfg1 8-frame rollout (25 steps / frame, %s), all generated latents" % dtype, rel_l2(lat[:, 4:], g["all_latents"][:, 4:]), tol_all)
    margin("cfg1 8-frame rollout (%s), worst frame" % dtype, max(rel_l2(lat[:, 4 + k], g["all_latents"][:, 4 + k]) for k in range(g["pred_frames"])), tol_worst)


@pytest.mark.parametrize("dtype,tol_first", [("fp16", 2e-2), ("bf16", 2.1e-2)])   # measured fp16: 6.5e-3; bf16 UNet + fp16 VAE (round 6): 6.8e-3 (all-bf16: 2.6e-2)
def test_config3_rollout_16_frames_f128(ctx, nets, dtype, tol_first):
    """configs[3]: 11_27_ucf_final (F=128, D_lat=1024), 16 autoregressive frames with the 512x512 round trip, on the CHAOTIC
    (unscaled) synthetic weights.  Only the first generated frame is asserted: from the second frame on the distance measures how
    the random-weight networks amplify a rounding through the feedback loop (fp16 5.6e-2 over all frames, 9.4e-2 at the sixteenth;
    HIP against HIP under a 1e-3 perturbation grows the same way), not arithmetic — those figures are printed, and the full-length
    ASSERTED check of this config is test_config3_full_length_contractive below."""
    g = gold("sd_cfg3_rollout.pt")
    lat, _ = _rollout("11_27_ucf_final", g, nets, dtype)
    for k in (0, 7, 15):
        print("[parity] cfg3 frame %d rel-L2 %.3e" % (k, rel_l2(lat[:, 4 + k], g["all_latents"][:, 4 + k])))
    margin("cfg3 16-frame rollout (%s), first generated frame" % dtype, rel_l2(lat[:, 4], g["all_latents"][:, 4]), tol_first)
    print("[parity] cfg3 16-frame rollout (%s), chaotic weights, printed only: all generated latents %.3e, last frame %.3e"
          % (dtype, rel_l2(lat[:, 4:], g["all_latents"][:, 4:]), rel_l2(lat[:, -1], g["all_latents"][:, -1])))
    assert torch.isfinite(lat).all()


@pytest.mark.parametrize("dtype,tol_first,tol_all,tol_last", [("fp16", 1.5e-2, 4e-2, 6e-2), ("bf16", 2.1e-2, 9e-2, 1.4e-1)])   # measured fp16: 5.1e-3, 1.7e-2, 2.4e-2; bf16 UNet + fp16 VAE (round 6): 6.8e-3, 2.9e-2, 4.6e-2 (all-bf16: 1.6e-2, 3.7e-2, 5.3e-2)
def test_config3_full_length_contractive(ctx, nets, dtype, tol_first, tol_all, tol_last):
    """configs[3] at its FULL length: 11_27_ucf_final, 16 autoregressive frames, each with all 50 DDIM steps of the full-size UNet
    between the 512 x 512 VAE passes (800 UNet calls in the oracle fixture, oracle/gen_golden_sd.py cfg3c), free-running, in the
    non-chaotic weight regime of test_config2_free_running_contractive."""
    g = gold("sd_cfg3_full_contractive.pt")
    assert g["pred_frames"] == 16 and g["start_step"] == 0 and g["unet_calls"] == 800
    usd, vsd = nets
    lat, sdu = _rollout("11_27_ucf_final", g, (GG.contractive_unet(usd), vsd), dtype)
    assert sdu.ctx.model_dtype(_lib.SVG_UNET) == dtype
    per = [rel_l2(lat[:, 4 + k], g["all_latents"][:, 4 + k]) for k in range(16)]
    print("[parity] cfg3 full length (%s) per-frame rel-L2: " % dtype + " ".join("%.2e" % e for e in per))
    margin("cfg3 FULL length (16 frames x 50 steps, %s): first generated frame" % dtype, per[0], tol_first)
    margin("cfg3 FULL length (%s): all generated latents" % dtype, rel_l2(lat[:, 4:], g["all_latents"][:, 4:]), tol_all)
    margin("cfg3 FULL length (%s): last frame (16 autoregressive steps)" % dtype, per[-1], tol_last)


@pytest.mark.parametrize("dtype,tol_first,tol_all", [("fp16", 2e-2, 2.9e-2), ("bf16", 1.2e-1, 1.6e-1), ("fp8", 1e-1, 1e-1)])   # measured fp16: 7.5e-3, 9.6e-3; bf16: 5.3e-2, 5.6e-2 (guidance 7.5 amplifies the per-call error); fp8: profiles/r05_parity.md
def test_config4_full_ddim_length_text_guided(ctx, nets, dtype, tol_first, tol_all):
    """configs[4] at the full DDIM length: 11_27_ucf_text_final (text-conditioned Transformer, d = 2432), guidance_scale 7.5 with
    distinct uncond / cond embeddings, four autoregressive frames of 50 steps each (200 batch-2 UNet calls in the oracle fixture,
    oracle/gen_golden_sd.py cfg4c), free-running on the non-chaotic weights.
    fp8 leg = the arithmetic configs[4] names, built exactly as `bench.py --workload cfg4` builds it (SDUtils(fp8=True,
    dtype='fp16')): under guidance the library's DDIM loop keeps e4m3 operands on the 16 x 16 level's convs only (csrc/unet.cpp
    kFp8SitesGuided; per-site study profiles/r05_fp8_sites.txt) — asserted <= 1e-1; the same loop with EVERY eligible conv in e4m3
    ($SVG_FP8_SITES_GUIDED=4095, round 4's placement) is run too and printed (1.8e-1: why it is not the default)."""
    from sd_video_gen_amd.predict import sample_clips, bouncing_ball_clips
    g = gold("sd_cfg4_text_guided_contractive.pt")
    assert g["pred_frames"] == 4 and g["start_step"] == 0 and g["guidance_scale"] == 7.5 and g["unet_calls"] == 200
    usd, vsd = nets
    m, cfg = GG.build_text_transformer()

    def run(ar):
        sdu = _sdu("11_27_ucf_text_final", (GG.contractive_unet(usd), vsd), dtype)
        clip = bouncing_ball_clips(1, cfg.FRAME_SIZE, 5, seed=GG.CLIP_SEED)
        lat = sample_clips(m, sdu, clip.cuda(), 4, denoise=True, start_step=0, seeds=[GG.NOISE_SEED], text_embeddings=GG.text_emb_pair().cuda(),
                           guidance_scale=7.5, cls_list=[g["class"]], cpu_noise=True).cpu()
        ar.check_fp8(sdu.unet.ctx, 2)
        assert lat.shape == g["all_latents"].shape
        return [rel_l2(lat[:, 4 + k], g["all_latents"][:, 4 + k]) for k in range(4)], rel_l2(lat[:, 4:], g["all_latents"][:, 4:])

    with _arith(dtype) as ar:
        per, e_all = run(ar)
        print("[parity] cfg4 full DDIM length (%s) per-frame rel-L2: " % dtype + " ".join("%.2e" % e for e in per))
        margin("cfg4 text + guidance 7.5, 50 steps per frame (%s): first generated frame" % dtype, per[0], tol_first)
        margin("cfg4 text + guidance 7.5, 50 steps per frame (%s): all four generated frames" % dtype, e_all, tol_all)
        if dtype == "fp8":
            os.environ["SVG_FP8_SITES_GUIDED"] = "4095"
            _lib.env_refresh()
            try:
                per_all, e_all_sites = run(ar)
            finally:
                os.environ.pop("SVG_FP8_SITES_GUIDED", None)
                _lib.env_refresh()
            print("[parity] cfg4 full DDIM length, EVERY eligible conv in e4m3 under guidance (not the default) per-frame rel-L2: " + " ".join("%.2e" % e for e in per_all))
            margin("cfg4 text + guidance 7.5 (fp8, all 33 eligible convs e4m3 — round 4's placement, printed for the record): all four frames", e_all_sites, 3e-1)
            assert e_all_sites > e_all, "the guided placement must be the more accurate one"


@pytest.mark.parametrize("fp8,tol_call,tol_guided,tol_steps", [(0, 2.5e-2, 2e-1, 1.5e-2), (1, 9e-2, 6e-1, 2.5e-2)])   # fp8 = 1 measured: 8.4e-2 (all 33 e4m3 convs: a direct call), 4.1e-1, 7.4e-3 (the guided loop keeps the 16 x 16 level only)
def test_config4_guidance_7p5_full_size(ctx, nets, fp8, tol_call, tol_guided, tol_steps):
    """configs[4]: a real prompt + guidance_scale 7.5 (evaluation/predict_fvd2_denoise.py:203,227-229): the batch-2 UNet call
    with DIFFERENT uncond / cond embeddings, the CFG combine and three scheduler steps, full-size UNet.  fp8 = 1: the same rows in the
    arithmetic configs[4] names (MX-fp8 3x3 convs, fp16 storage elsewhere)."""
    with _arith("fp8" if fp8 else "fp16") as ar:
        _config4_guidance(ctx, nets, fp8, tol_call, tol_guided, tol_steps, ar)


def _config4_guidance(ctx, nets, fp8, tol_call, tol_guided, tol_steps, ar):
    usd, _ = nets
    c = SO.SD_UNET
    ctx.configure(_lib.SVG_UNET, block_out=list(c["block_out"]), layers=2, heads=8, ctx_dim=768, groups=32, attn=list(c["attn"]), fp8=fp8, f16=fp8)
    ctx.load_state_dict(_lib.SVG_UNET, usd)
    ctx.finalize(_lib.SVG_UNET)
    g = torch.Generator().manual_seed(11)
    z = torch.randn(1, 4, 64, 64, generator=g) * 0.8
    emb = torch.randn(2, 77, 768, generator=g)                     # [uncond; cond], distinct
    noise = torch.randn(1, 4, 64, 64, generator=g)
    x2 = torch.cat([z, z])
    e = ctx.unet_forward(x2.cuda(), torch.tensor([500.0, 500.0]).cuda(), emb.cuda()).cpu()
    ref = SO.unet_forward(usd, x2, 500, emb)
    tag = " [fp8 convs]" if fp8 else ""
    margin("cfg4 batch-2 UNet call (uncond/cond rows), full size" + tag, rel_l2(e, ref), tol_call)      # measured 1.05e-2 (bf16)
    margin("cfg4 guided noise u + 7.5 (c - u)" + tag, rel_l2(e[:1] + 7.5 * (e[1:] - e[:1]), ref[:1] + 7.5 * (ref[1:] - ref[:1])), tol_guided)
    S = 47
    got = ctx.ddim_loop(z.cuda(), emb.cuda(), num_steps=50, start_step=S, guidance=7.5, noise=noise.cuda()).cpu()
    want = SO.gen_i2i_latents(usd, emb, z, 50, 7.5, S, noise=noise)
    margin("cfg4 3 DDIM steps at guidance 7.5" + tag, rel_l2(got, want), tol_steps)                       # measured 5.5e-3 (bf16)
    ar.check_fp8(ctx, 2)


def test_config4_text_loop_with_prompt_and_guidance(ctx, nets):
    """configs[4] end to end at full size: 11_27_ucf_text_final (text-conditioned Transformer d = 2432, F = 128), the CLIP text
    encoder on the class prompt, guidance_scale 7.5 (evaluation/predict_fvd2_denoise.py:203,227-229): two generated frames with
    two DDIM steps each against the loop oracle driven with the same CLIP oracle embeddings."""
    from oracle import clip_oracle as CO, loop_oracle
    from sd_video_gen_amd import config as svg_config, sd_layout
    from sd_video_gen_amd.predict import sample_clips, bouncing_ball_clips
    from sd_video_gen_amd.sd_utils import SDUtils
    from sd_video_gen_amd.transformer_text import Transformer as TextTransformer
    usd, vsd = nets
    svg_config.set_args(["--dataset", "ucf", "--config", "11_27_ucf_text_final", "--denoise", "1"])
    cfg = svg_config.load_config("11_27_ucf_text_final")
    sdu = SDUtils(weights={"vae": vsd, "unet": usd, "text_encoder": "synthetic"}, verbose=False, seed=2)
    torch.manual_seed(9)
    m = TextTransformer(dim_model=cfg.DIM_MODEL[0], num_heads=cfg.NUM_HEADS[0], num_encoder_layers=cfg.NUM_ENCODER_LAYERS[0],
                        num_decoder_layers=cfg.NUM_DECODER_LAYERS[0], st_weights="synthetic").eval()
    names = ["WallPushups"]
    prompt = ["a person doing WallPushups"]
    emb = sdu.encode_text(prompt)                                   # (2,77,768) = [uncond(''); text(prompt)] from the HIP CLIP tower
    csd = {k: v.cpu() for k, v in sd_layout.seeded_weights(sd_layout.clip_text_shapes(), 2 + 3).items()}
    emb_ref = CO.encode_text(csd, prompt)
    margin("cfg4 CLIP embeddings of the prompt", rel_l2(emb.cpu(), emb_ref), 2.1e-6)
    clip = bouncing_ball_clips(1, cfg.FRAME_SIZE, 5, seed=GG.CLIP_SEED)
    S, N = 48, 2
    lat = sample_clips(m, sdu, clip.cuda(), N, denoise=True, start_step=S, seeds=[GG.NOISE_SEED], text_embeddings=emb,
                       guidance_scale=7.5, cls_list=names, cpu_noise=True).cpu()
    noise = GG.loop_noise(GG.NOISE_SEED, cfg.FRAME_SIZE, N, S)
    ref = loop_oracle.sample_clip({k: v for k, v in m.state_dict().items()}, cfg.NUM_HEADS[0], vsd, clip[0], N, noise, denoise=True, start_step=S,
                                  unet_sd=usd, text_emb=emb_ref, txt=m.encode_classes(names).cpu(), guidance_scale=7.5)
    assert lat.shape == ref.shape == (1, 4 + N, 1024)
    margin("cfg4 conditioning latents (VAE encode @128)", rel_l2(lat[:, :4], ref[:, :4]), sd_tol(1.4e-3, 1.5e-2))
    margin("cfg4 text + guidance 7.5 loop, generated frames", rel_l2(lat[:, 4:], ref[:, 4:]), sd_tol(1.6e-2, 1.0e-1))      # measured 5.5e-3 fp16 / 3.5e-2 bf16


def test_config4_text_transformer_full_size(ctx):
    """configs[4]: 11_27_ucf_text_final — d = 2048 + 384 = 2432 (head dim 304), D_lat 1024, 4 + 8 layers, 0.6 G parameters."""
    from sd_video_gen_amd import config as svg_config
    from sd_video_gen_amd.transformer_text import Transformer as TextTransformer, predict as predict_text
    svg_config.set_args(["--dataset", "ucf", "--config", "11_27_ucf_text_final"])
    cfg = svg_config.load_config("11_27_ucf_text_final")
    torch.manual_seed(4)
    m = TextTransformer(dim_model=cfg.DIM_MODEL[0], num_heads=cfg.NUM_HEADS[0], num_encoder_layers=cfg.NUM_ENCODER_LAYERS[0],
                        num_decoder_layers=cfg.NUM_DECODER_LAYERS[0], st_weights="synthetic").eval()
    assert m.dim_model == 2432 and m.d_lat == 1024
    sd = m.state_dict()
    X = torch.randn(2, 6, 1024)
    names = ["WallPushups", "PlayingGuitar"]
    txt = m.encode_classes(names).cpu()
    out = m(X.cuda(), names, X.cuda(), m.get_tgt_mask(6).cuda(), pe_row=torch.zeros(2, dtype=torch.int32)).cpu()
    for b in range(2):
        ref = TO.forward(sd, X[b:b + 1], X[b:b + 1], 8, TO.get_tgt_mask(6), txt=txt[b:b + 1])
        margin("cfg4 text Transformer d=2432 forward, clip %d" % b, rel_l2(out[:, b:b + 1], ref), 4.5e-6)
    p = predict_text(m, X[:1].cuda(), names[:1]).cpu()
    margin("cfg4 predict_text (D_lat,)", rel_l2(p, TO.predict(sd, X[:1], 8, txt=txt[:1])), 3.8e-6)
