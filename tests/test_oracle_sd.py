"""CPU: what pins the SD-side oracle (oracle/sd_oracle.py) in the absence of reference fixtures
(the reference has no tests and no weights are reachable offline):
exact SD v1.4 parameter counts, DDIM closed forms, and structural identities of the wrapper."""
import math
import os
import sys

import numpy as np
import pytest
import torch

from conftest import rel_l2

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import sd_oracle as SO  # noqa: E402

TINY_UNET = dict(block_out=(64, 128), layers=1, heads=4, ctx_dim=64, groups=32, in_ch=4, out_ch=4, attn=(1, 0))
TINY_VAE = dict(block_out=(64, 128), layers=1, groups=32, latent=4)


def test_sd_v14_parameter_counts():
    assert SO.count(SO.unet_shapes()) == 859_520_964
    vs = SO.vae_shapes()
    assert SO.count(vs) == 83_653_863
    enc = SO.count({k: v for k, v in vs.items() if k.startswith("encoder.")})
    dec = SO.count({k: v for k, v in vs.items() if k.startswith("decoder.")})
    assert enc == 34_163_592 and dec == 49_490_179            # + quant_conv 72 + post_quant_conv 20
    assert enc + 72 == 34_163_664 and dec + 20 == 49_490_199  # SURVEY appendix A.3 groups them this way


def test_ddim_table_and_schedule():
    s = SO.DDIM(50)
    assert list(s.timesteps[:3]) == [980, 960, 940] and s.timesteps[-1] == 0 and len(s.timesteps) == 50
    assert abs(float(s.alphas_cumprod[0]) - 0.99915) < 1e-6
    assert abs(float(s.alphas_cumprod[999]) - 0.00466) < 1e-5
    x = torch.randn(2, 4, 8, 8)
    e = torch.randn(2, 4, 8, 8)
    a_t, a_p = float(s.alphas_cumprod[500]), float(s.alphas_cumprod[480])
    x0 = ((x - math.sqrt(1 - a_t) * e) / math.sqrt(a_t)).clamp(-1, 1)
    assert torch.allclose(s.step(e, 500, x), math.sqrt(a_p) * x0 + math.sqrt(1 - a_p) * e)
    # last step: prev < 0 -> alpha_prev = 1 -> returns the clipped x0
    a0 = float(s.alphas_cumprod[0])
    assert torch.allclose(s.step(e, 0, x), ((x - math.sqrt(1 - a0) * e) / math.sqrt(a0)).clamp(-1, 1))


def test_gen_i2i_identities():
    """start_step=50 -> zero UNet calls; guidance 0 -> the cond half never matters (sd_utils.py:256-257)."""
    lat = torch.randn(1, 4, 8, 8)
    calls = []

    def fake_unet(x, t, c):
        calls.append(t)
        return torch.cat([x[:1] * 0.1, torch.full_like(x[:1], 1e6)])    # cond half is garbage
    emb = torch.zeros(2, 7, 64)
    out = SO.gen_i2i_latents(None, emb, lat, 50, 0.0, 0, unet=fake_unet)
    assert len(calls) == 50 and calls[0] == 980 and calls[-1] == 0 and torch.isfinite(out).all()
    calls.clear()
    out = SO.gen_i2i_latents(None, emb, lat, 50, 0.0, 48, noise=torch.zeros_like(lat), unet=fake_unet)
    assert calls == [20, 0]
    hist = SO.gen_i2i_latents(None, emb, lat, 50, 0.0, 45, noise=torch.zeros_like(lat), unet=fake_unet, return_all_latents=True)
    assert hist.shape[0] == 6


def test_tiny_networks_run_and_are_finite():
    usd = SO.seeded_weights(SO.unet_shapes(TINY_UNET), 1)
    x = torch.randn(2, 4, 16, 16)
    ctx = torch.randn(2, 7, 64)
    e = SO.unet_forward(usd, x, 500, ctx, TINY_UNET)
    assert e.shape == x.shape and torch.isfinite(e).all() and e.std() > 1e-3
    # batch rows are independent: guidance-0 batch-1 == first half of the duplicated batch
    e1 = SO.unet_forward(usd, x[:1], 500, ctx[:1], TINY_UNET)
    assert rel_l2(e1, e[:1]) < 1e-5
    vsd = SO.seeded_weights(SO.vae_shapes(TINY_VAE), 2)
    img = torch.randint(0, 256, (2, 32, 32, 3), dtype=torch.uint8)
    z = SO.encode_img(vsd, img, torch.randn(2, 4, 16, 16), TINY_VAE)
    assert z.shape == (2, 4, 16, 16)
    out = SO.decode_img_latents(vsd, z, TINY_VAE)
    assert out.shape == (2, 32, 32, 3) and out.dtype == torch.uint8


def test_resize_nearest_matches_integer_rule():
    img = torch.randint(0, 256, (1, 64, 64, 3), dtype=torch.uint8)
    up = SO.resize_nearest_u8(img, 512, 512)
    assert torch.equal(up[0, ::8, ::8], img[0]) and torch.equal(up[0, 7::8, 7::8], img[0])
    assert torch.equal(SO.resize_nearest_u8(up, 64, 64), img)


def test_seeded_weights_reproducible_and_scaled():
    a = SO.seeded_weights(SO.vae_shapes(TINY_VAE), 3)
    b = SO.seeded_weights(SO.vae_shapes(TINY_VAE), 3)
    assert all(torch.equal(a[k], b[k]) for k in a)
    w = a["encoder.down_blocks.1.resnets.0.conv1.weight"]
    assert abs(float(w.std()) * math.sqrt(np.prod(w.shape[1:])) - 0.6) < 0.05
