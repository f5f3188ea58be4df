"""GPU: the CLIP text encoder of the path (utils/sd_utils.py:78-95 encode_text; svg_clip_text_forward) against the CPU
oracle, which tests/test_oracle_clip.py pins against the installed transformers CLIPTextModel."""
import json
import os
import sys

import pytest
import torch

from conftest import margin, rel_l2

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import clip_oracle as CO, sd_oracle as SO  # noqa: E402
from sd_video_gen_amd import _lib  # noqa: E402

pytestmark = pytest.mark.gpu
TINY = dict(vocab=1000, d_model=64, heads=4, layers=2, ffn=128, max_pos=77)
VCFG = dict(block_out=(64, 128, 128, 128), layers=1, groups=32, latent=4)
UCFG = dict(block_out=(64, 128), layers=1, heads=4, ctx_dim=768, groups=32, in_ch=4, out_ch=4, attn=(1, 0))


def _load(ctx, cfg, sd):
    ctx.configure(_lib.SVG_CLIP_TEXT, **cfg)
    ctx.load_state_dict(_lib.SVG_CLIP_TEXT, sd)
    return ctx.finalize(_lib.SVG_CLIP_TEXT)


@pytest.mark.parametrize("cfg,B,T", [(TINY, 1, 77), (TINY, 5, 77), (TINY, 3, 20), (CO.SD_CLIP, 2, 77)], ids=["tiny-b1", "tiny-b5-chunked", "tiny-T20", "vit-l-14-b2"])
def test_clip_text_forward(ctx, cfg, B, T):
    sd = SO.seeded_weights(CO.clip_text_shapes(cfg), 17)
    n = _load(ctx, cfg, sd)
    assert n == sum(v.numel() for v in sd.values())
    g = torch.Generator().manual_seed(B * 100 + T)
    ids = torch.randint(0, cfg["vocab"], (B, T), generator=g)
    out = ctx.clip_text_forward(ids, cfg["d_model"]).cpu()
    assert out.shape == (B, T, cfg["d_model"])
    margin("CLIP text tower %s B=%d T=%d (f32 MFMA vs fp32 oracle)" % ("ViT-L/14" if cfg is CO.SD_CLIP else "tiny", B, T),
           rel_l2(out, CO.forward(sd, ids, cfg)), 2.2e-6)
    # causal: a token's state does not depend on later tokens
    ids2 = ids.clone()
    ids2[:, T // 2:] = (ids2[:, T // 2:] + 1) % cfg["vocab"]
    out2 = ctx.clip_text_forward(ids2, cfg["d_model"]).cpu()
    assert torch.equal(out2[:, : T // 2], out[:, : T // 2]) and not torch.equal(out2[:, T // 2:], out[:, T // 2:])


def test_clip_rejects_bad_calls(ctx):
    _load(ctx, TINY, SO.seeded_weights(CO.clip_text_shapes(TINY), 1))
    with pytest.raises(ValueError):
        ctx.clip_text_forward(torch.zeros(1, 78, dtype=torch.long), 64)           # longer than max_position_embeddings
    ctx.configure(_lib.SVG_CLIP_TEXT, **TINY)
    sd = SO.seeded_weights(CO.clip_text_shapes(TINY), 1)
    sd.pop("encoder.layers.1.mlp.fc2.bias")
    ctx.load_state_dict(_lib.SVG_CLIP_TEXT, sd)
    with pytest.raises(ValueError, match="fc2.bias"):
        ctx.finalize(_lib.SVG_CLIP_TEXT)


def test_encode_text_matches_the_reference_algorithm(tmp_path, monkeypatch):
    """SDUtils.encode_text = tokenizer + text encoder for the prompts and for '' * n, cat([uncond, text]) — with the synthetic
    ViT-L/14 text weights of SDUtils(weights='synthetic') and with a local text_encoder/ directory in transformers' format."""
    from safetensors.torch import save_file
    from sd_video_gen_amd import config as svg_config, sd_layout
    from sd_video_gen_amd.sd_utils import SDUtils
    svg_config.set_args(["--dataset", "synthetic-ball", "--config", "model_10_26", "--denoise", "1"])
    monkeypatch.delenv("SVG_SD_WEIGHTS", raising=False)
    arch = {"vae": VCFG, "unet": UCFG}
    sdu = SDUtils(weights="synthetic", arch=arch, verbose=False, seed=5)
    assert sdu.clip_source == "synthetic" and sdu.text_encoder.n_params == 123_060_480 and sdu.tokenizer.model_max_length == 77
    csd = {k: v.cpu() for k, v in sd_layout.seeded_weights(sd_layout.clip_text_shapes(), 5 + 3).items()}
    prompts = ["a person doing WallPushups", "PlayingGuitar"]
    emb = sdu.encode_text(prompts)
    assert emb.shape == (4, 77, 768) and emb.is_cuda
    margin("encode_text(2 prompts) vs oracle", rel_l2(emb.cpu(), CO.encode_text(csd, prompts)), 2.1e-6)
    e0 = sdu.encode_text([""])                                                   # prediction/predict.py:148
    assert e0.shape == (2, 77, 768) and torch.equal(e0[0], e0[1])                # [uncond(''); text('')]
    margin("encode_text(['']) vs oracle", rel_l2(e0.cpu(), CO.encode_text(csd, [""])), 2.1e-6)
    del sdu
    # local directory, transformers 4.x naming ("text_model." prefix + the position_ids buffer), tiny architecture
    d = tmp_path / "sd"
    os.makedirs(d / "text_encoder")
    tsd = SO.seeded_weights(CO.clip_text_shapes(dict(TINY, d_model=768, heads=12, ffn=256)), 9)
    tcfg = dict(TINY, d_model=768, heads=12, ffn=256)
    save_file({"text_model." + k: v.contiguous() for k, v in tsd.items()} | {"text_model.embeddings.position_ids": torch.arange(77)[None].float()},
              str(d / "text_encoder" / "model.safetensors"))
    with open(d / "text_encoder" / "config.json", "w") as f:
        json.dump({"vocab_size": 1000, "hidden_size": 768, "num_attention_heads": 12, "num_hidden_layers": 2, "intermediate_size": 256,
                   "max_position_embeddings": 77, "hidden_act": "quick_gelu"}, f)
    monkeypatch.setenv("SVG_SD_WEIGHTS", str(d))
    vsd, usd = SO.seeded_weights(SO.vae_shapes(VCFG), 3), SO.seeded_weights(SO.unet_shapes(UCFG), 4)
    with pytest.raises(FileNotFoundError, match="tokenizer"):                    # real text-encoder weights need the real tokenizer files
        SDUtils(weights={"vae": vsd, "unet": usd}, arch=arch, verbose=False)
    monkeypatch.setenv("SVG_ALLOW_SYNTHETIC_WEIGHTS", "1")                        # (opt-in: stand-in tokenizer)
    s2 = SDUtils(weights={"vae": vsd, "unet": usd}, arch=arch, verbose=False)
    assert s2.clip_source.startswith("local:") and s2.clip_arch["layers"] == 2
    e = s2.encode_text(["Archery"])
    ids = CO.stand_in_ids(["Archery"], 77, 1000)
    margin("encode_text from a local text_encoder/ directory", rel_l2(e[1:].cpu(), CO.forward(tsd, ids, tcfg)), 1.7e-6)
