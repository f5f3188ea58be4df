"""CPU: the CLIP text-tower oracle (oracle/clip_oracle.py) pinned against the transformers package installed in the
build container (CLIPTextModel with seeded weights — the hub checkpoint is unreachable offline), plus the host-side
tokenizer stand-in and the layout tables."""
import os
import sys

import pytest
import torch

from conftest import rel_l2

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import clip_oracle as CO  # noqa: E402

TINY = dict(vocab=1000, d_model=64, heads=4, layers=2, ffn=128, max_pos=77)


@pytest.mark.parametrize("cfg", [TINY, CO.SD_CLIP], ids=["tiny", "vit-l-14-text"])
def test_oracle_matches_transformers_clip_text_model(cfg):
    tr = pytest.importorskip("transformers")
    c = tr.CLIPTextConfig(vocab_size=cfg["vocab"], hidden_size=cfg["d_model"], intermediate_size=cfg["ffn"],
                          num_hidden_layers=cfg["layers"], num_attention_heads=cfg["heads"], max_position_embeddings=cfg["max_pos"],
                          hidden_act="quick_gelu", layer_norm_eps=1e-5)
    torch.manual_seed(0)
    m = tr.CLIPTextModel(c).eval()
    sd = {k: v.detach() for k, v in m.state_dict().items()}
    names = {(k[len("text_model."):] if k.startswith("text_model.") else k) for k in sd if "position_ids" not in k}
    assert names == set(CO.clip_text_shapes(cfg))
    assert sum(v.numel() for k, v in sd.items() if "position_ids" not in k) == sum(int(torch.tensor(s).prod()) for s in CO.clip_text_shapes(cfg).values())
    ids = CO.stand_in_ids(["", "a person doing WallPushups", "x " * 100], cfg["max_pos"], cfg["vocab"])
    ids = ids.clamp(max=cfg["vocab"] - 1)
    with torch.no_grad():
        want = m(input_ids=ids)[0]
        got = CO.forward(sd, ids, cfg)
    assert got.shape == (3, cfg["max_pos"], cfg["d_model"])
    assert rel_l2(got, want) < 2e-6


def test_param_count_of_the_sd_text_encoder():
    assert sum(int(torch.tensor(s).prod()) for s in CO.clip_text_shapes().values()) == 123_060_480


def test_stand_in_tokenizer():
    from sd_video_gen_amd.sd_utils import StandInTokenizer
    from sd_video_gen_amd import sd_layout
    assert sd_layout.clip_text_shapes() == CO.clip_text_shapes() and sd_layout.SD_CLIP == CO.SD_CLIP
    tok = StandInTokenizer()
    empty = tok([""], padding="max_length", max_length=tok.model_max_length, return_tensors="pt").input_ids
    assert empty.shape == (1, 77) and empty[0, 0] == 49406 and (empty[0, 1:] == 49407).all()      # what CLIPTokenizer gives ''
    prompts = ["", "a person doing WallPushups", "PlayingGuitar " * 90]
    ids = tok(prompts, padding="max_length", max_length=77, truncation=True, return_tensors="pt").input_ids
    assert torch.equal(ids, CO.stand_in_ids(prompts))
    assert ids.shape == (3, 77) and ids[2, 0] == 49406 and ids[2, -1] == 49407 and int(ids.max()) < 49408
    assert torch.equal(tok("abc").input_ids, tok(["abc"]).input_ids)
