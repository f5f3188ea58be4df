"""GPU: the MiniLM sentence encoder in the library (svg_minilm_encode: BertModel + masked mean pooling + L2 normalisation, f32)
against the CPU oracle (oracle/minilm_oracle.py, pinned to transformers.BertModel) on seeded weights at the all-MiniLM-L6-v2
size, through the host module that stands where the reference's `sent_transformer` stands (models/transformer_text.py:12,82-83)."""
import os
import sys

import pytest
import torch

from conftest import margin, rel_l2

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import minilm_oracle as MO  # noqa: E402
from sd_video_gen_amd import _lib, minilm  # noqa: E402

pytestmark = pytest.mark.gpu
UCF = ["Apply Eye Makeup", "Baseball Pitch", "Wall Pushups", "Yo Yo", "Playing Guitar", "Blowing Candles", "x"]


def test_minilm_full_size_vs_oracle(ctx):
    enc = minilm.SentenceEncoder(weights="synthetic", seed=5, ctx=ctx)
    sd = {k[len("0.auto_model."):]: v for k, v in enc.state_dict().items() if "position_ids" not in k}
    assert sum(v.numel() for v in sd.values()) == 22_713_216
    out, hid = enc.encode(UCF, return_hidden=True)
    assert enc.n_params == 22_713_216 and ctx.model_dtype(_lib.SVG_MINILM) == "f32"
    ref = MO.encode(sd, UCF)
    assert out.shape == (len(UCF), 384) and torch.allclose(out.norm(dim=1).cpu(), torch.ones(len(UCF)), atol=1e-5)
    margin("MiniLM (all-MiniLM-L6-v2 size) sentence embeddings vs oracle", rel_l2(out.cpu(), ref), 2e-5)
    ids, lens = MO.stand_in_ids(UCF)
    mask = (torch.arange(ids.shape[1])[None, :] < lens[:, None]).long()
    h_ref = MO.bert_forward(sd, ids, mask)
    for b in range(len(UCF)):
        assert rel_l2(hid[b, : lens[b]].cpu(), h_ref[b, : lens[b]]) < 2e-5
    # a sentence alone == its row in the padded batch (padding mask), and long inputs are truncated at the sequence limit
    assert rel_l2(enc.encode(["Yo Yo"]).cpu(), ref[3:4]) < 2e-5
    long = enc.encode(["Playing Guitar " * 100, "x"])
    assert torch.isfinite(long).all() and rel_l2(long[1:].cpu(), ref[6:7]) < 2e-5
    # more sentences than one weight-stream pass serves (336 rows): 101 UCF-style names
    names = ["class number %d of the data set" % i for i in range(101)]
    big = enc.encode(names)
    assert rel_l2(big[[0, 50, 100]].cpu(), MO.encode(sd, [names[0], names[50], names[100]])) < 2e-5


def test_text_transformer_uses_the_library_encoder(ctx):
    from sd_video_gen_amd import config as svg_config
    from sd_video_gen_amd.transformer_text import Transformer as TextTransformer
    svg_config.set_args(["--dataset", "ball", "--config", "model_10_26"])
    torch.manual_seed(1)
    m = TextTransformer(dim_model=64, num_heads=8, num_encoder_layers=1, num_decoder_layers=1, st_weights="synthetic").eval().use_context(ctx)
    txt = m.encode_classes(["Archery", "Archery", "Bowling"])
    assert txt.shape == (3, 384) and torch.equal(txt[0], txt[1]) and not torch.equal(txt[0], txt[2])
    sd = {k[len("sent_transformer.0.auto_model."):]: v.cpu() for k, v in m.state_dict().items() if k.startswith("sent_transformer.") and "position_ids" not in k}
    assert rel_l2(txt.cpu(), MO.encode(sd, ["Archery", "Archery", "Bowling"])) < 2e-5
    X = torch.randn(3, 5, 256).cuda()
    out = m(X, ["Archery", "Archery", "Bowling"], X, m.get_tgt_mask(5).cuda())
    assert out.shape == (5, 3, 256) and torch.isfinite(out).all()
