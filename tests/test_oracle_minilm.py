"""CPU: the MiniLM sentence-encoder oracle (oracle/minilm_oracle.py) pinned against the transformers package installed in the
build container (BertModel from a config with seeded weights — the hub checkpoint is unreachable offline), the pooling /
normalisation identities, and the host-side tokenizer stand-in + parameter tree of the product."""
import os
import sys

import pytest
import torch

from conftest import rel_l2

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import minilm_oracle as MO  # noqa: E402

TINY = dict(vocab=2000, d_model=64, heads=4, layers=2, ffn=128, max_pos=64)


@pytest.mark.parametrize("cfg", [TINY, MO.MINILM], ids=["tiny", "all-MiniLM-L6-v2"])
def test_oracle_matches_transformers_bert_model(cfg):
    tr = pytest.importorskip("transformers")
    c = tr.BertConfig(vocab_size=cfg["vocab"], hidden_size=cfg["d_model"], intermediate_size=cfg["ffn"], num_hidden_layers=cfg["layers"],
                      num_attention_heads=cfg["heads"], max_position_embeddings=cfg["max_pos"], hidden_act="gelu", layer_norm_eps=1e-12,
                      type_vocab_size=2, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    torch.manual_seed(0)
    m = tr.BertModel(c).eval()
    sd = {k: v.detach() for k, v in m.state_dict().items() if "position_ids" not in k and "token_type_ids" not in k}
    assert set(sd) == set(MO.bert_shapes(cfg)) and all(tuple(sd[k].shape) == tuple(v) for k, v in MO.bert_shapes(cfg).items())
    ids, lens = MO.stand_in_ids(["Apply Eye Makeup", "x", "a person doing WallPushups, twice", "Playing Guitar " * 30], 48, cfg["vocab"])
    mask = (torch.arange(ids.shape[1])[None, :] < lens[:, None]).long()
    with torch.no_grad():
        want = m(input_ids=ids, attention_mask=mask)[0]
        got = MO.bert_forward(sd, ids, mask, cfg)
    for b in range(ids.shape[0]):          # positions behind the padding are arbitrary in both (never pooled)
        assert rel_l2(got[b, : lens[b]], want[b, : lens[b]]) < 3e-6
    emb = MO.pool_normalize(got, mask)
    want_emb = torch.nn.functional.normalize((want * mask[:, :, None]).sum(1) / mask.sum(1, keepdim=True).clamp(min=1e-9), dim=1)
    assert rel_l2(emb, want_emb) < 3e-6
    assert torch.allclose(emb.norm(dim=1), torch.ones(4), atol=1e-6)
    # padding invariance: a sentence encoded alone equals its row in the padded batch
    alone = MO.encode(sd, ["x"], cfg, lambda s: MO.stand_in_ids(s, 48, cfg["vocab"]))
    assert rel_l2(alone, emb[1:2]) < 3e-6


def test_param_count_of_all_minilm_l6_v2():
    assert sum(int(torch.tensor(s).prod()) for s in MO.bert_shapes().values()) == 22_713_216


def test_product_tokenizer_and_parameter_tree():
    from sd_video_gen_amd import minilm
    tok = minilm.StandInWordPiece()
    ids, lens = tok(["Apply Eye Makeup", "", "Playing Guitar " * 100])
    ref_ids, ref_lens = MO.stand_in_ids(["Apply Eye Makeup", "", "Playing Guitar " * 100], 128)
    assert torch.equal(ids, ref_ids) and torch.equal(lens, ref_lens)
    assert ids[1, 0] == 101 and ids[1, 1] == 102 and lens[1] == 2 and ids.shape[1] == 128
    enc = minilm.SentenceEncoder(weights="synthetic", cfg=TINY, seed=3)
    keys = set(enc.state_dict().keys())
    assert "0.auto_model.embeddings.word_embeddings.weight" in keys and "0.auto_model.encoder.layer.1.output.LayerNorm.bias" in keys
    assert "0.auto_model.embeddings.position_ids" in keys            # the persistent buffer of transformers 4.21's BertEmbeddings
    assert {k[len("0.auto_model."):] for k in keys if "position_ids" not in k} == set(MO.bert_shapes(TINY))
    enc2 = minilm.SentenceEncoder(weights=None, cfg=TINY)
    assert len(enc2.state_dict()) == 0 and not enc2.loaded
    enc2.load_state_dict(enc.state_dict())
    assert enc2.loaded and all(torch.equal(enc2.state_dict()[k], v) for k, v in enc.state_dict().items())
