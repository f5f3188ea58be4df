"""ORACLE — TEST INFRASTRUCTURE ONLY (imported by tests/ and smoke; never by the product).

CPU restatement of the class-name encoder of the text-conditioned latent Transformer (reference:
models/transformer_text.py:12 `SentenceTransformer('sentence-transformers/all-MiniLM-L6-v2')`, :82-83 `.encode(cls_list)`).
The arithmetic lives in third-party dependencies whose sources are NOT under /root/reference: sentence-transformers
(unpinned in environment.yml) on top of transformers==4.21.0 (environment.yml:157).  Restated from the published algorithms:

  transformers modeling_bert.BertModel (all-MiniLM-L6-v2: 6 layers, hidden 384, 12 heads, intermediate 1536, vocab 30522,
  512 positions, 2 token types, hidden_act gelu, layer_norm_eps 1e-12):
      x = LayerNorm(word_embeddings[ids] + position_embeddings[arange(T)] + token_type_embeddings[0])
      per layer:  a = softmax(q k^T / sqrt(hd) + (1 - attention_mask) * -inf) v
                  x = LayerNorm(x + attention.output.dense(a))
                  x = LayerNorm(x + output.dense(gelu(intermediate.dense(x))))             gelu = x * Phi(x) (erf form)
  sentence-transformers models.Pooling(pooling_mode_mean_tokens) and models.Normalize:
      s = sum_t x[t] * mask[t] / clamp(sum_t mask[t], min=1e-9);   out = s / max(||s||_2, 1e-12)

PINNED: tests/test_oracle_minilm.py checks the BertModel part against the transformers package installed in the build container
(BertModel built from a BertConfig with seeded weights, tiny and at the all-MiniLM-L6-v2 size, with padded batches).
sentence-transformers itself is not installed: pooling and normalisation are restated from its documentation (two lines).
The hub weights and vocab.txt are unavailable offline; `stand_in_ids` is the host tokenizer stand-in for synthetic weights.
"""
import math
import re
import zlib

import torch
import torch.nn.functional as F

MINILM = dict(vocab=30522, d_model=384, heads=12, layers=6, ffn=1536, max_pos=512)
CLS, SEP, PAD = 101, 102, 0


def bert_shapes(cfg=MINILM, pooler=True):
    d, f = cfg["d_model"], cfg["ffn"]
    s = {"embeddings.word_embeddings.weight": (cfg["vocab"], d), "embeddings.position_embeddings.weight": (cfg["max_pos"], d),
         "embeddings.token_type_embeddings.weight": (2, d), "embeddings.LayerNorm.weight": (d,), "embeddings.LayerNorm.bias": (d,)}
    for i in range(cfg["layers"]):
        p = "encoder.layer.%d." % i
        for n in ("attention.self.query", "attention.self.key", "attention.self.value", "attention.output.dense"):
            s[p + n + ".weight"] = (d, d)
            s[p + n + ".bias"] = (d,)
        s[p + "intermediate.dense.weight"] = (f, d)
        s[p + "intermediate.dense.bias"] = (f,)
        s[p + "output.dense.weight"] = (d, f)
        s[p + "output.dense.bias"] = (d,)
        for n in ("attention.output.LayerNorm", "output.LayerNorm"):
            s[p + n + ".weight"] = (d,)
            s[p + n + ".bias"] = (d,)
    if pooler:
        s["pooler.dense.weight"] = (d, d)
        s["pooler.dense.bias"] = (d,)
    return s


def bert_forward(sd, input_ids, attention_mask, cfg=MINILM):
    """input_ids (B,T) long, attention_mask (B,T) 0/1 -> last_hidden_state (B,T,d)"""
    B, T = input_ids.shape
    d, H = cfg["d_model"], cfg["heads"]
    hd = d // H
    ln = lambda x, p: F.layer_norm(x, (d,), sd[p + ".weight"], sd[p + ".bias"], 1e-12)
    x = sd["embeddings.word_embeddings.weight"][input_ids] + sd["embeddings.position_embeddings.weight"][:T][None] \
        + sd["embeddings.token_type_embeddings.weight"][0][None, None]
    x = ln(x, "embeddings.LayerNorm")
    bias = torch.zeros(B, 1, 1, T).masked_fill(attention_mask[:, None, None, :] == 0, float("-inf"))
    for i in range(cfg["layers"]):
        p = "encoder.layer.%d." % i
        q = F.linear(x, sd[p + "attention.self.query.weight"], sd[p + "attention.self.query.bias"])
        k = F.linear(x, sd[p + "attention.self.key.weight"], sd[p + "attention.self.key.bias"])
        v = F.linear(x, sd[p + "attention.self.value.weight"], sd[p + "attention.self.value.bias"])
        q, k, v = (t.reshape(B, T, H, hd).transpose(1, 2) for t in (q, k, v))
        a = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(hd) + bias, dim=-1) @ v
        a = a.transpose(1, 2).reshape(B, T, d)
        x = ln(x + F.linear(a, sd[p + "attention.output.dense.weight"], sd[p + "attention.output.dense.bias"]), p + "attention.output.LayerNorm")
        h = F.gelu(F.linear(x, sd[p + "intermediate.dense.weight"], sd[p + "intermediate.dense.bias"]))
        x = ln(x + F.linear(h, sd[p + "output.dense.weight"], sd[p + "output.dense.bias"]), p + "output.LayerNorm")
    return x


def pool_normalize(hidden, attention_mask):
    """sentence-transformers Pooling(mean) + Normalize"""
    m = attention_mask[:, :, None].float()
    s = (hidden * m).sum(1) / m.sum(1).clamp(min=1e-9)
    return F.normalize(s, p=2, dim=1)


def stand_in_ids(sentences, max_length=128, vocab=30522):
    """host tokenizer stand-in (the real vocab.txt is hub-only): lower-case, split into words and punctuation, one crc32-hashed id
    per piece in [1000, vocab), [CLS] ... [SEP], zero padding to the longest row.  -> (ids (n,T) long, lengths (n,))"""
    rows = []
    for s in sentences:
        pieces = re.findall(r"[a-z0-9]+|[^\sa-z0-9]", str(s).lower())
        rows.append([CLS] + [1000 + zlib.crc32(w.encode()) % (vocab - 1000) for w in pieces][: max_length - 2] + [SEP])
    T = max(len(r) for r in rows)
    ids = torch.tensor([r + [PAD] * (T - len(r)) for r in rows], dtype=torch.long)
    return ids, torch.tensor([len(r) for r in rows], dtype=torch.long)


def encode(sd, sentences, cfg=MINILM, tokenize=stand_in_ids):
    """SentenceTransformer.encode(sentences) -> (n, d) unit-norm f32"""
    ids, lens = tokenize(list(sentences))
    mask = (torch.arange(ids.shape[1])[None, :] < lens[:, None]).long()
    return pool_normalize(bert_forward(sd, ids, mask, cfg), mask)
