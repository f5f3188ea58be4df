"""ORACLE — TEST INFRASTRUCTURE ONLY.  Not shipped, not measured, never imported by
the product path (only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may import this).

CPU fp32 restatement of the reference's latent-sequence Transformer, written with
explicit tensor ops (no nn.Transformer) so every step the HIP path must reproduce
is visible.  Follows:

  * models/transformer.py:47-68   forward  (embedding*sqrt(d), PE, permute, nn.Transformer, out)
  * models/transformer.py:70-89   get_tgt_mask (lower-triangular 0 / -inf float mask)
  * models/positional_encoding.py:7-35   sinusoid table, indexed by dim 0 of a
    BATCH-FIRST tensor (quirk: row b of the batch gets PE(b) on every position)
  * torch.nn.Transformer defaults the reference relies on (transformer.py:38-44):
    post-norm, ReLU, dim_feedforward=2048, layer_norm_eps=1e-5, final encoder and
    decoder LayerNorm, sequence-first layout.
  * prediction/predict.py:16-42   predict(): eval mode, src == tgt, causal mask,
    returns the last sequence position of batch row 0.

Parity pin: tests/golden/transformer_*.pt hold outputs of the LIVE reference module
(imported from /root/reference by oracle/gen_golden.py); tests/test_oracle_transformer.py
checks this restatement against them.
"""
import math

import torch
import torch.nn.functional as F

DIM_FEEDFORWARD = 2048   # nn.Transformer default (transformer.py:38-44 passes none)
LN_EPS = 1e-5
PE_MAX_LEN = 64          # transformer.py:33-35


def positional_table(dim_model, max_len=PE_MAX_LEN):
    """positional_encoding.py:16-30 -> (max_len, 1, dim_model)."""
    pe = torch.zeros(max_len, dim_model)
    pos = torch.arange(0, max_len, dtype=torch.float).view(-1, 1)
    div = torch.exp(torch.arange(0, dim_model, 2).float() * (-math.log(10000.0)) / dim_model)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe.unsqueeze(0).transpose(0, 1)


def get_tgt_mask(size):
    """transformer.py:70-89."""
    mask = torch.tril(torch.ones(size, size) == 1).float()
    mask = mask.masked_fill(mask == 0, float("-inf"))
    mask = mask.masked_fill(mask == 1, float(0.0))
    return mask


def _pad_bias(pad):
    """key_padding_mask (B,Tk) as nn.MultiheadAttention canonicalises it: bool True -> -inf, float -> added as is"""
    if pad is None:
        return None
    if pad.dtype == torch.bool:
        return torch.zeros(pad.shape, dtype=torch.float32).masked_fill(pad, float("-inf"))
    return pad.float()


def _mha(sd, prefix, q_in, kv_in, num_heads, mask=None, drop=None, key_pad=None):
    """nn.MultiheadAttention (seq-first): q_in (Tq,B,d), kv_in (Tk,B,d).  drop: train-mode dropout on the probabilities.
    key_pad (B,Tk): key-padding mask, added to the scores of every head and query of batch row b (transformer.py:64)."""
    Tq, B, d = q_in.shape
    Tk = kv_in.shape[0]
    hd = d // num_heads
    w = sd[prefix + "in_proj_weight"]
    b = sd[prefix + "in_proj_bias"]
    q = F.linear(q_in, w[:d], b[:d])
    k = F.linear(kv_in, w[d:2 * d], b[d:2 * d])
    v = F.linear(kv_in, w[2 * d:], b[2 * d:])
    q = q.reshape(Tq, B * num_heads, hd).transpose(0, 1)
    k = k.reshape(Tk, B * num_heads, hd).transpose(0, 1)
    v = v.reshape(Tk, B * num_heads, hd).transpose(0, 1)
    s = torch.bmm(q, k.transpose(1, 2)) / math.sqrt(hd)
    if mask is not None:
        s = s + mask
    if key_pad is not None:
        s = s + _pad_bias(key_pad)[:, None, None, :].expand(B, num_heads, 1, Tk).reshape(B * num_heads, 1, Tk)
    p = torch.softmax(s, dim=-1)
    if drop is not None:
        p = drop(p)
    o = torch.bmm(p, v).transpose(0, 1).reshape(Tq, B, d)
    return F.linear(o, sd[prefix + "out_proj.weight"], sd[prefix + "out_proj.bias"])


def _ln(sd, prefix, x):
    return F.layer_norm(x, (x.shape[-1],), sd[prefix + "weight"], sd[prefix + "bias"], LN_EPS)


def _ffn(sd, prefix, x, drop=None):
    h = F.relu(F.linear(x, sd[prefix + "linear1.weight"], sd[prefix + "linear1.bias"]))
    if drop is not None:
        h = drop(h)
    return F.linear(h, sd[prefix + "linear2.weight"], sd[prefix + "linear2.bias"])


def count_layers(sd, stem):
    n = 0
    while (stem + "%d.norm1.weight" % n) in sd:
        n += 1
    return n


def forward(sd, src, tgt, num_heads, tgt_mask=None, txt=None, drop=None, src_pad_mask=None, tgt_pad_mask=None):
    """transformer.py:47-68.  src/tgt (B,T,D_lat) -> (T_tgt,B,D_lat).  eval mode (no dropout) unless `drop` is given:
    drop(x) is then applied at every dropout site of the train-mode module, in execution order — after the positional
    encoding of src and of tgt (positional_encoding.py:35), and per nn.Transformer layer on the attention probabilities,
    after each attention / feed-forward sublayer (dropout1/2/3) and inside the feed-forward (activation -> dropout -> linear2).
    With `txt` (B,384): the text-conditioned variant, models/transformer_text.py:71-111 — the embedding layer is
    `project_image_embedding` and every token is cat(proj(x), txt[b]) * sqrt(d), d = DIM_MODEL + 384 (:33-35,:82-92).
    src_pad_mask (B,Ts) / tgt_pad_mask (B,Tt): nn.Transformer's src_key_padding_mask (encoder self-attention keys) and
    tgt_key_padding_mask (decoder self-attention keys); the cross-attention gets none (memory_key_padding_mask is not passed).
    (That file cannot be imported here — sentence_transformers is missing — so this branch is pinned through its exact
    equivalence with the pinned base path: tests/test_oracle_transformer.py::test_text_variant_equivalence.)"""
    if txt is None:
        d = sd["embedding.weight"].shape[0]
        emb = lambda x: F.linear(x, sd["embedding.weight"], sd["embedding.bias"]) * math.sqrt(d)
    else:
        d = sd["project_image_embedding.weight"].shape[0] + txt.shape[-1]

        def emb(x):
            e = F.linear(x, sd["project_image_embedding.weight"], sd["project_image_embedding.bias"])
            return torch.cat((e, txt.unsqueeze(1).repeat(1, x.shape[1], 1)), dim=-1) * math.sqrt(d)
    pe = sd.get("positional_encoder.pos_encoding", positional_table(d))
    s = emb(src)
    t = emb(tgt)
    # positional_encoding.py:33-35 — slices by dim 0 of the batch-first tensor (quirk §9.1)
    s = s + pe[: s.size(0)]
    t = t + pe[: t.size(0)]
    s = s.permute(1, 0, 2)
    t = t.permute(1, 0, 2)
    dr = drop if drop is not None else (lambda x: x)
    s = dr(s)      # the sites see sequence-first tensors (the mask of an element does not depend on the layout)
    t = dr(t)
    # encoder
    for i in range(count_layers(sd, "transformer.encoder.layers.")):
        p = "transformer.encoder.layers.%d." % i
        s = _ln(sd, p + "norm1.", s + dr(_mha(sd, p + "self_attn.", s, s, num_heads, drop=drop, key_pad=src_pad_mask)))
        s = _ln(sd, p + "norm2.", s + dr(_ffn(sd, p, s, drop)))
    mem = _ln(sd, "transformer.encoder.norm.", s)
    # decoder
    for i in range(count_layers(sd, "transformer.decoder.layers.")):
        p = "transformer.decoder.layers.%d." % i
        t = _ln(sd, p + "norm1.", t + dr(_mha(sd, p + "self_attn.", t, t, num_heads, tgt_mask, drop=drop, key_pad=tgt_pad_mask)))
        t = _ln(sd, p + "norm2.", t + dr(_mha(sd, p + "multihead_attn.", t, mem, num_heads, drop=drop)))
        t = _ln(sd, p + "norm3.", t + dr(_ffn(sd, p, t, drop)))
    t = _ln(sd, "transformer.decoder.norm.", t)
    return F.linear(t, sd["out.weight"], sd["out.bias"])


def predict(sd, input_sequence, num_heads, txt=None):
    """prediction/predict.py:16-42 (and predict_text.py:48-74 with `txt`) -> (D_lat,)."""
    mask = get_tgt_mask(input_sequence.size(1))
    pred = forward(sd, input_sequence, input_sequence, num_heads, mask, txt)
    return pred.permute(1, 0, 2)[0, -1]


def rollout(sd, new_batch, num_heads, pred_frames, post=None):
    """prediction/predict.py:124-197 with the VAE taken out: `new_batch` (1,6,D) is the encoded
    clip with SOS in front.  `post(pred)` stands for the optional denoise round trip (:145-185).
    Returns (all_latents (1,4+N,D), trace of (X.shape, all_latents.shape) per iteration)."""
    X = new_batch
    inputs = new_batch[:, 1:]                      # :136-141 (SOS skipped)
    preds = torch.zeros(1, 0, new_batch.shape[-1])
    trace = []
    all_latents = None
    for _ in range(pred_frames):
        shape_in = tuple(X.shape)
        pred = predict(sd, X, num_heads)
        if post is not None:
            pred = post(pred)
        preds = torch.cat((preds, pred.reshape(1, 1, -1)), dim=1)           # :187-188
        all_latents = torch.cat([inputs[:, :-1], preds], dim=1)           # :193 (drops last cond frame)
        X = all_latents[:, -5:]                                           # :196 (window 5)
        trace.append((shape_in, tuple(all_latents.shape)))
    return all_latents, trace
