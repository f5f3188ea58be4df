"""ORACLE — TEST INFRASTRUCTURE ONLY (imported by tests/, smoke and bench's cpu_baseline leg; never by the product).

CPU restatement of the CLIP text tower the reference calls at utils/sd_utils.py:59-60,78-95:
`CLIPTokenizer` / `CLIPTextModel.from_pretrained('openai/clip-vit-large-patch14')`, `text_encoder(input_ids)[0]`.
The arithmetic lives in a third-party dependency whose source is NOT under /root/reference: transformers==4.21.0
(environment.yml:157), modeling_clip.py (CLIPTextEmbeddings, CLIPAttention, CLIPMLP, CLIPEncoderLayer,
CLIPTextTransformer).  Restated from that published algorithm:

    x = token_embedding[ids] + position_embedding[arange(T)]
    for each layer:  r = x; h = layer_norm1(x)
                     q = q_proj(h) * hd**-0.5; k = k_proj(h); v = v_proj(h)          (per head)
                     a = softmax(q k^T + causal_mask) v;  x = r + out_proj(a)
                     r = x; h = layer_norm2(x); x = r + fc2(quick_gelu(fc1(h)))      quick_gelu(u) = u * sigmoid(1.702 u)
    last_hidden_state = final_layer_norm(x)

PINNED: tests/test_oracle_clip.py checks this restatement against the transformers package installed in the build
container (5.15.0: same text-tower math as 4.21.0) on seeded weights, at the tiny and at the ViT-L/14 text size.  The
hub weights and the tokenizer's vocab/merges files are unavailable offline; `stand_in_ids` reproduces the ids the real
tokenizer gives the empty prompt (BOS, EOS, then EOS padding) and hashes the words of other prompts (synthetic weights only).
"""
import math
import zlib

import torch
import torch.nn.functional as F

SD_CLIP = dict(vocab=49408, d_model=768, heads=12, layers=12, ffn=3072, max_pos=77)
BOS, EOS = 49406, 49407


def clip_text_shapes(cfg=SD_CLIP):
    d, f = cfg["d_model"], cfg["ffn"]
    s = {"embeddings.token_embedding.weight": (cfg["vocab"], d), "embeddings.position_embedding.weight": (cfg["max_pos"], d),
         "final_layer_norm.weight": (d,), "final_layer_norm.bias": (d,)}
    for i in range(cfg["layers"]):
        p = "encoder.layers.%d." % i
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            s[p + "self_attn." + n + ".weight"] = (d, d)
            s[p + "self_attn." + n + ".bias"] = (d,)
        s[p + "mlp.fc1.weight"] = (f, d)
        s[p + "mlp.fc1.bias"] = (f,)
        s[p + "mlp.fc2.weight"] = (d, f)
        s[p + "mlp.fc2.bias"] = (d,)
        for n in ("layer_norm1", "layer_norm2"):
            s[p + n + ".weight"] = (d,)
            s[p + n + ".bias"] = (d,)
    return s


def forward(sd, input_ids, cfg=SD_CLIP):
    """input_ids (B,T) long -> last_hidden_state (B,T,d) f32"""
    sd = {(k[len("text_model."):] if k.startswith("text_model.") else k): v for k, v in sd.items()}
    B, T = input_ids.shape
    d, H = cfg["d_model"], cfg["heads"]
    hd = d // H
    x = sd["embeddings.token_embedding.weight"][input_ids] + sd["embeddings.position_embedding.weight"][:T][None]
    mask = torch.full((T, T), float("-inf")).triu(1)
    for i in range(cfg["layers"]):
        p = "encoder.layers.%d." % i
        r = x
        h = F.layer_norm(x, (d,), sd[p + "layer_norm1.weight"], sd[p + "layer_norm1.bias"], 1e-5)
        q = F.linear(h, sd[p + "self_attn.q_proj.weight"], sd[p + "self_attn.q_proj.bias"]) * hd ** -0.5
        k = F.linear(h, sd[p + "self_attn.k_proj.weight"], sd[p + "self_attn.k_proj.bias"])
        v = F.linear(h, sd[p + "self_attn.v_proj.weight"], sd[p + "self_attn.v_proj.bias"])
        q, k, v = (t.reshape(B, T, H, hd).transpose(1, 2) for t in (q, k, v))
        a = torch.softmax(q @ k.transpose(-1, -2) + mask, dim=-1) @ v
        a = a.transpose(1, 2).reshape(B, T, d)
        x = r + F.linear(a, sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"])
        r = x
        h = F.layer_norm(x, (d,), sd[p + "layer_norm2.weight"], sd[p + "layer_norm2.bias"], 1e-5)
        h = F.linear(h, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])
        h = h * torch.sigmoid(1.702 * h)
        x = r + F.linear(h, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
    return F.layer_norm(x, (d,), sd["final_layer_norm.weight"], sd["final_layer_norm.bias"], 1e-5)


def stand_in_ids(prompts, max_length=77, vocab=49408):
    """(n, max_length) ids: BOS, one id per whitespace-separated word (crc32 hash into the vocabulary), EOS, EOS padding —
    for '' exactly what CLIPTokenizer(padding='max_length') returns."""
    out = []
    for p in prompts:
        bos, eos = vocab - 2, vocab - 1            # 49406 / 49407 at CLIP's vocabulary size
        ids = [bos] + [zlib.crc32(w.lower().encode()) % (vocab - 2) for w in p.split()][: max_length - 2] + [eos]
        out.append(ids + [eos] * (max_length - len(ids)))
    return torch.tensor(out, dtype=torch.long)


def encode_text(sd, prompts, cfg=SD_CLIP, tokenize=stand_in_ids):
    """utils/sd_utils.py:78-95 -> (2n, T, d) = [uncond(''); text]"""
    text = forward(sd, tokenize(list(prompts)), cfg)
    uncond = forward(sd, tokenize([""] * len(prompts)), cfg)
    return torch.cat([uncond, text])
