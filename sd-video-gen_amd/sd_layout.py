"""Parameter layout of the SD v1.4 networks in diffusers' state_dict naming (what ``svg_load_weight`` expects
for SVG_VAE / SVG_UNET; reference call sites utils/sd_utils.py:52-53,65-66), plus a seeded synthetic-weight
generator: no checkpoint can be fetched offline, so benches and smoke tests run on random-init weights of the
exact architecture.  A user with a local diffusers-format state_dict hands that over instead.
"""
import math

import numpy as np
import torch

SD_UNET = dict(block_out=(320, 640, 1280, 1280), layers=2, heads=8, ctx_dim=768, groups=32, in_ch=4, out_ch=4,
               attn=(1, 1, 1, 0))
SD_VAE = dict(block_out=(128, 256, 512, 512), layers=2, groups=32, latent=4)
SD_CLIP = dict(vocab=49408, d_model=768, heads=12, layers=12, ffn=3072, max_pos=77)      # openai/clip-vit-large-patch14, text tower
SCALE = 0.18215


# =================================================================================================
# parameter tables (name -> shape) in diffusers' state_dict naming
# =================================================================================================
def _res_shapes(p, cin, cout, temb=None):
    s = {p + ".norm1.weight": (cin,), p + ".norm1.bias": (cin,),
         p + ".conv1.weight": (cout, cin, 3, 3), p + ".conv1.bias": (cout,),
         p + ".norm2.weight": (cout,), p + ".norm2.bias": (cout,),
         p + ".conv2.weight": (cout, cout, 3, 3), p + ".conv2.bias": (cout,)}
    if temb:
        s[p + ".time_emb_proj.weight"] = (cout, temb)
        s[p + ".time_emb_proj.bias"] = (cout,)
    if cin != cout:
        s[p + ".conv_shortcut.weight"] = (cout, cin, 1, 1)
        s[p + ".conv_shortcut.bias"] = (cout,)
    return s


def _st_shapes(p, C, ctx_dim):
    t = p + ".transformer_blocks.0"
    s = {p + ".norm.weight": (C,), p + ".norm.bias": (C,),
         p + ".proj_in.weight": (C, C, 1, 1), p + ".proj_in.bias": (C,),
         p + ".proj_out.weight": (C, C, 1, 1), p + ".proj_out.bias": (C,)}
    for n in ("norm1", "norm2", "norm3"):
        s[t + "." + n + ".weight"] = (C,)
        s[t + "." + n + ".bias"] = (C,)
    for a, kd in (("attn1", C), ("attn2", ctx_dim)):
        s[t + "." + a + ".to_q.weight"] = (C, C)
        s[t + "." + a + ".to_k.weight"] = (C, kd)
        s[t + "." + a + ".to_v.weight"] = (C, kd)
        s[t + "." + a + ".to_out.0.weight"] = (C, C)
        s[t + "." + a + ".to_out.0.bias"] = (C,)
    s[t + ".ff.net.0.proj.weight"] = (8 * C, C)
    s[t + ".ff.net.0.proj.bias"] = (8 * C,)
    s[t + ".ff.net.2.weight"] = (C, 4 * C)
    s[t + ".ff.net.2.bias"] = (C,)
    return s


def unet_shapes(cfg=SD_UNET):
    bo, L, ctx_dim = cfg["block_out"], cfg["layers"], cfg["ctx_dim"]
    attn = cfg.get("attn", (1,) * (len(bo) - 1) + (0,))
    c0, temb = bo[0], 4 * bo[0]
    s = {"time_embedding.linear_1.weight": (temb, c0), "time_embedding.linear_1.bias": (temb,),
         "time_embedding.linear_2.weight": (temb, temb), "time_embedding.linear_2.bias": (temb,),
         "conv_in.weight": (c0, cfg["in_ch"], 3, 3), "conv_in.bias": (c0,)}
    skips, cin = [c0], c0
    for i, co in enumerate(bo):
        for j in range(L):
            s.update(_res_shapes("down_blocks.%d.resnets.%d" % (i, j), cin, co, temb))
            cin = co
            if attn[i]:
                s.update(_st_shapes("down_blocks.%d.attentions.%d" % (i, j), co, ctx_dim))
            skips.append(co)
        if i < len(bo) - 1:
            s["down_blocks.%d.downsamplers.0.conv.weight" % i] = (co, co, 3, 3)
            s["down_blocks.%d.downsamplers.0.conv.bias" % i] = (co,)
            skips.append(co)
    s.update(_res_shapes("mid_block.resnets.0", cin, cin, temb))
    s.update(_st_shapes("mid_block.attentions.0", cin, ctx_dim))
    s.update(_res_shapes("mid_block.resnets.1", cin, cin, temb))
    for i in range(len(bo)):
        bi = len(bo) - 1 - i
        co = bo[bi]
        for j in range(L + 1):
            s.update(_res_shapes("up_blocks.%d.resnets.%d" % (i, j), cin + skips.pop(), co, temb))
            cin = co
            if attn[bi]:
                s.update(_st_shapes("up_blocks.%d.attentions.%d" % (i, j), co, ctx_dim))
        if i < len(bo) - 1:
            s["up_blocks.%d.upsamplers.0.conv.weight" % i] = (co, co, 3, 3)
            s["up_blocks.%d.upsamplers.0.conv.bias" % i] = (co,)
    s["conv_norm_out.weight"] = (c0,)
    s["conv_norm_out.bias"] = (c0,)
    s["conv_out.weight"] = (cfg["out_ch"], c0, 3, 3)
    s["conv_out.bias"] = (cfg["out_ch"],)
    return s


def _vae_attn_shapes(p, C):
    s = {p + ".group_norm.weight": (C,), p + ".group_norm.bias": (C,)}
    for n in ("query", "key", "value", "proj_attn"):
        s[p + "." + n + ".weight"] = (C, C)
        s[p + "." + n + ".bias"] = (C,)
    return s


def vae_shapes(cfg=SD_VAE):
    bo, L, lat = cfg["block_out"], cfg["layers"], cfg["latent"]
    s = {"encoder.conv_in.weight": (bo[0], 3, 3, 3), "encoder.conv_in.bias": (bo[0],)}
    cin = bo[0]
    for i, co in enumerate(bo):
        for j in range(L):
            s.update(_res_shapes("encoder.down_blocks.%d.resnets.%d" % (i, j), cin, co))
            cin = co
        if i < len(bo) - 1:
            s["encoder.down_blocks.%d.downsamplers.0.conv.weight" % i] = (co, co, 3, 3)
            s["encoder.down_blocks.%d.downsamplers.0.conv.bias" % i] = (co,)
    cm = bo[-1]
    s.update(_res_shapes("encoder.mid_block.resnets.0", cm, cm))
    s.update(_vae_attn_shapes("encoder.mid_block.attentions.0", cm))
    s.update(_res_shapes("encoder.mid_block.resnets.1", cm, cm))
    s.update({"encoder.conv_norm_out.weight": (cm,), "encoder.conv_norm_out.bias": (cm,),
              "encoder.conv_out.weight": (2 * lat, cm, 3, 3), "encoder.conv_out.bias": (2 * lat,),
              "quant_conv.weight": (2 * lat, 2 * lat, 1, 1), "quant_conv.bias": (2 * lat,),
              "post_quant_conv.weight": (lat, lat, 1, 1), "post_quant_conv.bias": (lat,),
              "decoder.conv_in.weight": (cm, lat, 3, 3), "decoder.conv_in.bias": (cm,)})
    s.update(_res_shapes("decoder.mid_block.resnets.0", cm, cm))
    s.update(_vae_attn_shapes("decoder.mid_block.attentions.0", cm))
    s.update(_res_shapes("decoder.mid_block.resnets.1", cm, cm))
    cin = cm
    for i in range(len(bo)):
        co = bo[len(bo) - 1 - i]
        for j in range(L + 1):
            s.update(_res_shapes("decoder.up_blocks.%d.resnets.%d" % (i, j), cin, co))
            cin = co
        if i < len(bo) - 1:
            s["decoder.up_blocks.%d.upsamplers.0.conv.weight" % i] = (co, co, 3, 3)
            s["decoder.up_blocks.%d.upsamplers.0.conv.bias" % i] = (co,)
    s.update({"decoder.conv_norm_out.weight": (bo[0],), "decoder.conv_norm_out.bias": (bo[0],),
              "decoder.conv_out.weight": (3, bo[0], 3, 3), "decoder.conv_out.bias": (3,)})
    return s


def count(shapes):
    return sum(int(np.prod(v)) for v in shapes.values())


def seeded_weights(shapes, seed, gain=0.6, device="cpu"):
    """Synthetic weights (no checkpoints offline): matrices/convs N(0, gain/sqrt(fan_in)), norm scale
    1 + N(0,.1), norm shift / biases N(0,.05).  Generated per tensor from (seed, name) so any subset
    reproduces, on any device, the same values."""
    import zlib
    sd = {}
    for name, shape in shapes.items():
        g = torch.Generator().manual_seed((seed * 1000003 + zlib.crc32(name.encode())) % (2 ** 31))
        if len(shape) >= 2:
            fan_in = int(np.prod(shape[1:]))
            t = torch.randn(shape, generator=g) * (gain / math.sqrt(fan_in))
        elif "norm" in name and name.endswith(".weight"):
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        else:
            t = 0.05 * torch.randn(shape, generator=g)
        sd[name] = t.to(device)
    return sd


def clip_text_shapes(cfg=SD_CLIP):
    """transformers CLIPTextModel state_dict names (without the 'text_model.' prefix) -> shapes"""
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
