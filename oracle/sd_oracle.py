"""ORACLE — TEST INFRASTRUCTURE ONLY.  Not shipped, not measured as the product, never imported by the
product path (only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this).

CPU fp32 restatement, in plain torch functional ops, of the third-party numerics the reference's
SD wrapper calls into (utils/sd_utils.py:52-66,128-169,222-267):

  * AutoencoderKL        diffusers==0.2.3 (environment.yml:126), SD v1.4 vae/config.json
  * UNet2DConditionModel diffusers==0.2.3, SD v1.4 unet/config.json
  * DDIMScheduler        diffusers==0.2.3 scheduling_ddim.py (beta 0.00085..0.012 scaled_linear, 1000 steps,
                         clip_sample=True, set_alpha_to_one=True, eta=0)

The diffusers sources are NOT under /root/reference and the package is not installed here, so this
restates its published algorithm and reads weights by the diffusers state_dict
key names.  PARITY UNPINNED for the SD side: the reference holds no tests, fixtures or golden vectors
for it and no weights can be fetched offline.  What pins it instead (tests/test_oracle_sd.py): exact
parameter counts of the SD v1.4 architecture (UNet 859 520 964, VAE 83 653 863 = 34 163 664 encoder
+ quant/post-quant + 49 490 199 decoder), the DDIM closed-form checks, and structural identities.

Call-site restatements of the reference wrapper itself (these ARE pinned by reading the source):
  encode_img            utils/sd_utils.py:128-145
  decode_img_latents    utils/sd_utils.py:156-169
  gen_i2i_latents       utils/sd_utils.py:222-267
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

SD_UNET = dict(block_out=(320, 640, 1280, 1280), layers=2, heads=8, ctx_dim=768, groups=32, in_ch=4, out_ch=4,
               attn=(1, 1, 1, 0))
SD_VAE = dict(block_out=(128, 256, 512, 512), layers=2, groups=32, latent=4)
SCALE = 0.18215


# =================================================================================================
# precision: fp32 (default: what a CPU host executes) or the CUDA-autocast policy
# =================================================================================================
# utils/sd_utils.py:246 wraps the UNet loop in torch.autocast('cuda'): on the reference's GPU the UNet does NOT run in fp32.
# `with autocast_fp16():` makes this restatement apply torch 1.11's CUDA autocast policy op by op (the VAE call sites at
# sd_utils.py:140,162 sit outside the autocast block and stay fp32):
#   * conv2d / linear / matmul (bmm, einsum): every floating input is cast to fp16, the result IS an fp16 tensor (the products are
#     accumulated in fp32 by the tensor cores and rounded once: computed here in fp32 from the fp16-rounded operands, then rounded);
#   * group_norm / layer_norm / softmax: inputs cast to fp32, result fp32 (autocast's fp32 list);
#   * everything else (silu, gelu, +, *, cat, chunk, nearest interpolate) runs in the dtype of its inputs with the usual type
#     promotion: fp16 in -> fp16 out (one rounding per op), fp16 (+) fp32 -> fp32.
# torch.autocast('cpu') is NOT that policy (its fp32 list lacks the norms and softmax), hence the explicit form.
_AC = [None]


class autocast_fp16:
    def __init__(self, dtype=torch.float16):
        self.dtype = dtype

    def __enter__(self):
        self.prev = _AC[0]
        _AC[0] = self.dtype
        return self

    def __exit__(self, *a):
        _AC[0] = self.prev
        return False


def _lo(fn, *ts, **kw):
    """an op of autocast's lower-precision list"""
    if _AC[0] is None:
        return fn(*ts, **kw)
    lo = [t.to(_AC[0]).float() if isinstance(t, torch.Tensor) and t.is_floating_point() else t for t in ts]
    return fn(*lo, **kw).to(_AC[0])


def _f32(t):
    """input of an op of autocast's fp32 list"""
    return t.float() if _AC[0] is not None else t


def _linear(x, w, b=None):
    return _lo(F.linear, x, w, b)


def _matmul(a, b):
    return _lo(torch.matmul, a, b)


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


# =================================================================================================
# blocks
# =================================================================================================
def _gn(sd, p, x, groups, eps):
    return F.group_norm(_f32(x), groups, sd[p + ".weight"], sd[p + ".bias"], eps)


def _conv(sd, p, x, stride=1, padding=1):
    return _lo(F.conv2d, x, sd[p + ".weight"], sd[p + ".bias"], stride=stride, padding=padding)


def resnet(sd, p, x, groups, eps, temb=None):
    """ResnetBlock2D (output_scale_factor 1)."""
    h = _conv(sd, p + ".conv1", F.silu(_gn(sd, p + ".norm1", x, groups, eps)))
    if temb is not None:
        h = h + _linear(F.silu(temb), sd[p + ".time_emb_proj.weight"], sd[p + ".time_emb_proj.bias"])[:, :, None, None]
    h = _conv(sd, p + ".conv2", F.silu(_gn(sd, p + ".norm2", h, groups, eps)))
    if (p + ".conv_shortcut.weight") in sd:
        x = _conv(sd, p + ".conv_shortcut", x, padding=0)
    return x + h


def _cross_attention(sd, p, x, context, heads):
    """CrossAttention: to_q/k/v without bias, softmax(q k^T * d^-1/2) v, to_out.0 with bias."""
    B, S, C = x.shape
    d = C // heads
    q = _linear(x, sd[p + ".to_q.weight"])
    k = _linear(context, sd[p + ".to_k.weight"])
    v = _linear(context, sd[p + ".to_v.weight"])
    sp = lambda t: t.reshape(B, -1, heads, d).permute(0, 2, 1, 3)
    q, k, v = sp(q), sp(k), sp(v)
    a = torch.softmax(_f32(_matmul(q, k.transpose(-1, -2)) * (d ** -0.5)), dim=-1)
    o = _matmul(a, v).permute(0, 2, 1, 3).reshape(B, S, C)
    return _linear(o, sd[p + ".to_out.0.weight"], sd[p + ".to_out.0.bias"])


def spatial_transformer(sd, p, x, context, heads, groups):
    B, C, H, W = x.shape
    res = x
    h = _gn(sd, p + ".norm", x, groups, 1e-6)
    h = _conv(sd, p + ".proj_in", h, padding=0)
    h = h.permute(0, 2, 3, 1).reshape(B, H * W, C)
    t = p + ".transformer_blocks.0"
    ln = lambda n, y: F.layer_norm(_f32(y), (C,), sd[t + "." + n + ".weight"], sd[t + "." + n + ".bias"], 1e-5)
    n1 = ln("norm1", h)
    h = h + _cross_attention(sd, t + ".attn1", n1, n1, heads)
    h = h + _cross_attention(sd, t + ".attn2", ln("norm2", h), context, heads)
    ff = _linear(ln("norm3", h), sd[t + ".ff.net.0.proj.weight"], sd[t + ".ff.net.0.proj.bias"])
    hs, gate = ff.chunk(2, dim=-1)                                   # GEGLU
    h = h + _linear(hs * F.gelu(gate), sd[t + ".ff.net.2.weight"], sd[t + ".ff.net.2.bias"])
    h = h.reshape(B, H, W, C).permute(0, 3, 1, 2)
    return _conv(sd, p + ".proj_out", h, padding=0) + res


def timestep_embedding(t, dim):
    """get_timestep_embedding(flip_sin_to_cos=True, downscale_freq_shift=0)."""
    half = dim // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    a = t.float()[:, None] * freqs[None, :]
    return torch.cat([torch.cos(a), torch.sin(a)], dim=-1)


# =================================================================================================
# UNet2DConditionModel
# =================================================================================================
def unet_forward(sd, x, t, context, cfg=SD_UNET):
    bo, L, heads, groups = cfg["block_out"], cfg["layers"], cfg["heads"], cfg["groups"]
    attn = cfg.get("attn", (1,) * (len(bo) - 1) + (0,))
    t = torch.as_tensor(t, dtype=torch.float32).reshape(-1)
    if t.numel() == 1:
        t = t.repeat(x.shape[0])
    temb = timestep_embedding(t, bo[0])
    temb = _linear(temb, sd["time_embedding.linear_1.weight"], sd["time_embedding.linear_1.bias"])
    temb = _linear(F.silu(temb), sd["time_embedding.linear_2.weight"], sd["time_embedding.linear_2.bias"])
    h = _conv(sd, "conv_in", x)
    skips = [h]
    for i in range(len(bo)):
        for j in range(L):
            h = resnet(sd, "down_blocks.%d.resnets.%d" % (i, j), h, groups, 1e-5, temb)
            if attn[i]:
                h = spatial_transformer(sd, "down_blocks.%d.attentions.%d" % (i, j), h, context, heads, groups)
            skips.append(h)
        if i < len(bo) - 1:
            h = _conv(sd, "down_blocks.%d.downsamplers.0.conv" % i, h, stride=2, padding=1)
            skips.append(h)
    h = resnet(sd, "mid_block.resnets.0", h, groups, 1e-5, temb)
    h = spatial_transformer(sd, "mid_block.attentions.0", h, context, heads, groups)
    h = resnet(sd, "mid_block.resnets.1", h, groups, 1e-5, temb)
    for i in range(len(bo)):
        bi = len(bo) - 1 - i
        for j in range(L + 1):
            h = torch.cat([h, skips.pop()], dim=1)
            h = resnet(sd, "up_blocks.%d.resnets.%d" % (i, j), h, groups, 1e-5, temb)
            if attn[bi]:
                h = spatial_transformer(sd, "up_blocks.%d.attentions.%d" % (i, j), h, context, heads, groups)
        if i < len(bo) - 1:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = _conv(sd, "up_blocks.%d.upsamplers.0.conv" % i, h)
    h = F.silu(_gn(sd, "conv_norm_out", h, groups, 1e-5))
    return _conv(sd, "conv_out", h)            # fp16 under autocast_fp16 (the caller's scheduler arithmetic promotes it to fp32)


# =================================================================================================
# AutoencoderKL
# =================================================================================================
def _vae_attn(sd, p, x, groups):
    B, C, H, W = x.shape
    h = _gn(sd, p + ".group_norm", x, groups, 1e-6).reshape(B, C, H * W).transpose(1, 2)
    q = F.linear(h, sd[p + ".query.weight"], sd[p + ".query.bias"])
    k = F.linear(h, sd[p + ".key.weight"], sd[p + ".key.bias"])
    v = F.linear(h, sd[p + ".value.weight"], sd[p + ".value.bias"])
    scale = 1.0 / math.sqrt(math.sqrt(C))          # 1 head; applied to q and k each
    a = torch.softmax((q * scale) @ (k * scale).transpose(-1, -2), dim=-1)
    o = F.linear(a @ v, sd[p + ".proj_attn.weight"], sd[p + ".proj_attn.bias"])
    return o.transpose(1, 2).reshape(B, C, H, W) + x


def vae_encode_moments(sd, x, cfg=SD_VAE):
    """AutoencoderKL.encode up to the posterior parameters: (N,3,H,W) in [-1,1] -> (N,8,H/8,W/8)."""
    bo, L, groups = cfg["block_out"], cfg["layers"], cfg["groups"]
    h = _conv(sd, "encoder.conv_in", x)
    for i in range(len(bo)):
        for j in range(L):
            h = resnet(sd, "encoder.down_blocks.%d.resnets.%d" % (i, j), h, groups, 1e-6)
        if i < len(bo) - 1:
            h = F.pad(h, (0, 1, 0, 1))                      # Downsample2D with padding=0
            h = _conv(sd, "encoder.down_blocks.%d.downsamplers.0.conv" % i, h, stride=2, padding=0)
    h = resnet(sd, "encoder.mid_block.resnets.0", h, groups, 1e-6)
    h = _vae_attn(sd, "encoder.mid_block.attentions.0", h, groups)
    h = resnet(sd, "encoder.mid_block.resnets.1", h, groups, 1e-6)
    h = _conv(sd, "encoder.conv_out", F.silu(_gn(sd, "encoder.conv_norm_out", h, groups, 1e-6)))
    return _conv(sd, "quant_conv", h, padding=0)


def vae_sample(moments, eps=None):
    """DiagonalGaussianDistribution.sample(): mean + exp(0.5*clamp(logvar,-30,20)) * eps."""
    mean, logvar = moments.chunk(2, dim=1)
    if eps is None:
        return mean
    return mean + torch.exp(0.5 * logvar.clamp(-30.0, 20.0)) * eps


def vae_decode(sd, z, cfg=SD_VAE):
    bo, L, groups = cfg["block_out"], cfg["layers"], cfg["groups"]
    h = _conv(sd, "post_quant_conv", z, padding=0)
    h = _conv(sd, "decoder.conv_in", h)
    h = resnet(sd, "decoder.mid_block.resnets.0", h, groups, 1e-6)
    h = _vae_attn(sd, "decoder.mid_block.attentions.0", h, groups)
    h = resnet(sd, "decoder.mid_block.resnets.1", h, groups, 1e-6)
    for i in range(len(bo)):
        for j in range(L + 1):
            h = resnet(sd, "decoder.up_blocks.%d.resnets.%d" % (i, j), h, groups, 1e-6)
        if i < len(bo) - 1:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = _conv(sd, "decoder.up_blocks.%d.upsamplers.0.conv" % i, h)
    return _conv(sd, "decoder.conv_out", F.silu(_gn(sd, "decoder.conv_norm_out", h, groups, 1e-6)))


# =================================================================================================
# the reference wrapper's call sites
# =================================================================================================
def encode_img(sd, imgs_u8, eps=None, cfg=SD_VAE):
    """utils/sd_utils.py:128-145.  imgs (N,H,W,3) uint8 -> (N,4,H/8,W/8); eps = the .sample() draws."""
    x = imgs_u8 / 255.0
    x = x.float().permute(0, 3, 1, 2)
    x = 2 * (x - 0.5)
    return vae_sample(vae_encode_moments(sd, x, cfg), eps) * SCALE


def decode_img_latents(sd, latents, cfg=SD_VAE, return_float=False):
    """utils/sd_utils.py:156-169 -> numpy-style (N,8h,8w,3) uint8 (as a tensor)."""
    dec = vae_decode(sd, 1 / SCALE * latents, cfg)
    imgs = (dec / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 1).numpy()
    imgs = torch.from_numpy((imgs * 255).round().astype("uint8"))
    return (imgs, dec) if return_float else imgs


def resize_nearest_u8(img_nhwc, oh, ow):
    """prediction/predict.py:158,178: F.interpolate on the uint8 NCHW tensor, default mode nearest."""
    x = img_nhwc.permute(0, 3, 1, 2)
    return F.interpolate(x.float(), (oh, ow)).to(torch.uint8).permute(0, 2, 3, 1)


class DDIM:
    """diffusers 0.2.3 DDIMScheduler as constructed at utils/sd_utils.py:233-237."""

    def __init__(self, num_inference_steps=50, beta_start=0.00085, beta_end=0.012, num_train=1000):
        betas = np.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train, dtype=np.float32) ** 2
        self.alphas_cumprod = np.cumprod(1.0 - betas, axis=0)
        self.ratio = num_train // num_inference_steps
        self.timesteps = np.arange(0, num_train, self.ratio)[::-1].copy()
        self.final_alpha_cumprod = np.float32(1.0)          # set_alpha_to_one

    def add_noise(self, x0, noise, t):
        a = float(self.alphas_cumprod[t])
        return math.sqrt(a) * x0 + math.sqrt(1 - a) * noise

    def step(self, eps, t, x):
        prev = t - self.ratio
        a_t = float(self.alphas_cumprod[t])
        a_p = float(self.alphas_cumprod[prev]) if prev >= 0 else float(self.final_alpha_cumprod)
        x0 = (x - math.sqrt(1 - a_t) * eps) / math.sqrt(a_t)
        x0 = x0.clamp(-1, 1)                                   # clip_sample
        return math.sqrt(a_p) * x0 + math.sqrt(1 - a_p) * eps   # eta = 0


def gen_i2i_latents(sd, text_embeddings, latents, num_inference_steps=50, guidance_scale=7.5, start_step=10,
                    noise=None, cfg=SD_UNET, return_all_latents=False, unet=None):
    """utils/sd_utils.py:222-267 as executed on a CPU host (autocast('cuda') is a no-op there: fp32) — or, inside
    `with autocast_fp16():`, as executed on the reference's GPU: the UNet call under the CUDA autocast policy (its fp16 result and
    the fp16 guidance combine of :256-257 are promoted to fp32 by the scheduler's arithmetic on the fp32 latents).
    `noise` = the randn_like draw of :242.  `unet(x, t, ctx)` may replace the oracle UNet (tests)."""
    unet = unet or (lambda x, t, c: unet_forward(sd, x, t, c, cfg))
    sch = DDIM(num_inference_steps)
    if start_step > 0:
        latents = sch.add_noise(latents, noise, int(sch.timesteps[start_step]))
    hist = [latents]
    for t in sch.timesteps[start_step:]:
        inp = torch.cat([latents] * 2)
        eps = unet(inp, int(t), text_embeddings)
        e_u, e_t = eps.chunk(2)
        eps = e_u + guidance_scale * (e_t - e_u)
        latents = sch.step(eps, int(t), latents)
        hist.append(latents)
    return torch.cat(hist, dim=0) if return_all_latents else latents


class LMS:
    """diffusers 0.2.3 LMSDiscreteScheduler as constructed at utils/sd_utils.py:70-72 and driven by denoise_img_latents
    (utils/sd_utils.py:97-126): restated from the published algorithm (third-party source absent from /root/reference)."""

    def __init__(self, beta_start=0.00085, beta_end=0.012, num_train=1000):
        betas = np.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train, dtype=np.float32) ** 2
        ac = np.cumprod(1.0 - betas, axis=0)
        self.train_sigmas = ((1 - ac) / ac) ** 0.5
        self.num_train = num_train

    def set_timesteps(self, n):
        self.timesteps = np.linspace(self.num_train - 1, 0, n, dtype=float)
        lo, hi = np.floor(self.timesteps).astype(int), np.ceil(self.timesteps).astype(int)
        fr = np.mod(self.timesteps, 1.0)
        self.sigmas = np.concatenate([(1 - fr) * self.train_sigmas[lo] + fr * self.train_sigmas[hi], [0.0]])
        self.derivatives = []

    def coeff(self, order, t, cur):
        from scipy import integrate

        def f(tau):
            p = 1.0
            for k in range(order):
                if k != cur:
                    p *= (tau - self.sigmas[t - k]) / (self.sigmas[t - cur] - self.sigmas[t - k])
            return p
        return integrate.quad(f, self.sigmas[t], self.sigmas[t + 1], epsrel=1e-4)[0]

    def step(self, eps, i, x, order=4):
        self.derivatives.append(eps)                      # (x - (x - sigma eps)) / sigma
        if len(self.derivatives) > order:
            self.derivatives.pop(0)
        order = min(i + 1, order)
        return x + sum(self.coeff(order, i, o) * d for o, d in zip(range(order), reversed(self.derivatives)))


def denoise_img_latents(sd, text_embeddings, latents, num_inference_steps=50, guidance_scale=7.5, cfg=SD_UNET, unet=None):
    """utils/sd_utils.py:97-126 on a CPU host (fp32)."""
    unet = unet or (lambda x, t, c: unet_forward(sd, x, t, c, cfg))
    sch = LMS()
    sch.set_timesteps(num_inference_steps)
    latents = latents * float(sch.sigmas[0])
    for i, t in enumerate(sch.timesteps):
        sigma = float(sch.sigmas[i])
        inp = torch.cat([latents] * 2) / ((sigma ** 2 + 1) ** 0.5)
        eps = unet(inp, float(t), text_embeddings)
        e_u, e_t = eps.chunk(2)
        latents = sch.step(e_u + guidance_scale * (e_t - e_u), i, latents)
    return latents
