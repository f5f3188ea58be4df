#!/usr/bin/env python3
"""CPU analysis (not a pytest module; run by hand): WHERE does 16-bit storage put the decode@512 error?

The reference decodes in fp32 (utils/sd_utils.py:162).  The HIP decoder accumulates in f32 but stores every tensor it writes to HBM in
fp16.  This script replays the oracle's decoder (oracle/sd_oracle.py: vae_decode) on the fixture's denoised latent with the library's
rounding points made explicit, and switches them off per site:
  conv / GN+SiLU outputs      every tensor a kernel writes is rounded to fp16 (`r`)
  residual stream             x + h of a resnet, the mid-block attention's x + o: the running sum a block hands to the next
  last GN+SiLU -> conv_out    the 16-bit A operand of the final conv
and reports, against the all-fp32 decode: rel-L2 of the float image, mean |error| in uint8 LSB, and the share of uint8 pixels that
change at 512 x 512 and after the nearest resize to F = 64 (prediction/predict.py:178).
usage: python tests/analysis_vae_decoder_storage.py [variant ...]   -> prints a table, writes gpurun_out/r05_vae_decoder_storage.json"""
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import gen_golden_sd as GG, sd_oracle as SO  # noqa: E402


def decode(sd, z, round_ops, stream_f32_levels, last_f32, dt=torch.float16, operands_only=False):
    """round_ops: round conv / norm outputs to `dt`; stream_f32_levels: set of levels ('mid', 0, 1, 2, 3) whose residual stream stays f32;
    last_f32: conv_norm_out's output (the A operand of conv_out) stays f32; operands_only: NOTHING is stored in 16 bits, only the two
    operands of every matrix product (weights, conv / linear inputs) are rounded — the floor of any implementation on 16-bit MFMAs"""
    cfg = SO.SD_VAE
    bo, L, groups = cfg["block_out"], cfg["layers"], cfg["groups"]
    rr = lambda t: t.to(dt).float()
    r = rr if (round_ops and not operands_only) else (lambda t: t)
    ro = rr if (round_ops or operands_only) else (lambda t: t)          # matrix operands
    if round_ops or operands_only:
        sd = {k: (rr(v) if v.dim() >= 2 else v) for k, v in sd.items()}     # the packed weights are 16-bit (biases / norm parameters stay f32)

    def conv(p, x, padding=1):
        return F.conv2d(ro(x), sd[p + ".weight"], sd[p + ".bias"], padding=padding)

    def gn_silu(p, x):
        return r(F.silu(F.group_norm(x, groups, sd[p + ".weight"], sd[p + ".bias"], 1e-6)))

    def resnet(p, x, lvl):
        keep = lvl in stream_f32_levels
        h = r(conv(p + ".conv1", gn_silu(p + ".norm1", x)))
        h = conv(p + ".conv2", gn_silu(p + ".norm2", h))                      # f32 accumulator of the conv2 kernel
        sc = r(conv(p + ".conv_shortcut", r(x), padding=0)) if (p + ".conv_shortcut.weight") in sd else x
        y = sc + h                                                            # added in the epilogue, in f32
        return y if keep else r(y)

    def attn(p, x, lvl):
        B, C, H, W = x.shape
        h = r(F.group_norm(x, groups, sd[p + ".group_norm.weight"], sd[p + ".group_norm.bias"], 1e-6)).reshape(B, C, H * W).transpose(1, 2)
        h = ro(h)
        q = ro(r(F.linear(h, sd[p + ".query.weight"], sd[p + ".query.bias"])))
        k = ro(r(F.linear(h, sd[p + ".key.weight"], sd[p + ".key.bias"])))
        v = ro(r(F.linear(h, sd[p + ".value.weight"], sd[p + ".value.bias"])))
        scale = 1.0 / (C ** 0.25)
        a = ro(r(torch.softmax((q * scale) @ (k * scale).transpose(-1, -2), dim=-1)))
        o = F.linear(ro(r(a @ v)), sd[p + ".proj_attn.weight"], sd[p + ".proj_attn.bias"]).transpose(1, 2).reshape(B, C, H, W)
        y = o + x
        return y if lvl in stream_f32_levels else r(y)

    h = r(conv("post_quant_conv", z, padding=0))
    h = r(conv("decoder.conv_in", h))
    h = resnet("decoder.mid_block.resnets.0", h, "mid")
    h = attn("decoder.mid_block.attentions.0", h, "mid")
    h = resnet("decoder.mid_block.resnets.1", h, "mid")
    for i in range(len(bo)):
        for j in range(L + 1):
            h = resnet("decoder.up_blocks.%d.resnets.%d" % (i, j), h, i)
        if i < len(bo) - 1:
            h = F.interpolate(r(h), scale_factor=2.0, mode="nearest")        # the upsampler conv's A operand is 16-bit
            y = conv("decoder.up_blocks.%d.upsamplers.0.conv" % i, h)
            h = y if (i + 1) in stream_f32_levels else r(y)
    g = F.silu(F.group_norm(h, groups, sd["decoder.conv_norm_out.weight"], sd["decoder.conv_norm_out.bias"], 1e-6))
    if last_f32:                                                              # (an f32 operand here = a conv outside the 16-bit MFMA path)
        return F.conv2d(g, sd["decoder.conv_out.weight"], sd["decoder.conv_out.bias"], padding=1)
    return conv("decoder.conv_out", r(g))                                     # the image leaves in f32 (clamp + uint8 in the epilogue)


VARIANTS = {
    "fp32 (the reference's decoder)": dict(round_ops=False, stream_f32_levels=set(), last_f32=True),
    "fp16 storage everywhere (round 4)": dict(round_ops=True, stream_f32_levels=set(), last_f32=False),
    "f32 residual stream at 512^2 (up_blocks.3)": dict(round_ops=True, stream_f32_levels={3}, last_f32=False),
    "f32 residual stream at 256^2 + 512^2": dict(round_ops=True, stream_f32_levels={2, 3}, last_f32=False),
    "f32 residual stream everywhere": dict(round_ops=True, stream_f32_levels={"mid", 0, 1, 2, 3}, last_f32=False),
    "f32 stream everywhere + f32 operand of conv_out": dict(round_ops=True, stream_f32_levels={"mid", 0, 1, 2, 3}, last_f32=True),
    "fp16 storage, only conv_out's operand f32": dict(round_ops=True, stream_f32_levels=set(), last_f32=True),
    "fp16 matrix OPERANDS only, every tensor stored f32 (floor)": dict(round_ops=False, stream_f32_levels=set(), last_f32=False, operands_only=True),
    "fp16 operands only + f32 conv_out": dict(round_ops=False, stream_f32_levels=set(), last_f32=True, operands_only=True),
    "bf16 storage everywhere": dict(round_ops=True, stream_f32_levels=set(), last_f32=False, dt=torch.bfloat16),
}


def to_u8(img):
    return ((img / 2 + 0.5).clamp(0, 1) * 255).round().to(torch.uint8)


def main():
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    gold = torch.load(os.path.join(ROOT, "tests", "golden", "sd_cfg2_stages_autocast.pt"), weights_only=False)
    vsd = SO.seeded_weights(SO.vae_shapes(), GG.VAE_SEED)
    z = gold["den"].reshape(1, 4, 64, 64) / SO.SCALE
    want = [a for a in sys.argv[1:]]
    rows, ref = [], None
    with torch.no_grad():
        for name, kw in VARIANTS.items():
            if ref is not None and want and not any(w in name for w in want):
                continue
            t0 = time.time()
            img = decode(vsd, z, **kw)
            if ref is None:
                ref = img
                assert float((img - SO.vae_decode(vsd, z)).abs().max()) < 1e-4     # the replay IS the oracle's decoder
            u, u0 = to_u8(img), to_u8(ref)
            d = (u.int() - u0.int()).abs()
            row = dict(variant=name, image_rel_l2=float((img - ref).norm() / ref.norm()), mean_abs_err_lsb=float(((img - ref).abs() * 127.5).mean()),
                       u8_changed_512=float((d > 0).float().mean()), u8_max_diff=int(d.max()),
                       u8_changed_at_F=float((u[:, :, ::8, ::8] != u0[:, :, ::8, ::8]).float().mean()),   # nearest 512 -> 64 keeps pixel floor(8 i)
                       seconds=time.time() - t0)
            rows.append(row)
            print("%-62s image rel-L2 %.2e | mean |err| %.3f LSB | uint8 changed: %.2f %% at 512^2, %.2f %% at F (max diff %d)   [%.0f s]"
                  % (name, row["image_rel_l2"], row["mean_abs_err_lsb"], 100 * row["u8_changed_512"], 100 * row["u8_changed_at_F"], row["u8_max_diff"], row["seconds"]), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "r05_vae_decoder_storage.json"), "w") as f:
        json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
