"""GPU: the MX block-scaled fp8 path (BASELINE configs[4] "fp8 MFMA on CDNA4"): OCP e4m3 elements with one E8M0 scale per 32 K
elements, v_mfma_scale_f32_16x16x128_f8f6f4.  The quantiser is checked bit for bit against a torch restatement of the OCP MX
v1.0 conversion (float8_e4m3fn, round to nearest even, saturating); the GEMM is checked (a) exactly on integer data, (b) against
the f32 product of the DEQUANTISED operands (only the accumulation order differs), and (c) against the unquantised product with
the stated fp8 tolerance."""
import math
import os
import sys

import pytest
import torch

from conftest import margin, rel_l2

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu
FP8_TOL = 6e-2        # MX e4m3 (3 mantissa bits, shared power-of-two scale per 32): measured 3.6e-2 .. 3.9e-2 rel-L2 on N(0,1) operands


def stream():
    return torch.cuda.current_stream().cuda_stream


def mx_quant_ref(x):
    """x (rows,K) float -> (e4m3 bytes (rows,K) uint8, E8M0 bytes (rows,K/32) uint8, dequantised float)"""
    rows, K = x.shape
    xb = x.float().reshape(rows, K // 32, 32)
    am = xb.abs().amax(dim=-1)
    e = torch.where(am > 0, torch.floor(torch.log2(am.double())).float() - 8, torch.full_like(am, -127.0)).clamp(-127, 126)
    # exact floor(log2) from the exponent bits (log2 in floating point can be off by one at powers of two)
    bits = am.view(torch.int32)
    e_bits = torch.where(am > 0, ((bits >> 23) & 0xff).float() - 127 - 8, torch.full_like(am, -127.0)).clamp(-127, 126)
    e = e_bits
    scaled = (xb * torch.exp2(-e)[..., None]).clamp(-448, 448)
    q = scaled.to(torch.float8_e4m3fn)
    deq = q.float() * torch.exp2(e)[..., None]
    return q.view(torch.uint8).reshape(rows, K), (e + 127).to(torch.uint8), deq.reshape(rows, K)


@pytest.mark.parametrize("rows,K", [(7, 128), (300, 640), (4096, 1280)])
def test_quant_mx_matches_ocp_reference(ctx, rows, K):
    g = torch.Generator(device="cuda").manual_seed(rows + K)
    x = (torch.randn(rows, K, device="cuda", generator=g) * torch.exp2(torch.randint(-6, 7, (rows, 1), device="cuda", generator=g).float())).to(torch.bfloat16)
    x[0, :32] = 0                                              # an all-zero block
    x[1, 5] = 448.0 * 2 ** 3                                    # a block whose maximum sits on e4m3's largest value
    q = torch.empty(rows, K, device="cuda", dtype=torch.uint8)
    sc = torch.empty(rows, K // 32, device="cuda", dtype=torch.uint8)
    ctx.check(ctx.lib.svg_op_quant_mx(ctx.h, x.data_ptr(), q.data_ptr(), sc.data_ptr(), rows, K, stream()), "quant_mx")
    q_ref, sc_ref, _ = mx_quant_ref(x.float())
    assert torch.equal(sc, sc_ref)
    # e4m3 has +0 / -0: compare values, not bit patterns of zero
    same = (q == q_ref) | (((q & 0x7f) == 0) & ((q_ref & 0x7f) == 0))
    assert same.all(), int((~same).sum())


def test_gemm_fp8_integer_exact(ctx):
    """integers in [-8, 8]: every block quantises exactly (amax 8 -> elements x * 32 <= 256, three mantissa bits suffice) and every
    product / partial sum is an exact f32: the GEMM equals the integer product bit for bit — operand and scale lane maps included."""
    g = torch.Generator(device="cuda").manual_seed(2)
    for (M, N, K) in [(128, 128, 128), (300, 320, 640), (1000, 64, 256), (257, 644, 1280)]:
        A = torch.randint(-8, 9, (M, K), device="cuda", generator=g).float()
        W = torch.randint(-8, 9, (N, K), device="cuda", generator=g).float()
        A[:, :32] *= 0.5                                       # blocks with different shared exponents along K and across rows
        W[::3, 32:64] *= 4
        A, W = A.to(torch.bfloat16), W.to(torch.bfloat16)
        out = torch.empty(M, N, device="cuda", dtype=torch.float32)
        ctx.check(ctx.lib.svg_op_gemm_fp8(ctx.h, A.data_ptr(), W.data_ptr(), None, None, out.data_ptr(), M, N, K, 0, 1, stream()), "gemm_fp8")
        ref = A.double() @ W.double().t()
        assert torch.equal(out.double(), ref), (M, N, K, float((out.double() - ref).abs().max()))


@pytest.mark.parametrize("M,N,K,act", [(4096, 640, 640, 0), (1024, 1280, 2560, 0), (7168, 1280, 1280, 1), (513, 324, 768, 2)])
def test_gemm_fp8(ctx, M, N, K, act):
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    A = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda", generator=g) / math.sqrt(K)).to(torch.bfloat16)
    b = torch.randn(N, device="cuda", generator=g)
    R = torch.randn(M, N, device="cuda", generator=g).to(torch.bfloat16)
    out = torch.empty(M, N, device="cuda", dtype=torch.float32)
    ctx.check(ctx.lib.svg_op_gemm_fp8(ctx.h, A.data_ptr(), W.data_ptr(), b.data_ptr(), R.data_ptr(), out.data_ptr(), M, N, K, act, 1, stream()), "gemm_fp8")
    _, _, Ad = mx_quant_ref(A.float())
    _, _, Wd = mx_quant_ref(W.float())
    f = {0: lambda t: t, 1: torch.nn.functional.silu, 2: torch.nn.functional.gelu}[act]
    ref_q = f(Ad @ Wd.t() + b + R.float())
    ref = f(A.float() @ W.float().t() + b + R.float())
    assert rel_l2(out, ref_q) < 2e-5                            # same quantised operands: accumulation order only
    pre = lambda t: t - b - R.float() if act == 0 else t        # the product itself (bias and residual would mask its error)
    margin("MX fp8 GEMM %dx%dx%d vs unquantised product" % (M, N, K), rel_l2(pre(out), pre(ref)), FP8_TOL)
    outb = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ctx.check(ctx.lib.svg_op_gemm_fp8(ctx.h, A.data_ptr(), W.data_ptr(), b.data_ptr(), R.data_ptr(), outb.data_ptr(), M, N, K, act, 0, stream()), "gemm_fp8")
    assert rel_l2(outb.float(), out) < 4e-3


def _conv_mx(ctx, x, w, b, r, f16=False, mode=0):
    """x (B,H,W,Cin) 16-bit NHWC, w (Cout,Cin,3,3) f32, optional bias / residual -> out (B,H,W,Cout) 16-bit, quantised activations + scales"""
    B, H, W, Cin = x.shape
    Cout = w.shape[0]
    Cp = (Cin + 127) // 128 * 128
    up = 2 if mode == 3 else 1
    out = torch.empty(B, up * H, up * W, Cout, device="cuda", dtype=x.dtype)
    q = torch.empty(B * H * W, Cp, device="cuda", dtype=torch.uint8)
    sc = torch.empty(B * H * W, Cp // 32, device="cuda", dtype=torch.uint8)
    fn = ctx.lib.svg_op_conv3x3_mx_f16 if f16 else ctx.lib.svg_op_conv3x3_mx
    ctx.check(fn(ctx.h, x.data_ptr(), w.data_ptr(), b.data_ptr() if b is not None else None, r.data_ptr() if r is not None else None,
                 out.data_ptr(), q.data_ptr(), sc.data_ptr(), B, H, W, Cin, Cout, mode, stream()), "conv3x3_mx")
    return out, q, sc


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(12, 64, 64, 128, 128), (6, 64, 64, 320, 320), (10, 32, 32, 640, 644), (28, 16, 16, 1280, 1280), (3, 128, 128, 192, 160)])
def test_conv3x3_mx_integer_exact(ctx, B, H, W, Cin, Cout):
    """integers in [-8, 8] with per-block power-of-two factors: every block quantises exactly and every partial sum is an exact f32, so
    the MX fp8 conv equals F.conv2d bit for bit — patch / weight / scale DMA maps, the lane map of v_mfma_scale_f32_16x16x128_f8f6f4,
    the 64-channel tail chunk (Cin = 320, 192: the second half of the last 128-channel chunk is padding), image borders, ragged
    channel tiles (Cout = 644) and the residual epilogue included."""
    g = torch.Generator(device="cuda").manual_seed(B + H + Cin + Cout)
    x = torch.randint(-8, 9, (B, H, W, Cin), device="cuda", generator=g).float()
    w = torch.randint(-8, 9, (Cout, Cin, 3, 3), device="cuda", generator=g).float()
    x[..., :32] *= 0.5                                       # different shared exponents along the channels, per pixel row and per tap
    x[:, ::3, :, 32:64] *= 4
    w[::5, 64:96] *= 0.25
    w[:, :, 1, 1] *= 2
    keep = (torch.rand(Cout, Cin, 3, 3, device="cuda", generator=g) < 0.06).float()       # sparse: sums stay below 2^24 at K = 11 520
    w = w * keep
    res = torch.randint(-64, 65, (B, H, W, Cout), device="cuda", generator=g).float()
    bias = torch.randint(-16, 17, (Cout,), device="cuda", generator=g).float()
    for dt, f16 in ((torch.bfloat16, False), (torch.float16, True)):
        out, _, _ = _conv_mx(ctx, x.to(dt), w, bias, res.to(dt), f16)
        ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), bias.double(), padding=1).permute(0, 2, 3, 1) + res.double()
        assert float(ref.abs().max()) < 2 ** 24
        want = ref.float().to(dt)                              # the only rounding: the 16-bit store
        assert torch.equal(out, want), (dt, float((out.double() - want.double()).abs().max()))


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(28, 8, 8, 1280, 1280), (12, 16, 16, 640, 640), (6, 32, 32, 192, 320)])
def test_conv3x3_mx_upsample_integer_exact(ctx, B, H, W, Cin, Cout):
    """the fused nearest-2x upsample in front of the MX fp8 conv (the UNet's upsamplers under fp8=1): exact on integer data against
    F.conv2d(F.interpolate(x, scale_factor=2))."""
    g = torch.Generator(device="cuda").manual_seed(B + H + Cin)
    x = torch.randint(-8, 9, (B, H, W, Cin), device="cuda", generator=g).float()
    x[..., 32:64] *= 0.5
    w = torch.randint(-8, 9, (Cout, Cin, 3, 3), device="cuda", generator=g).float() * (torch.rand(Cout, Cin, 3, 3, device="cuda", generator=g) < 0.06).float()
    bias = torch.randint(-16, 17, (Cout,), device="cuda", generator=g).float()
    out, _, _ = _conv_mx(ctx, x.to(torch.float16), w, bias, None, True, mode=3)
    xu = torch.nn.functional.interpolate(x.permute(0, 3, 1, 2).double(), scale_factor=2.0, mode="nearest")
    ref = torch.nn.functional.conv2d(xu, w.double(), bias.double(), padding=1).permute(0, 2, 3, 1)
    assert out.shape == ref.shape and float(ref.abs().max()) < 2 ** 15
    assert torch.equal(out, ref.float().to(torch.float16))


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(6, 64, 64, 320, 320), (12, 32, 32, 1280, 640)])
def test_conv3x3_mx_random(ctx, B, H, W, Cin, Cout):
    """N(0,1)-like operands: the quantised activations equal the OCP MX conversion of mx_quant_ref bit for bit (padding channels zero),
    the conv equals F.conv2d of the DEQUANTISED operands up to the accumulation order, and sits at the stated MX-fp8 distance from the
    unquantised conv."""
    g = torch.Generator(device="cuda").manual_seed(Cin + Cout)
    x = torch.randn(B, H, W, Cin, device="cuda", generator=g).to(torch.float16)
    w = torch.randn(Cout, Cin, 3, 3, device="cuda", generator=g) / math.sqrt(9 * Cin)
    bias = torch.randn(Cout, device="cuda", generator=g)
    out, q, sc = _conv_mx(ctx, x, w, bias, None, True)
    P, Cp = B * H * W, (Cin + 127) // 128 * 128
    q_ref, sc_ref, xd = mx_quant_ref(x.reshape(P, Cin).float())
    assert torch.equal(sc[:, :Cin // 32], sc_ref)
    same = (q[:, :Cin] == q_ref) | (((q[:, :Cin] & 0x7f) == 0) & ((q_ref & 0x7f) == 0))
    assert same.all() and (q[:, Cin:] == 0).all()
    # weights: blocks of 32 input channels per (output channel, tap)
    wt = w.permute(0, 2, 3, 1).reshape(Cout * 9, Cin)
    _, _, wd = mx_quant_ref(wt)
    wd = wd.reshape(Cout, 3, 3, Cin).permute(0, 3, 1, 2)
    conv = lambda a, ww: torch.nn.functional.conv2d(a.reshape(B, H, W, Cin).permute(0, 3, 1, 2).double(), ww.double(), bias.double(), padding=1).permute(0, 2, 3, 1)
    ref_q, ref = conv(xd, wd), conv(x.float(), w)
    assert rel_l2(out.float(), ref_q) < 6e-4                    # the fp16 store of the output (2^-11) is all that differs
    margin("MX fp8 conv3x3 %dx%dx%d Cin %d -> %d vs the unquantised conv" % (B, H, W, Cin, Cout), rel_l2(out.float() - bias, ref - bias), FP8_TOL)


def test_unet_step_fp8_conv_full_size(ctx):
    """configs[4] arithmetic where it pays (round 4): the resnets' 3x3 convs of the full-size UNet in MX fp8 (fp8=1, fp16 storage for the
    rest) against the fp32 oracle and against the fp16 path of the same library."""
    from oracle import sd_oracle as SO
    from sd_video_gen_amd import _lib
    usd = SO.seeded_weights(SO.unet_shapes(), 31)
    c = SO.SD_UNET
    g = torch.Generator().manual_seed(1)
    x = torch.randn(7, 4, 64, 64, generator=g)
    cc = torch.randn(7, 77, 768, generator=g)
    t = torch.tensor([980.0, 860.0, 700.0, 500.0, 320.0, 120.0, 0.0])
    outs = {}
    for fp8 in (0, 1):
        ctx.configure(_lib.SVG_UNET, block_out=list(c["block_out"]), layers=2, heads=8, ctx_dim=768, groups=32, attn=list(c["attn"]), fp8=fp8, f16=1)
        ctx.load_state_dict(_lib.SVG_UNET, usd)
        assert ctx.finalize(_lib.SVG_UNET) == 859_520_964
        outs[fp8] = ctx.unet_forward(x.cuda(), t.cuda(), cc.cuda()).cpu()
    assert not torch.equal(outs[0], outs[1])
    worst = 0.0
    for b in (0, 3, 6):
        ref = SO.unet_forward(usd, x[b:b + 1], float(t[b]), cc[b:b + 1])
        e = rel_l2(outs[1][b:b + 1], ref)
        print("[parity] fp8 convs, sample %d (t=%d): %.3e (fp16 path %.3e)" % (b, int(t[b]), e, rel_l2(outs[0][b:b + 1], ref)))
        worst = max(worst, e)
    margin("full-size UNet call, MX fp8 3x3 convs (batch 7), worst sample vs the fp32 oracle", worst, 9e-2)
    margin("full-size UNet call, MX fp8 convs vs the fp16 path", rel_l2(outs[1], outs[0]), 9e-2)


def test_unet_step_fp8_full_size(ctx, monkeypatch):
    """configs[4]: the full-size SD v1.4 UNet with fp8=1 (out-projections, ff.net.2, proj_out, 1x1 shortcuts, cross k at the
    32 x 32 level and below in MX fp8) against the fp32 oracle, and against the bf16 path of the same library."""
    from oracle import sd_oracle as SO
    from sd_video_gen_amd import _lib
    monkeypatch.setenv("SVG_FP8_PROJ", "1")                     # the round-2/3 placement (projections), off by default since round 4
    monkeypatch.setenv("SVG_FP8_CONV", "0")
    _lib.env_refresh()
    usd = SO.seeded_weights(SO.unet_shapes(), 31)
    c = SO.SD_UNET
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 4, 64, 64, generator=g)
    cc = torch.randn(2, 77, 768, generator=g)
    t = torch.tensor([500.0, 40.0])
    outs = {}
    for fp8 in (0, 1):
        ctx.configure(_lib.SVG_UNET, block_out=list(c["block_out"]), layers=2, heads=8, ctx_dim=768, groups=32, attn=list(c["attn"]), fp8=fp8)
        ctx.load_state_dict(_lib.SVG_UNET, usd)
        assert ctx.finalize(_lib.SVG_UNET) == 859_520_964
        outs[fp8] = ctx.unet_forward(x.cuda(), t.cuda(), cc.cuda()).cpu()
    ref = torch.cat([SO.unet_forward(usd, x[b:b + 1], float(t[b]), cc[b:b + 1]) for b in range(2)])
    e16 = rel_l2(outs[0], ref)
    margin("full-size UNet call, bf16 (same inputs)", e16, 2.5e-2)
    margin("full-size UNet call with fp8=1 vs the fp32 oracle", rel_l2(outs[1], ref), 7.4e-2)
    margin("full-size UNet call, fp8=1 vs bf16 path", rel_l2(outs[1], outs[0]), 7.5e-2)
    assert not torch.equal(outs[0], outs[1])                    # the fp8 projections really ran
