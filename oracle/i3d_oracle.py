"""ORACLE — TEST INFRASTRUCTURE ONLY (imported by tests/; never by the product).

CPU restatement of the FVD evaluation the reference runs at the end of its text loop (prediction/predict_text.py:155-168,
289-314 and evaluation/predict_fvd*.py): the Inception-v1 I3D network of evaluation/pytorch_i3d.py:136-322 (logits of
16-frame 224 x 224 clips) and the Fréchet distance of evaluation/fvd_2.py:7-78.  Follows, by line:

  pytorch_i3d.py:43-109    Unit3D: TF 'SAME' zero padding computed from the input size (compute_pad), Conv3d without padding,
                           BatchNorm3d (eval: running statistics, eps 1e-5), ReLU
  pytorch_i3d.py:8-40      MaxPool3dSamePadding: the same zero padding, then max pooling (the zeros take part in the max)
  pytorch_i3d.py:112-133   InceptionModule: cat([b0, b1b(b1a), b2b(b2a), b3b(maxpool 3x3x3)], dim=1)
  pytorch_i3d.py:178-301   the layer list; :303-312 forward: avg_pool [2,7,7] stride 1, logits 1x1x1 conv (bias, no BN / ReLU),
                           squeeze, mean over time
  fvd_2.py:7-14,109-136    preprocess: /255, bilinear resize (align_corners False) of the shorter side to 224, centre crop, -0.5, x2
  fvd_2.py:22-78           _symmetric_matrix_square_root (SVD, eps 1e-10), trace_sqrt_product, cov (unbiased), frechet_distance

PINNED: oracle/gen_golden_i3d.py imports the live reference modules (torch + numpy only: they import in the build container),
loads seeded weights made by `seeded_i3d_weights` into the reference InceptionI3d and records its logits, and runs the
reference's preprocess / frechet_distance on seeded inputs -> tests/golden/i3d_fvd.pt; tests/test_oracle_i3d.py checks this file
against it.  The pretrained i3d_pretrained_400.pt is not in the reference tree (.MISSING_LARGE_BLOBS)."""
import math

import torch
import torch.nn.functional as F

MIXED = [("Mixed_3b", 192, [64, 96, 128, 16, 32, 32]), ("Mixed_3c", 256, [128, 128, 192, 32, 96, 64]),
         ("Mixed_4b", 480, [192, 96, 208, 16, 48, 64]), ("Mixed_4c", 512, [160, 112, 224, 24, 64, 64]),
         ("Mixed_4d", 512, [128, 128, 256, 24, 64, 64]), ("Mixed_4e", 512, [112, 144, 288, 32, 64, 64]),
         ("Mixed_4f", 528, [256, 160, 320, 32, 128, 128]), ("Mixed_5b", 832, [256, 160, 320, 32, 128, 128]),
         ("Mixed_5c", 832, [384, 192, 384, 48, 128, 128])]
# (name, kind, ...) in execution order (pytorch_i3d.py:153-172)
LAYERS = [("Conv3d_1a_7x7", "unit", 3, 64, (7, 7, 7), (2, 2, 2)), ("MaxPool3d_2a_3x3", "pool", (1, 3, 3), (1, 2, 2)),
          ("Conv3d_2b_1x1", "unit", 64, 64, (1, 1, 1), (1, 1, 1)), ("Conv3d_2c_3x3", "unit", 64, 192, (3, 3, 3), (1, 1, 1)),
          ("MaxPool3d_3a_3x3", "pool", (1, 3, 3), (1, 2, 2)), ("Mixed_3b", "mixed"), ("Mixed_3c", "mixed"),
          ("MaxPool3d_4a_3x3", "pool", (3, 3, 3), (2, 2, 2)), ("Mixed_4b", "mixed"), ("Mixed_4c", "mixed"), ("Mixed_4d", "mixed"),
          ("Mixed_4e", "mixed"), ("Mixed_4f", "mixed"), ("MaxPool3d_5a_2x2", "pool", (2, 2, 2), (2, 2, 2)), ("Mixed_5b", "mixed"),
          ("Mixed_5c", "mixed")]


def i3d_shapes(num_classes=400, in_channels=3):
    s = {}

    def unit(p, cin, cout, k, bn=True, bias=False):
        s[p + ".conv3d.weight"] = (cout, cin) + tuple(k)
        if bias:
            s[p + ".conv3d.bias"] = (cout,)
        if bn:
            for n in ("weight", "bias", "running_mean", "running_var"):
                s[p + ".bn." + n] = (cout,)
    for L in LAYERS:
        if L[1] == "unit":
            unit(L[0], in_channels if L[0] == "Conv3d_1a_7x7" else L[2], L[3], L[4])
    for name, cin, oc in MIXED:
        unit(name + ".b0", cin, oc[0], (1, 1, 1))
        unit(name + ".b1a", cin, oc[1], (1, 1, 1))
        unit(name + ".b1b", oc[1], oc[2], (3, 3, 3))
        unit(name + ".b2a", cin, oc[3], (1, 1, 1))
        unit(name + ".b2b", oc[3], oc[4], (3, 3, 3))
        unit(name + ".b3b", cin, oc[5], (1, 1, 1))
    unit("logits", 1024, num_classes, (1, 1, 1), bn=False, bias=True)
    return s


def seeded_i3d_weights(seed, num_classes=400):
    """seeded weights of the exact architecture (the Kinetics checkpoint is not available): convs N(0, 1.4 / sqrt(fan_in)), BatchNorm
    scale 1 + N(0, .1), shift N(0, .05), running mean N(0, .1), running var in [0.5, 1.5]; per tensor from (seed, name)"""
    import zlib
    sd = {}
    for name, shape in i3d_shapes(num_classes).items():
        g = torch.Generator().manual_seed((seed * 1000003 + zlib.crc32(name.encode())) % (2 ** 31))
        if name.endswith("conv3d.weight"):
            fan = shape[1] * shape[2] * shape[3] * shape[4]
            sd[name] = torch.randn(shape, generator=g) * (1.4 / math.sqrt(fan))
        elif name.endswith("running_var"):
            sd[name] = 0.5 + torch.rand(shape, generator=g)
        elif name.endswith("running_mean"):
            sd[name] = 0.1 * torch.randn(shape, generator=g)
        elif name.endswith("bn.weight"):
            sd[name] = 1.0 + 0.1 * torch.randn(shape, generator=g)
        else:
            sd[name] = 0.05 * torch.randn(shape, generator=g)
    return sd


def _same_pad(x, kernel, stride):
    """compute_pad of Unit3D / MaxPool3dSamePadding (pytorch_i3d.py:10-36,72-95): zero padding, front = total // 2"""
    pads = []
    for d in (2, 1, 0):                                   # F.pad order: W, H, T
        size = x.shape[2 + d]
        k, s = kernel[d], stride[d]
        p = max(k - s, 0) if size % s == 0 else max(k - (size % s), 0)
        pads += [p // 2, p - p // 2]
    return F.pad(x, pads)


def unit3d(sd, p, x, kernel, stride=(1, 1, 1), bn=True, relu=True):
    x = _same_pad(x, kernel, stride)
    x = F.conv3d(x, sd[p + ".conv3d.weight"], sd.get(p + ".conv3d.bias"), stride=stride)
    if bn:
        x = F.batch_norm(x, sd[p + ".bn.running_mean"], sd[p + ".bn.running_var"], sd[p + ".bn.weight"], sd[p + ".bn.bias"], False, 0.0, 1e-5)
    return F.relu(x) if relu else x


def maxpool_same(x, kernel, stride):
    return F.max_pool3d(_same_pad(x, kernel, stride), kernel, stride)


def inception(sd, p, x):
    b0 = unit3d(sd, p + ".b0", x, (1, 1, 1))
    b1 = unit3d(sd, p + ".b1b", unit3d(sd, p + ".b1a", x, (1, 1, 1)), (3, 3, 3))
    b2 = unit3d(sd, p + ".b2b", unit3d(sd, p + ".b2a", x, (1, 1, 1)), (3, 3, 3))
    b3 = unit3d(sd, p + ".b3b", maxpool_same(x, (3, 3, 3), (1, 1, 1)), (1, 1, 1))
    return torch.cat([b0, b1, b2, b3], dim=1)


def i3d_forward(sd, x, upto=None):
    """x (B,3,T,224,224) f32 in [-1,1] -> logits (B,400) (pytorch_i3d.py:303-312); upto: stop after that endpoint (feature map)"""
    for L in LAYERS:
        if L[1] == "unit":
            x = unit3d(sd, L[0], x, L[4], L[5])
        elif L[1] == "pool":
            x = maxpool_same(x, L[2], L[3])
        else:
            x = inception(sd, L[0], x)
        if upto == L[0]:
            return x
    x = F.avg_pool3d(x, (2, 7, 7), (1, 1, 1))
    x = unit3d(sd, "logits", x, (1, 1, 1), bn=False, relu=False)
    return x.squeeze(3).squeeze(3).mean(dim=2)


# ---- fvd_2.py ------------------------------------------------------------------------------------------------------------------
def preprocess(videos_u8, resolution=224):
    """(b,t,h,w,c) uint8 tensor -> (b,c,t,224,224) f32 in [-1,1] (fvd_2.py:7-14,109-136)"""
    out = []
    for video in videos_u8:
        v = video.permute(0, 3, 1, 2).float() / 255.0
        t, c, h, w = v.shape
        scale = resolution / min(h, w)
        size = (resolution, math.ceil(w * scale)) if h < w else (math.ceil(h * scale), resolution)
        v = F.interpolate(v, size=size, mode="bilinear", align_corners=False)
        t, c, h, w = v.shape
        ws, hs = (w - resolution) // 2, (h - resolution) // 2
        v = v[:, :, hs:hs + resolution, ws:ws + resolution].permute(1, 0, 2, 3).contiguous() - 0.5
        out.append(v)
    return torch.stack(out) * 2


def _sym_sqrt(mat, eps=1e-10):
    u, s, v = torch.svd(mat)
    si = torch.where(s < eps, s, torch.sqrt(s))
    return u @ torch.diag(si) @ v.t()


def cov(m):
    """fvd_2.py:35-63 with rowvar=False: unbiased covariance of the rows' variables"""
    m = m.t().clone()
    fact = 1.0 / (m.size(1) - 1)
    m = m - m.mean(dim=1, keepdim=True)
    return fact * m @ m.t()


def frechet_distance(x1, x2):
    x1, x2 = x1.flatten(1), x2.flatten(1)
    m, mw = x1.mean(0), x2.mean(0)
    sigma, sigma_w = cov(x1), cov(x2)
    s = _sym_sqrt(sigma)
    tr = torch.trace(_sym_sqrt(s @ sigma_w @ s))
    return torch.trace(sigma + sigma_w) - 2.0 * tr + torch.sum((m - mw) ** 2)


def fvd_test_embeddings(seed=5):
    """seeded (n, 400) embedding sets of the golden fixture: rank-deficient pairs (n < 400) and a full-rank one"""
    g = torch.Generator().manual_seed(seed)
    e1 = torch.randn(64, 400, generator=g) * 2.0 + 0.3
    e2 = torch.randn(48, 400, generator=g) * 1.5
    e3 = torch.randn(600, 400, generator=g) @ (torch.randn(400, 400, generator=g) / 20.0)
    return e1, e2, e3
