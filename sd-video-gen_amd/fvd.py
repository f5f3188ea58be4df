"""FVD evaluation — the host side of the reference's ``evaluation/fvd_2.py`` (itself from VideoGPT): same function names and
contracts, the numerics in libsvg_hip.so (I3D forward ``svg_i3d_forward`` / ``svg_fvd_logits``, ``svg_frechet_distance``).

  load_i3d_pretrained(device)   fvd_2.py:91-97 — the reference reads ./models/i3d_pretrained_400.pt (not in its tree:
                                .MISSING_LARGE_BLOBS); here that file when present, else $SVG_I3D_WEIGHTS, else — explicit opt-in like
                                the SD networks — seeded synthetic weights of the exact architecture
  get_fvd_logits(videos, i3d)   fvd_2.py:16-19: uint8 (b,t,h,w,c) -> preprocess -> logits (b,400)
  frechet_distance(x1, x2)      fvd_2.py:66-78
  all_gather(tensor)            fvd_2.py:103-107: the one collective the reference ever sketched; rides on sharding.gather_rows
"""
import os

import numpy as np
import torch

from . import _lib


def i3d_shapes(num_classes=400):
    """state_dict names -> shapes of evaluation/pytorch_i3d.py's InceptionI3d (without the num_batches_tracked scalars)"""
    mixed = [("Mixed_3b", 192, [64, 96, 128, 16, 32, 32]), ("Mixed_3c", 256, [128, 128, 192, 32, 96, 64]),
             ("Mixed_4b", 480, [192, 96, 208, 16, 48, 64]), ("Mixed_4c", 512, [160, 112, 224, 24, 64, 64]),
             ("Mixed_4d", 512, [128, 128, 256, 24, 64, 64]), ("Mixed_4e", 512, [112, 144, 288, 32, 64, 64]),
             ("Mixed_4f", 528, [256, 160, 320, 32, 128, 128]), ("Mixed_5b", 832, [256, 160, 320, 32, 128, 128]),
             ("Mixed_5c", 832, [384, 192, 384, 48, 128, 128])]
    s = {}

    def unit(p, cin, cout, k, bn=True):
        s[p + ".conv3d.weight"] = (cout, cin, k, k, k)
        if bn:
            for n in ("weight", "bias", "running_mean", "running_var"):
                s[p + ".bn." + n] = (cout,)
        else:
            s[p + ".conv3d.bias"] = (cout,)
    unit("Conv3d_1a_7x7", 3, 64, 7)
    unit("Conv3d_2b_1x1", 64, 64, 1)
    unit("Conv3d_2c_3x3", 64, 192, 3)
    for name, cin, oc in mixed:
        unit(name + ".b0", cin, oc[0], 1)
        unit(name + ".b1a", cin, oc[1], 1)
        unit(name + ".b1b", oc[1], oc[2], 3)
        unit(name + ".b2a", cin, oc[3], 1)
        unit(name + ".b2b", oc[3], oc[4], 3)
        unit(name + ".b3b", cin, oc[5], 1)
    unit("logits", 1024, num_classes, 1, bn=False)
    return s


class I3D:
    """stands where the reference's ``i3d`` (InceptionI3d(400).eval()) stands: ``i3d(batch)`` -> logits"""

    def __init__(self, state_dict, ctx=None, num_classes=400, source="given"):
        self.ctx = ctx or _lib.default_context()
        self.num_classes, self.source = num_classes, source
        sd = {k: v for k, v in state_dict.items() if "num_batches_tracked" not in k}
        missing = [k for k in i3d_shapes(num_classes) if k not in sd]
        if missing:
            raise KeyError("I3D weights: missing %s (%d more)" % (missing[0], len(missing) - 1))
        self.ctx.configure(_lib.SVG_I3D, num_classes=num_classes)
        self.ctx.load_state_dict(_lib.SVG_I3D, sd)
        self.n_params = self.ctx.finalize(_lib.SVG_I3D)
        self.ctx.claim(_lib.SVG_I3D, self)

    def _check(self):
        if self.ctx.owner(_lib.SVG_I3D) is not self:
            raise RuntimeError("the I3D slot of this library context now holds another model's weights")

    def __call__(self, x):
        self._check()
        return self.ctx.i3d_forward(x, self.num_classes)

    def eval(self):
        return self


def load_i3d_pretrained(device=None, ctx=None, seed=0):
    from .sd_utils import synthetic_allowed
    for p in ("./models/i3d_pretrained_400.pt", os.environ.get("SVG_I3D_WEIGHTS", "")):
        if p and os.path.exists(p):
            return I3D(torch.load(p, map_location="cpu", weights_only=True), ctx, source=p)
    if not synthetic_allowed():
        raise FileNotFoundError("./models/i3d_pretrained_400.pt (fvd_2.py:94) not found; set $SVG_I3D_WEIGHTS or opt in to seeded "
                                "synthetic weights (SVG_ALLOW_SYNTHETIC_WEIGHTS=1 — FVD values are then not comparable with published ones)")
    from . import sd_layout
    import math
    import zlib
    sd = {}
    for name, shape in i3d_shapes().items():
        g = torch.Generator().manual_seed((seed * 1000003 + zlib.crc32(name.encode())) % (2 ** 31))
        if name.endswith("conv3d.weight"):
            sd[name] = torch.randn(shape, generator=g) * (1.4 / math.sqrt(shape[1] * shape[2] * shape[3] * shape[4]))
        elif name.endswith("running_var"):
            sd[name] = 0.5 + torch.rand(shape, generator=g)
        elif name.endswith("running_mean"):
            sd[name] = 0.1 * torch.randn(shape, generator=g)
        elif name.endswith("bn.weight"):
            sd[name] = 1.0 + 0.1 * torch.randn(shape, generator=g)
        else:
            sd[name] = 0.05 * torch.randn(shape, generator=g)
    return I3D(sd, ctx, source="synthetic")


def get_fvd_logits(videos, i3d, device=None):
    """videos: uint8 (b,t,h,w,c) numpy array / tensor, b a multiple of 16 in the reference (fvd_2.py:81: batches of 16)"""
    v = torch.as_tensor(np.asarray(videos) if not isinstance(videos, torch.Tensor) else videos)
    assert v.shape[0] % 16 == 0, "fvd_2.get_logits asserts batches of 16 clips"
    i3d._check()
    return torch.cat([i3d.ctx.fvd_logits(v[i:i + 16], i3d.num_classes) for i in range(0, v.shape[0], 16)], dim=0)


def preprocess(videos, target_resolution=224, ctx=None):
    """fvd_2.py:7-14: uint8 (b,t,h,w,c) -> float (b,c,t,224,224) in [-1, 1] (shorter side scaled bilinearly, centre crop), on device"""
    v = torch.as_tensor(np.asarray(videos) if not isinstance(videos, torch.Tensor) else videos)
    assert target_resolution == 224, "the I3D of the FVD takes 224 x 224 crops (fvd_2.py:7)"
    return (ctx or _lib.default_context()).fvd_preprocess(v)


def get_logits(i3d, videos, device=None):
    """fvd_2.py:81-89: preprocessed clips (b,c,t,h,w), b a multiple of 16 -> logits, in batches of 16"""
    assert videos.shape[0] % 16 == 0
    i3d._check()
    return torch.cat([i3d(videos[i:i + 16]) for i in range(0, videos.shape[0], 16)], dim=0)


class InceptionI3d(I3D):
    """evaluation/pytorch_i3d.py:136 by name: ``InceptionI3d(400, in_channels=3)`` then ``load_state_dict`` then call.  The library
    model is configured when the weights arrive."""

    def __init__(self, num_classes=400, spatial_squeeze=True, final_endpoint="Logits", name="inception_i3d", in_channels=3, dropout_keep_prob=0.5):
        if in_channels != 3 or final_endpoint != "Logits":
            raise ValueError("the FVD path uses the RGB InceptionI3d up to its logits")
        self.num_classes, self.source, self.ctx, self.n_params = num_classes, "unset", None, 0

    def load_state_dict(self, state_dict, strict=True):
        I3D.__init__(self, state_dict, self.ctx, self.num_classes, source="state_dict")
        return self

    def to(self, device):
        return self

    def _check(self):
        if self.ctx is None:
            raise RuntimeError("InceptionI3d: load_state_dict() has not been called")
        I3D._check(self)


def frechet_distance(x1, x2, ctx=None):
    return (ctx or _lib.default_context()).frechet_distance(x1, x2)


def all_gather(tensor):
    """fvd_2.py:103-107: every rank's (n, 400) logits, concatenated in rank order"""
    from . import sharding
    return sharding.gather_rows(tensor)
