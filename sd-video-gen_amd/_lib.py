"""ctypes binding of libsvg_hip.so (include/svg_hip.h).

There is no CPU fallback: if the shared library is missing or cannot be loaded the import of any
product entry point raises.  PyTorch is used only for device memory and streams
(``tensor.data_ptr()``, ``torch.cuda.current_stream().cuda_stream``).
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SVG_LIB") or os.path.join(_HERE, "libsvg_hip.so")   # $SVG_LIB: A/B another build of the same ABI

SVG_TRANSFORMER, SVG_VAE, SVG_UNET, SVG_CLIP_TEXT, SVG_MINILM, SVG_I3D = 0, 1, 2, 3, 4, 5
SVG_ERR_RUNTIME, SVG_ERR_INVALID = -1, -2        # enum svg_status

_lib = None

_vp, _i, _f, _i64 = C.c_void_p, C.c_int, C.c_float, C.c_int64

SVG_TENSOR_PARAM, SVG_TENSOR_GRAD, SVG_TENSOR_EXP_AVG, SVG_TENSOR_EXP_AVG_SQ = 0, 1, 2, 3   # enum svg_tensor_kind


class TrainCfg(C.Structure):
    """struct svg_train_cfg (include/svg_hip.h)."""
    _fields_ = [("frames_to_predict", _i), ("feat_h", _i), ("feat_w", _i),
                ("w_mse", _f), ("w_l1", _f), ("w_gdl", _f), ("gdl_alpha", _f), ("w_contrastive", _f), ("temperature", _f),
                ("dropout_p", _f), ("seed", C.c_uint64)]



# name -> argtypes (restype is int unless listed in _RESTYPES); mirrors include/svg_hip.h
SIGNATURES = {
    "svg_create": [_i, C.POINTER(_vp)],
    "svg_destroy": [_vp],
    "svg_last_error": [_vp],
    "svg_version": [],
    "svg_env_refresh": [],
    "svg_model_configure": [_vp, _i, C.c_char_p],
    "svg_load_weight": [_vp, _i, C.c_char_p, _vp, C.POINTER(_i64), _i],
    "svg_finalize": [_vp, _i, C.POINTER(_i64)],
    "svg_transformer_forward": [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp],
    "svg_transformer_forward_text": [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp],
    "svg_transformer_forward_padded": [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "svg_transformer_loss": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _i, _vp, _vp],
    "svg_transformer_forward_train": [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _f, C.c_uint64, _vp, _vp],
    "svg_transformer_adam_step": [_vp, _f, _f, _f, _f, _vp],
    "svg_transformer_tensor": [_vp, _i, C.c_char_p, _vp, _i64, _vp],
    "svg_clip_text_forward": [_vp, _vp, _i, _i, _vp, _vp],
    "svg_minilm_encode": [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp],
    "svg_i3d_forward": [_vp, _vp, _i, _i, _i, _i, _vp, _vp],
    "svg_fvd_logits": [_vp, _vp, _i, _i, _i, _i, _vp, _vp],
    "svg_fvd_preprocess": [_vp, _vp, _i, _i, _i, _i, _vp, _vp],
    "svg_frechet_distance": [_vp, _vp, _i, _vp, _i, _i, C.POINTER(C.c_double), _vp],
    "svg_vae_encode": [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp],
    "svg_vae_decode": [_vp, _vp, _i, _i, _i, _vp, _i, _i, _vp, _vp],
    "svg_unet_forward": [_vp, _vp, _i, _i, _i, _vp, _vp, _i, _vp, _vp],
    "svg_ddim_loop": [_vp, _vp, _i, _i, _i, _vp, _i, _i, _i, _f, _vp, _vp, _vp],
    "svg_ddim_step": [_vp, _vp, _vp, _vp, _i64, _i, _i, _vp],
    "svg_resize_bilinear_f32": [_vp, _vp, _i, _i, _i, _vp, _i, _i, _vp],
    "svg_resize_nearest_u8": [_vp, _vp, _i, _i, _i, _i, _vp, _i, _i, _vp],
    "svg_op_gemm": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "svg_op_conv3x3": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "svg_op_conv3x3_gn": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _i, C.POINTER(_i), _vp],
    "svg_op_gemm_lnstats": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, C.POINTER(_i), _vp],
    "svg_op_gemm_cat": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "svg_op_ff_fused": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp],
    "svg_op_xattn_fused": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _vp],
    "svg_op_dropout_mask": [_vp, C.c_uint64, _i, _f, _vp, _i64, _vp],
    "svg_op_conv3x3_mx": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "svg_op_quant_mx": [_vp, _vp, _vp, _vp, _i64, _i, _vp],
    "svg_op_gemm_fp8": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "svg_op_groupnorm": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp],
    "svg_op_layernorm": [_vp, _vp, _vp, _vp, _vp, _i, _i, _f, _vp],
    "svg_op_attention": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i64, _i64, _i64, _i64, _f, _vp],
    "svg_op_xf_gemm": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "svg_prof_enable": [_vp, _i],
    "svg_prof_reset": [_vp],
    "svg_prof_report": [_vp, C.c_char_p, _i],
    "svg_workspace_bytes": [_vp],
    "svg_transformer_status": [_vp],
    "svg_workspace_growths": [_vp],
    "svg_reserve_workspace": [_vp, _i64],
    "svg_plan_begin": [_vp],
    "svg_plan_end": [_vp, C.POINTER(_i64)],
    "svg_debug_captures_active": [],
}
# fp16-storage twins of the 16-bit operator hooks (svg_op_<name>_f16: same arguments)
for _n in ("gemm", "conv3x3", "conv3x3_gn", "conv3x3_mx", "gemm_lnstats", "gemm_cat", "ff_fused", "xattn_fused", "quant_mx", "gemm_fp8", "groupnorm", "layernorm",
           "attention"):
    SIGNATURES["svg_op_%s_f16" % _n] = SIGNATURES["svg_op_" + _n]
SIGNATURES["svg_model_dtype"] = [_vp, _i]
_RESTYPES = {"svg_destroy": None, "svg_env_refresh": None, "svg_model_dtype": C.c_char_p, "svg_last_error": C.c_char_p, "svg_version": C.c_char_p,
             "svg_workspace_bytes": _i64, "svg_workspace_growths": _i64}


def host_cpu_quota():
    """(cpus, limited): CPUs this process may burn — the affinity mask capped by the cgroup's CFS bandwidth (cpu.max = "quota
    period") — and whether a finite CFS quota is in force at all (cpu.max = "max ..." or no cgroup file: not limited)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    limited = False
    for path in (os.environ.get("SVG_CGROUP_CPU_MAX", "/sys/fs/cgroup/cpu.max"),):      # (the override exists for the host tests)
        try:
            q, p = open(path).read().split()
            if q != "max":
                n = min(n, max(1, int(q) // int(p)))
                limited = True
        except (OSError, ValueError):
            pass
    return n, limited


def fit_host_threads():
    """Caps torch's intra-op pool to a quarter of the CPU quota WHEN the cgroup sets a finite CFS quota.  torch sizes the pool from
    the VISIBLE cores (128 threads on the 256-core GPU host) while the cgroup grants 16 CPUs per 100 ms CFS period; the pool's idle
    workers spin, the group burns its 1.6 CPU-seconds in ~37 ms and the kernel then parks EVERY thread of the process — including
    the one inside hipLaunchKernel — until the next 100-ms period (profiles/r03_stall_trace.json, r03_throttle_before.txt /
    r03_throttle_after.txt: the round-2 "100-ms-tick stall", 28 -> 94 training it/s).  Without a quota (cpu.max = "max") nothing
    throttles and the pool is left alone.  An explicit OMP_NUM_THREADS / $SVG_HOST_THREADS wins.  The quota is shared by the ranks
    of THIS node: LOCAL_WORLD_SIZE (torch.distributed.run sets it), never WORLD_SIZE — a multi-node launch must not over-divide."""
    if os.environ.get("OMP_NUM_THREADS"):
        return torch.get_num_threads()
    explicit = int(os.environ.get("SVG_HOST_THREADS", "0") or 0)
    if explicit:
        if torch.get_num_threads() > explicit:
            torch.set_num_threads(explicit)
        return torch.get_num_threads()
    cpus, limited = host_cpu_quota()
    if not limited:
        return torch.get_num_threads()
    ranks = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1") or 1))
    want = max(1, cpus // (4 * ranks))
    if torch.get_num_threads() > want:
        torch.set_num_threads(want)
    return torch.get_num_threads()


def load():
    """dlopen the library and declare every symbol of the header; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    fit_host_threads()
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libsvg_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C sd-video-gen_amd/csrc`. There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, _i)
    _lib = lib
    return lib


def env_refresh():
    """the library caches its $SVG_* knobs per name: call after changing one in-process (svg_env_refresh)"""
    load().svg_env_refresh()


def source_hash():
    """hash of the kernel sources this library was built from (svg_version(): '... src <hash>')"""
    return load().svg_version().decode().rsplit(" ", 1)[-1]


def _ptr(t):
    if t is None:
        return None
    if isinstance(t, torch.Tensor):
        assert t.is_contiguous(), "tensor handed to the HIP library must be contiguous"
        return t.data_ptr()
    return t


def _stream():
    return torch.cuda.current_stream().cuda_stream


class Context:
    """One svg_ctx: a set of model slots + one workspace arena on one GPU.  Calls on a context are serialised by the caller; several contexts
    may be driven from different host threads at once (predict.sample_clips_streams) — the library's rules for that are in INTEGRATION.md
    ("Ownership and threading")."""

    def __init__(self, device_index=None):
        if not torch.cuda.is_available():
            raise RuntimeError("the sd-video-gen HIP path needs a GPU (gfx950); no CPU fallback exists")
        self.lib = load()
        if device_index is None:
            device_index = torch.cuda.current_device()
        self.device = torch.device("cuda", device_index)
        torch.cuda.set_device(self.device)
        h = _vp()
        if self.lib.svg_create(device_index, C.byref(h)) != 0:
            raise RuntimeError("svg_create: " + self.lib.svg_last_error(None).decode())
        self.h = h
        self._owner = {}          # model slot -> weakref of the host object whose weights the slot holds

    # ---- slot ownership --------------------------------------------------------------------------
    # A context has ONE slot per model id; svg_model_configure replaces what is in it.  Host objects that upload
    # weights claim the slot and check the claim before every forward, so two models sharing a context re-upload
    # (latent Transformers: the parameters live on the host) or refuse (SD networks) instead of computing with each
    # other's weights.
    def claim(self, model, obj):
        import weakref
        self._owner[model] = weakref.ref(obj)

    def owner(self, model):
        r = self._owner.get(model)
        return r() if r is not None else None

    def close(self):
        if getattr(self, "h", None):
            self.lib.svg_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc, what):
        if rc != 0:
            msg = self.lib.svg_last_error(self.h).decode()
            # SVG_ERR_INVALID: arguments / shapes / call order (ValueError, like torch's shape errors in the reference);
            # anything else is a HIP failure
            raise (ValueError if rc == SVG_ERR_INVALID else RuntimeError)("%s: %s" % (what, msg))

    # ---- models ------------------------------------------------------------------------------
    def configure(self, model, **kv):
        s = ";".join("%s=%s" % (k, ",".join(str(int(x)) for x in (v if isinstance(v, (list, tuple)) else [v])))
                     for k, v in kv.items())
        if model == SVG_VAE:
            self.vae_down = 2 ** (len(kv.get("block_out", (128, 256, 512, 512))) - 1)
        self._owner.pop(model, None)
        self.check(self.lib.svg_model_configure(self.h, model, s.encode()), "svg_model_configure")

    def load_state_dict(self, model, sd):
        for name, t in sd.items():
            if not torch.is_floating_point(t):
                continue
            t = t.detach().to(torch.float32).contiguous()
            shape = (_i64 * max(t.dim(), 1))(*(list(t.shape) if t.dim() else [1]))
            self.check(self.lib.svg_load_weight(self.h, model, name.encode(), t.data_ptr(), shape, max(t.dim(), 1)),
                       "svg_load_weight(%s)" % name)

    def model_dtype(self, model):
        """'bf16' / 'fp16' for the SD networks, 'f32' for the Transformer / CLIP, None when the slot is empty"""
        r = self.lib.svg_model_dtype(self.h, model)
        return r.decode() if r else None

    def finalize(self, model):
        n = _i64(0)
        self.check(self.lib.svg_finalize(self.h, model, C.byref(n)), "svg_finalize")
        return n.value

    # ---- hot path ----------------------------------------------------------------------------
    @staticmethod
    def _pad_bias(pad, B, T, device):
        """key-padding mask (B,T) -> additive f32 bias: bool True -> -inf (as torch canonicalises it), float -> as is"""
        if pad is None:
            return None
        pad = torch.as_tensor(pad).to(device)
        if tuple(pad.shape) != (B, T):
            raise ValueError("key-padding mask must be (batch, sequence) = (%d, %d), got %s" % (B, T, tuple(pad.shape)))
        if pad.dtype == torch.bool:
            return torch.zeros((B, T), device=device, dtype=torch.float32).masked_fill(pad, float("-inf")).contiguous()
        return pad.float().contiguous()

    def transformer_forward(self, src, tgt, mask=None, pe_row=None, text=None, src_pad_mask=None, tgt_pad_mask=None):
        B, Ts, D = src.shape
        Tt = tgt.shape[1]
        if src_pad_mask is not None or tgt_pad_mask is not None:
            src = src.contiguous().float()
            tgt_c = src if tgt is src else tgt.contiguous().float()
            out = torch.empty((Tt, B, D), device=src.device, dtype=torch.float32)
            mask = mask.contiguous().float() if mask is not None else None
            pe_row = pe_row.to(device=src.device, dtype=torch.int32).contiguous() if pe_row is not None else None
            text = text.to(device=src.device, dtype=torch.float32).contiguous() if text is not None else None
            sp, tp = self._pad_bias(src_pad_mask, B, Ts, src.device), self._pad_bias(tgt_pad_mask, B, Tt, src.device)
            self.check(self.lib.svg_transformer_forward_padded(self.h, _ptr(src), _ptr(tgt_c), _ptr(text), B, Ts, Tt, _ptr(mask), _ptr(sp),
                                                               _ptr(tp), _ptr(pe_row), _ptr(out), _stream()), "svg_transformer_forward_padded")
            return out
        src = src.contiguous().float()
        tgt_c = src if tgt is src else tgt.contiguous().float()
        out = torch.empty((Tt, B, D), device=src.device, dtype=torch.float32)
        mask = mask.contiguous().float() if mask is not None else None
        pe_row = pe_row.to(device=src.device, dtype=torch.int32).contiguous() if pe_row is not None else None
        if text is not None:
            text = text.to(device=src.device, dtype=torch.float32).contiguous()
            assert text.shape[0] == B
            self.check(self.lib.svg_transformer_forward_text(self.h, _ptr(src), _ptr(tgt_c), _ptr(text), B, Ts, Tt, _ptr(mask),
                                                             _ptr(pe_row), _ptr(out), _stream()), "svg_transformer_forward_text")
            return out
        self.check(self.lib.svg_transformer_forward(self.h, _ptr(src), _ptr(tgt_c), B, Ts, Tt, _ptr(mask), _ptr(pe_row),
                                                    _ptr(out), _stream()), "svg_transformer_forward")
        return out

    def transformer_status(self, sync=True):
        """raises RuntimeError when a layer-walking Transformer forward on this device gave up since the last check (its output is
        NaN-filled: re-issue it).  sync: wait for the current stream first, so that the forward whose result is about to be read is covered."""
        if sync:
            torch.cuda.current_stream().synchronize()
        self.check(self.lib.svg_transformer_status(self.h), "svg_transformer_status")

    # ---- training step of the latent Transformer ---------------------------------------------------
    def transformer_loss(self, cfg, src, tgt, expected, mask=None, text=None, backward=True, read_losses=True):
        """-> dict(total, mse, l1, gdl, contrastive).  backward=True: train mode, gradients left in the library.
        read_losses=False: nothing is copied back and the stream is not synchronised (returns None)."""
        B, Ts, _ = src.shape
        Tt = tgt.shape[1]
        src = src.contiguous().float()
        tgt = tgt.contiguous().float()
        expected = expected.contiguous().float()
        assert expected.shape == tgt.shape
        mask = mask.contiguous().float() if mask is not None else None
        text = text.to(device=src.device, dtype=torch.float32).contiguous() if text is not None else None
        out = (_f * 5)() if read_losses else None
        self.check(self.lib.svg_transformer_loss(self.h, C.byref(cfg), _ptr(src), _ptr(tgt), _ptr(expected), _ptr(text), B, Ts, Tt,
                                                 _ptr(mask), int(bool(backward)), out, _stream()), "svg_transformer_loss")
        if out is None:
            return None
        return dict(zip(("total", "mse", "l1", "gdl", "contrastive"), [float(v) for v in out]))

    def transformer_forward_train(self, src, tgt, mask=None, text=None, dropout_p=0.1, seed=0):
        """train-mode forward (dropout active) -> (Tt, B, D_lat)"""
        B, Ts, D = src.shape
        Tt = tgt.shape[1]
        src = src.contiguous().float()
        tgt = tgt.contiguous().float()
        mask = mask.contiguous().float() if mask is not None else None
        text = text.to(device=src.device, dtype=torch.float32).contiguous() if text is not None else None
        out = torch.empty((Tt, B, D), device=src.device, dtype=torch.float32)
        self.check(self.lib.svg_transformer_forward_train(self.h, _ptr(src), _ptr(tgt), _ptr(text), B, Ts, Tt, _ptr(mask), float(dropout_p),
                                                          int(seed) & (2 ** 64 - 1), _ptr(out), _stream()), "svg_transformer_forward_train")
        return out

    def transformer_adam_step(self, lr, betas=(0.9, 0.999), eps=1e-8):
        self.check(self.lib.svg_transformer_adam_step(self.h, lr, betas[0], betas[1], eps, _stream()), "svg_transformer_adam_step")

    def transformer_tensor(self, name, like, kind=SVG_TENSOR_PARAM):
        """Copy of parameter `name` (or its gradient / Adam moment) shaped like `like`, on the CPU."""
        out = torch.empty(tuple(like.shape), dtype=torch.float32)
        torch.cuda.synchronize()      # training steps may have run on another (non-blocking) stream than the current one
        self.check(self.lib.svg_transformer_tensor(self.h, kind, name.encode(), out.data_ptr(), out.numel(), _stream()),
                   "svg_transformer_tensor(%s)" % name)
        return out

    def dropout_mask(self, seed, site, p, n):
        out = torch.empty(n, device="cuda", dtype=torch.float32)
        self.check(self.lib.svg_op_dropout_mask(self.h, seed, site, p, _ptr(out), n, _stream()), "svg_op_dropout_mask")
        return out

    def clip_text_forward(self, input_ids, d_model=768):
        """input_ids (B,T) integer tensor -> (B,T,d_model) f32 last_hidden_state"""
        ids = input_ids.to(device=self.device, dtype=torch.int32).contiguous()
        B, T = ids.shape
        out = torch.empty((B, T, d_model), device=self.device, dtype=torch.float32)
        self.check(self.lib.svg_clip_text_forward(self.h, _ptr(ids), B, T, _ptr(out), _stream()), "svg_clip_text_forward")
        return out

    def minilm_encode(self, input_ids, lengths, d_model=384, return_hidden=False):
        """input_ids (B,T), lengths (B) integer tensors -> (B,d_model) unit-norm sentence embeddings [, (B,T,d_model) hidden states]"""
        ids = input_ids.to(device=self.device, dtype=torch.int32).contiguous()
        lens = lengths.to(device=self.device, dtype=torch.int32).contiguous()
        B, T = ids.shape
        out = torch.empty((B, d_model), device=self.device, dtype=torch.float32)
        hid = torch.empty((B, T, d_model), device=self.device, dtype=torch.float32) if return_hidden else None
        self.check(self.lib.svg_minilm_encode(self.h, _ptr(ids), _ptr(lens), B, T, _ptr(out), _ptr(hid), _stream()), "svg_minilm_encode")
        return (out, hid) if return_hidden else out

    # ---- FVD evaluation ---------------------------------------------------------------------------
    def i3d_forward(self, x, num_classes=400):
        """x (B,3,T,224,224) f32 in [-1,1] -> I3D logits (B,num_classes)"""
        x = x.to(self.device).contiguous().float()
        B, c, T, H, W = x.shape
        assert c == 3
        out = torch.empty((B, num_classes), device=self.device, dtype=torch.float32)
        self.check(self.lib.svg_i3d_forward(self.h, _ptr(x), B, T, H, W, _ptr(out), _stream()), "svg_i3d_forward")
        return out

    def fvd_logits(self, videos_u8, num_classes=400):
        """videos (B,T,H,W,3) uint8 -> I3D logits (B,num_classes) (fvd_2.get_fvd_logits: preprocess + network)"""
        v = torch.as_tensor(videos_u8).to(self.device).contiguous()
        assert v.dtype == torch.uint8 and v.dim() == 5 and v.shape[-1] == 3
        B, T, H, W, _ = v.shape
        out = torch.empty((B, num_classes), device=self.device, dtype=torch.float32)
        self.check(self.lib.svg_fvd_logits(self.h, _ptr(v), B, T, H, W, _ptr(out), _stream()), "svg_fvd_logits")
        return out

    def fvd_preprocess(self, videos_u8):
        v = torch.as_tensor(videos_u8).to(self.device).contiguous()
        B, T, H, W, _ = v.shape
        out = torch.empty((B, 3, T, 224, 224), device=self.device, dtype=torch.float32)
        self.check(self.lib.svg_fvd_preprocess(self.h, _ptr(v), B, T, H, W, _ptr(out), _stream()), "svg_fvd_preprocess")
        return out

    def frechet_distance(self, x1, x2):
        """(n1,d), (n2,d) embeddings -> float (fvd_2.frechet_distance)"""
        x1 = x1.to(self.device).flatten(1).contiguous().float()
        x2 = x2.to(self.device).flatten(1).contiguous().float()
        out = C.c_double(0.0)
        self.check(self.lib.svg_frechet_distance(self.h, _ptr(x1), x1.shape[0], _ptr(x2), x2.shape[0], x1.shape[1], C.byref(out), _stream()),
                   "svg_frechet_distance")
        return float(out.value)

    def vae_encode(self, img_u8, H=None, W=None, eps=None, return_moments=False):
        """img_u8: (N,h,w,3) uint8 on device; nearest-resized to (H,W) when given."""
        N, sh, sw, c = img_u8.shape
        assert c == 3 and img_u8.dtype == torch.uint8
        H = H or sh
        W = W or sw
        img_u8 = img_u8.contiguous()
        f = getattr(self, "vae_down", 8)
        z = torch.empty((N, 4, H // f, W // f), device=img_u8.device, dtype=torch.float32)
        mom = torch.empty((N, 8, H // f, W // f), device=img_u8.device, dtype=torch.float32) if return_moments else None
        eps = eps.contiguous().float() if eps is not None else None
        self.check(self.lib.svg_vae_encode(self.h, _ptr(img_u8), N, sh, sw, H, W, _ptr(eps), _ptr(z), _ptr(mom), _stream()),
                   "svg_vae_encode")
        return (z, mom) if return_moments else z

    def vae_decode(self, z, out_hw=None, return_float=False):
        N, c, h, w = z.shape
        assert c == 4
        z = z.contiguous().float()
        f = getattr(self, "vae_down", 8)
        oh, ow = out_hw if out_hw else (f * h, f * w)
        img = torch.empty((N, oh, ow, 3), device=z.device, dtype=torch.uint8)
        fo = torch.empty((N, 3, f * h, f * w), device=z.device, dtype=torch.float32) if return_float else None
        self.check(self.lib.svg_vae_decode(self.h, _ptr(z), N, h, w, _ptr(img), oh, ow, _ptr(fo), _stream()), "svg_vae_decode")
        return (img, fo) if return_float else img

    def unet_forward(self, x, timesteps, ctx_emb):
        N, c, h, w = x.shape
        x = x.contiguous().float()
        t = torch.as_tensor(timesteps, dtype=torch.float32, device=x.device).reshape(-1)
        if t.numel() == 1:
            t = t.repeat(N)
        t = t.contiguous()
        ctx_emb = ctx_emb.contiguous().float()
        out = torch.empty_like(x)
        self.check(self.lib.svg_unet_forward(self.h, _ptr(x), N, h, w, _ptr(t), _ptr(ctx_emb), ctx_emb.shape[1], _ptr(out),
                                             _stream()), "svg_unet_forward")
        return out

    def ddim_loop(self, z, text_emb, num_steps=50, start_step=0, guidance=7.5, noise=None, return_hist=False):
        N, c, h, w = z.shape
        z = z.contiguous().float().clone()
        text_emb = text_emb.contiguous().float()
        assert text_emb.shape[0] == 2 * N, "text embeddings must be [uncond; cond] (2N rows)"
        noise = noise.contiguous().float() if noise is not None else None
        hist = torch.empty(((num_steps - start_step + 1) * N, c, h, w), device=z.device, dtype=torch.float32) if return_hist else None
        self.check(self.lib.svg_ddim_loop(self.h, _ptr(z), N, h, w, _ptr(text_emb), text_emb.shape[1], num_steps, start_step,
                                          float(guidance), _ptr(noise), _ptr(hist), _stream()), "svg_ddim_loop")
        return hist if return_hist else z

    def ddim_step(self, x, eps, t, t_prev):
        x = x.contiguous().float()
        eps = eps.contiguous().float()
        out = torch.empty_like(x)
        self.check(self.lib.svg_ddim_step(self.h, _ptr(x), _ptr(eps), _ptr(out), x.numel(), int(t), int(t_prev), _stream()),
                   "svg_ddim_step")
        return out

    def resize_bilinear_f32(self, x, oh, ow):
        """(N,C,h,w) f32 -> (N,C,oh,ow), F.interpolate(mode='bilinear') semantics (evaluation/predict_fvd.py:165)"""
        N, Cc, h, w = x.shape
        x = x.contiguous().float()
        out = torch.empty((N, Cc, oh, ow), device=x.device, dtype=torch.float32)
        self.check(self.lib.svg_resize_bilinear_f32(self.h, _ptr(x), N * Cc, h, w, _ptr(out), oh, ow, _stream()), "svg_resize_bilinear_f32")
        return out

    def resize_nearest_u8(self, img, oh, ow):
        N, h, w, c = img.shape
        img = img.contiguous()
        out = torch.empty((N, oh, ow, c), device=img.device, dtype=torch.uint8)
        self.check(self.lib.svg_resize_nearest_u8(self.h, _ptr(img), N, h, w, c, _ptr(out), oh, ow, _stream()), "svg_resize")
        return out

    # ---- profiling -----------------------------------------------------------------------------
    def prof_enable(self, on=True, detail=False):
        """on: hipEvent brackets per kernel family; detail: additionally per call-site signature (names starting with '@')"""
        self.lib.svg_prof_enable(self.h, (2 if detail else 1) if on else 0)

    def prof_reset(self):
        self.check(self.lib.svg_prof_reset(self.h), "svg_prof_reset")

    def prof_report(self):
        buf = C.create_string_buffer(1 << 20)
        self.check(self.lib.svg_prof_report(self.h, buf, len(buf)), "svg_prof_report")
        out = {}
        for line in buf.value.decode().strip().splitlines():
            name, calls, ms, flops, nbytes = line.split()
            out[name] = dict(calls=int(calls), ms=float(ms), flops=float(flops), bytes=float(nbytes))
        return out

    def workspace_bytes(self):
        return int(self.lib.svg_workspace_bytes(self.h))

    def workspace_growths(self):
        """(re)allocations of the workspace since the context was created: constant once the workload has been planned"""
        return int(self.lib.svg_workspace_growths(self.h))

    def reserve_workspace(self, nbytes):
        self.check(self.lib.svg_reserve_workspace(self.h, int(nbytes)), "svg_reserve_workspace")

    def planning(self):
        """``with ctx.planning() as plan:`` — the model calls inside run their planning pass only (nothing is launched, the returned
        tensors are NOT written); on exit the workspace is sized once for the largest of them (``plan.bytes``)."""
        return _Planning(self)


class _Planning:
    def __init__(self, ctx):
        self.ctx, self.bytes = ctx, None

    def __enter__(self):
        self.ctx.check(self.ctx.lib.svg_plan_begin(self.ctx.h), "svg_plan_begin")
        return self

    def __exit__(self, et, ev, tb):
        n = _i64(0)
        rc = self.ctx.lib.svg_plan_end(self.ctx.h, C.byref(n))
        self.bytes = n.value
        if et is None:
            self.ctx.check(rc, "svg_plan_end")
        return False


def captures_active():
    """capture windows open inside the library right now (svg_debug_captures_active)"""
    return int(load().svg_debug_captures_active())


_default_ctx = {}


def default_context():
    """Process-wide context for the current GPU."""
    idx = torch.cuda.current_device() if torch.cuda.is_available() else -1
    if idx not in _default_ctx:
        _default_ctx[idx] = Context(idx if idx >= 0 else None)
    return _default_ctx[idx]
