"""Host-side mirror of the reference's latent-sequence model (models/transformer.py:9-94).

Same constructor signature, same ``state_dict`` keys (the parameters are created
with ``nn.Linear`` / ``nn.Transformer`` / ``nn.Linear`` in the reference's order, so a seed
reproduces the reference's initial weights and its checkpoints load unchanged), same ``forward``
contract — but ``forward`` runs on the HIP library (f32-input MFMA weight-streaming kernels).
There is no CPU forward: calling it without a GPU / without libsvg_hip.so raises.
"""
import math

import torch
import torch.nn as nn

from . import _lib
from .config import parse_config_args


class PositionalEncoding(nn.Module):
    """models/positional_encoding.py:7-35 — buffer only; the add happens inside the HIP forward,
    indexed by BATCH row like the reference (quirk: pos_encoding[:x.size(0)] on a batch-first tensor)."""

    def __init__(self, dim_model, dropout_p, max_len):
        super().__init__()
        self.dropout_p = dropout_p
        pos_encoding = torch.zeros(max_len, dim_model)
        positions = torch.arange(0, max_len, dtype=torch.float).view(-1, 1)
        division = torch.exp(torch.arange(0, dim_model, 2).float() * (-math.log(10000.0)) / dim_model)
        pos_encoding[:, 0::2] = torch.sin(positions * division)
        pos_encoding[:, 1::2] = torch.cos(positions * division)
        self.register_buffer("pos_encoding", pos_encoding.unsqueeze(0).transpose(0, 1))


class LibraryTraining:
    """Training-step methods shared by the two host models (this file and transformer_text.py).  The optimisation runs inside
    the library on the uploaded copy of the weights (gradients and Adam moments live there); the nn.Parameters of the module
    are refreshed from it by pull_weights(), which state_dict() does on its own."""
    _train_seed = 0

    def _text_of(self, cls_list):
        return None

    def training_loss(self, cfg, new_batch, tgt_mask=None, cls_list=None, backward=True, read_losses=True):
        """Loss of one trainer iteration on the encoded batch `new_batch` (B, T, D_lat): src = new_batch, tgt = new_batch[:, :-1],
        expected = new_batch[:, 1:] (trainers/trainer.py:124-145; trainer_text.py passes the class names as well).  backward=True
        runs in train mode (dropout_p of `cfg`) and leaves the gradients in the library for adam_step(); False is the validation
        loss (eval mode).  -> dict of the loss terms."""
        if not new_batch.is_cuda:
            raise RuntimeError("the training step runs on the HIP library and needs CUDA tensors; there is no CPU fallback")
        ctx = self._sync_weights()
        y_input = new_batch[:, :-1]
        if tgt_mask is None:
            tgt_mask = self.get_tgt_mask(y_input.size(1)).to(new_batch.device)
        return ctx.transformer_loss(cfg, new_batch, y_input, new_batch[:, 1:], tgt_mask, self._text_of(cls_list), backward, read_losses)

    def _forward_train(self, src, tgt, tgt_mask, text):
        """model.train() forward: dropout with a fresh seed per call (the masks are a function of (seed, site, element))"""
        ctx = self._sync_weights()
        LibraryTraining._train_seed += 1
        return ctx.transformer_forward_train(src, tgt, tgt_mask, text, self.positional_encoder.dropout_p, LibraryTraining._train_seed)

    def adam_step(self, lr, betas=(0.9, 0.999), eps=1e-8):
        ctx = self._sync_weights()
        ctx.transformer_adam_step(lr, betas, eps)
        self._lib_ahead = True

    def pull_weights(self):
        """copies the library's (trained) weights back into this module's parameters"""
        if not getattr(self, "_lib_ahead", False):
            return
        ctx = self._ctx
        if ctx is None or ctx.owner(_lib.SVG_TRANSFORMER) is not self:
            raise RuntimeError("the library slot that held this module's trained weights was taken by another model before pull_weights()")
        with torch.no_grad():
            for name, p in self.named_parameters():
                if name.startswith("sent_transformer."):      # the frozen class-name encoder of the text variant: not trained
                    continue
                p.copy_(ctx.transformer_tensor(name, p).to(p.device))
        self._lib_ahead = False
        self._uploaded_version = self._weights_version()      # the copy in the library IS these parameters: no re-upload

    def grad_of(self, name):
        """gradient of parameter `name` from the last training_loss(backward=True), as a CPU tensor"""
        return self._ctx.transformer_tensor(name, dict(self.named_parameters())[name], _lib.SVG_TENSOR_GRAD)

    def state_dict(self, *a, **k):
        self.pull_weights()
        return super().state_dict(*a, **k)

    def _check_not_ahead(self):
        if getattr(self, "_lib_ahead", False):
            raise RuntimeError("this module's trained weights live in a library slot that was taken over (or its parameters were "
                               "modified on the host) before pull_weights(); call state_dict()/pull_weights() after training steps")


class Transformer(LibraryTraining, nn.Module):
    def __init__(self, num_tokens=0, dim_model=256, num_heads=8, num_encoder_layers=6,
                 num_decoder_layers=6, dropout_p=0.1):
        super().__init__()
        self.config, self.args = parse_config_args()          # transformer.py:23 (argv/cwd are API)
        self.dim_model = dim_model
        self.num_heads = num_heads
        self.num_encoder_layers = num_encoder_layers
        self.num_decoder_layers = num_decoder_layers
        self.height = self.config.FRAME_SIZE
        self.width = self.config.FRAME_SIZE
        self.compression = 8
        d_lat = self.height // self.compression * self.width // self.compression * 4
        self.d_lat = d_lat
        # parameter containers in the reference's construction order (transformer.py:33-45)
        self.positional_encoder = PositionalEncoding(dim_model=dim_model, dropout_p=dropout_p, max_len=64)
        self.embedding = nn.Linear(d_lat, dim_model)
        self.transformer = nn.Transformer(d_model=dim_model, nhead=num_heads, num_encoder_layers=num_encoder_layers,
                                          num_decoder_layers=num_decoder_layers, dropout=dropout_p)
        self.out = nn.Linear(dim_model, d_lat)
        self._ctx = None
        self._uploaded_version = None

    # ---- weight hand-over -------------------------------------------------------------------------
    def _weights_version(self):
        return tuple(int(p._version) for p in self.parameters()) + (id(self._ctx),)

    def use_context(self, ctx):
        """bind this module to a specific library context (default: the process-wide one of the current GPU)"""
        self._bound_ctx = ctx
        self._uploaded_version = None
        return self

    def _sync_weights(self):
        ctx = getattr(self, "_bound_ctx", None) or _lib.default_context()
        # the context's Transformer slot may have been taken by another module (another checkpoint, the text variant)
        # since the last call: upload again unless the slot still holds THIS module's current parameters
        if self._ctx is ctx and ctx.owner(_lib.SVG_TRANSFORMER) is self and self._uploaded_version == self._weights_version():
            return ctx
        self._check_not_ahead()
        self._ctx = ctx
        ctx.configure(_lib.SVG_TRANSFORMER, d_lat=self.d_lat, d_model=self.dim_model, heads=self.num_heads,
                      enc_layers=self.num_encoder_layers, dec_layers=self.num_decoder_layers,
                      ffn=self.transformer.encoder.layers[0].linear1.out_features if self.num_encoder_layers else 2048)
        ctx.load_state_dict(_lib.SVG_TRANSFORMER, self.state_dict())
        self.n_params = ctx.finalize(_lib.SVG_TRANSFORMER)
        ctx.claim(_lib.SVG_TRANSFORMER, self)
        self._uploaded_version = self._weights_version()
        return ctx

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self._uploaded_version = None
        self._lib_ahead = False
        return r

    # ---- forward (transformer.py:47-68) ---------------------------------------------------------------
    def forward(self, src, tgt, tgt_mask=None, src_pad_mask=None, tgt_pad_mask=None, pe_row=None):
        if not src.is_cuda:
            raise RuntimeError("Transformer.forward runs on the HIP library and needs CUDA tensors; "
                               "there is no CPU fallback")
        if self.training and self.positional_encoder.dropout_p > 0:
            if pe_row is not None:
                raise NotImplementedError("pe_row is a sampling-path argument (eval mode)")
            if src_pad_mask is not None or tgt_pad_mask is not None:
                raise NotImplementedError("key-padding masks are served in eval mode (no caller of the reference trains with one: trainer.py:141)")
            return self._forward_train(src, tgt, tgt_mask, None)
        ctx = self._sync_weights()
        # key-padding masks (transformer.py:64): bool (create_pad_mask) or float, (B,T)
        return ctx.transformer_forward(src, tgt, tgt_mask, pe_row, src_pad_mask=src_pad_mask, tgt_pad_mask=tgt_pad_mask)

    def get_tgt_mask(self, size):
        """transformer.py:70-89."""
        mask = torch.tril(torch.ones(size, size) == 1).float()
        mask = mask.masked_fill(mask == 0, float("-inf"))
        mask = mask.masked_fill(mask == 1, float(0.0))
        return mask

    def create_pad_mask(self, matrix, pad_token):
        return matrix == pad_token
