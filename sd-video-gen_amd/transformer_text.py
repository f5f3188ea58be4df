"""Host-side mirror of the reference's text-conditioned latent Transformer (models/transformer_text.py:15-137,
BASELINE config 5): every token is ``cat(Linear(D_lat -> DIM_MODEL)(x), class_embedding_384) * sqrt(d)`` with
``d = DIM_MODEL + 384`` (transformer_text.py:33-35,82-92); the rest is the base model at width d.

The class-name embedding comes from ``SentenceTransformer('all-MiniLM-L6-v2').encode(cls_list)`` in the reference
(transformer_text.py:12,82-83): here ``self.sent_transformer`` is ``minilm.SentenceEncoder`` — the same BertModel + mean
pooling + L2 normalisation, run in the library (``svg_minilm_encode``), its parameters under the reference's checkpoint names
(``sent_transformer.0.auto_model.*``).  ``text_encoder=`` replaces it by a caller-supplied callable; ``text_encoder="hash"``
is a seeded per-string stand-in for host-only tests (equal strings -> equal unit-norm vectors; no network behind it)."""
import zlib

import torch
import torch.nn as nn

from . import _lib
from .config import parse_config_args
from .minilm import SentenceEncoder
from .transformer import LibraryTraining, PositionalEncoding

TEXT_EMBED_DIM = 384


class Transformer(LibraryTraining, nn.Module):
    def __init__(self, num_tokens=0, dim_model=256, num_heads=8, num_encoder_layers=6, num_decoder_layers=6,
                 dropout_p=0.1, text_encoder=None, st_weights=None):
        super().__init__()
        self.config, self.args = parse_config_args()
        self.text_embed_dim = TEXT_EMBED_DIM
        self.dim_model = dim_model + self.text_embed_dim
        self.img_embed_dim = dim_model
        self.num_heads = num_heads
        self.num_encoder_layers = num_encoder_layers
        self.num_decoder_layers = num_decoder_layers
        self.height = self.width = self.config.FRAME_SIZE
        self.compression = 8
        self.d_lat = self.height // 8 * self.width // 8 * 4
        self.text_encoder = text_encoder          # callable(list[str]) -> (n, 384) array / tensor, "hash", or None = sent_transformer
        # transformer_text.py:44: self.sent_transformer = SentenceTransformer('all-MiniLM-L6-v2') (hub weights there; here
        # $SVG_MINILM_WEIGHTS, a checkpoint's sent_transformer.* entries, or the synthetic opt-in: minilm.SentenceEncoder)
        self.sent_transformer = SentenceEncoder(weights=st_weights)
        # parameter containers in the reference's construction order (transformer_text.py:48-69)
        self.positional_encoder = PositionalEncoding(dim_model=self.dim_model, dropout_p=dropout_p, max_len=64)
        self.project_image_embedding = nn.Linear(self.d_lat, self.img_embed_dim)
        self.transformer = nn.Transformer(d_model=self.dim_model, nhead=num_heads, num_encoder_layers=num_encoder_layers,
                                          num_decoder_layers=num_decoder_layers, dropout=dropout_p)
        self.out = nn.Linear(self.dim_model, self.d_lat)
        self._ctx = None
        self._uploaded_version = None

    def encode_classes(self, cls_list):
        if callable(self.text_encoder):
            return torch.as_tensor(self.text_encoder(list(cls_list)), dtype=torch.float32)
        if self.text_encoder is None:
            return self.sent_transformer.encode(list(cls_list))          # transformer_text.py:82
        if self.text_encoder != "hash":
            raise ValueError("text_encoder must be a callable, 'hash' or None")
        out = []
        for c in cls_list:
            g = torch.Generator().manual_seed(zlib.crc32(str(c).encode()) % (2 ** 31))
            v = torch.randn(TEXT_EMBED_DIM, generator=g)
            out.append(v / v.norm())
        return torch.stack(out)

    def _text_of(self, cls_list):
        if cls_list is None:
            raise ValueError("the text-conditioned model needs the class names (or a (B,384) tensor) of the batch")
        return cls_list if isinstance(cls_list, torch.Tensor) else self.encode_classes(cls_list)

    def use_context(self, ctx):
        self._bound_ctx = ctx
        self.sent_transformer._ctx = ctx
        self.sent_transformer._uploaded = None
        self._uploaded_version = None
        return self

    def _weights_version(self):
        return tuple(int(p._version) for p in self.parameters()) + (id(self._ctx),)

    def _sync_weights(self):
        ctx = getattr(self, "_bound_ctx", None) or _lib.default_context()
        # the context's Transformer slot may have been taken by another module (another checkpoint, the text variant)
        # since the last call: upload again unless the slot still holds THIS module's current parameters
        if self._ctx is ctx and ctx.owner(_lib.SVG_TRANSFORMER) is self and self._uploaded_version == self._weights_version():
            return ctx
        self._check_not_ahead()
        self._ctx = ctx
        ctx.configure(_lib.SVG_TRANSFORMER, d_lat=self.d_lat, d_model=self.dim_model, heads=self.num_heads,
                      enc_layers=self.num_encoder_layers, dec_layers=self.num_decoder_layers, text_dim=self.text_embed_dim,
                      ffn=self.transformer.encoder.layers[0].linear1.out_features if self.num_encoder_layers else 2048)
        ctx.load_state_dict(_lib.SVG_TRANSFORMER, {n: t for n, t in self.state_dict().items() if not n.startswith("sent_transformer.")})
        self.n_params = ctx.finalize(_lib.SVG_TRANSFORMER)
        ctx.claim(_lib.SVG_TRANSFORMER, self)
        self._uploaded_version = self._weights_version()
        return ctx

    def load_state_dict(self, state_dict, strict=True):
        """Reference checkpoints are ``model.state_dict()`` of models/transformer_text.py, whose ``sent_transformer`` attribute
        (transformer_text.py:44) is an nn.Module: they carry the MiniLM weights as ``sent_transformer.0.auto_model.*``.  Those fill
        ``self.sent_transformer`` (the class-name encoder); a checkpoint without them leaves the encoder as it is."""
        st = {n[len("sent_transformer."):]: t for n, t in state_dict.items() if n.startswith("sent_transformer.")}
        own = {n: t for n, t in state_dict.items() if not n.startswith("sent_transformer.")}
        if st:
            self.sent_transformer.load_state_dict(st)
        want = {n for n in super().state_dict() if not n.startswith("sent_transformer.")}
        missing, unexpected = sorted(want - set(own)), sorted(set(own) - want)
        if strict and (missing or unexpected):
            raise RuntimeError("Error(s) in loading state_dict for Transformer: missing keys %s, unexpected keys %s" % (missing, unexpected))
        nn.Module.load_state_dict(self, own, strict=False)
        self._uploaded_version = None
        self._lib_ahead = False
        return nn.modules.module._IncompatibleKeys(missing, unexpected)

    def forward(self, src, cls_list, tgt, tgt_mask=None, src_pad_mask=None, tgt_pad_mask=None, pe_row=None):
        """transformer_text.py:71-111.  ``cls_list``: class names (one per batch row) or a (B,384) tensor."""
        if not src.is_cuda:
            raise RuntimeError("Transformer.forward runs on the HIP library and needs CUDA tensors; there is no CPU fallback")
        txt = self._text_of(cls_list)
        if self.training and self.positional_encoder.dropout_p > 0:
            if src_pad_mask is not None or tgt_pad_mask is not None:
                raise NotImplementedError("key-padding masks are served in eval mode")
            return self._forward_train(src, tgt, tgt_mask, txt)
        ctx = self._sync_weights()
        return ctx.transformer_forward(src, tgt, tgt_mask, pe_row, text=txt, src_pad_mask=src_pad_mask, tgt_pad_mask=tgt_pad_mask)

    def get_tgt_mask(self, size):
        mask = torch.tril(torch.ones(size, size) == 1).float()
        mask = mask.masked_fill(mask == 0, float("-inf"))
        return mask.masked_fill(mask == 1, float(0.0))

    def create_pad_mask(self, matrix, pad_token):
        return matrix == pad_token


def predict(model, input_sequence, cls_list):
    """prediction/predict_text.py:48-74 -> (D_lat,)."""
    model.eval()
    with torch.no_grad():
        tgt_mask = model.get_tgt_mask(input_sequence.size(1)).to(input_sequence.device)
        pred = model(input_sequence, cls_list, input_sequence, tgt_mask).permute(1, 0, 2)
    return pred[0, -1]
