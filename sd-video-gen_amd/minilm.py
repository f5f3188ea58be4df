"""Class-name encoder of the text-conditioned latent Transformer: what ``SentenceTransformer('sentence-transformers/
all-MiniLM-L6-v2').encode(cls_list)`` computes in the reference (models/transformer_text.py:12,82-83) — WordPiece tokenisation on
the host, BertModel + masked mean pooling + L2 normalisation in libsvg_hip.so (``svg_minilm_encode``, f32).

``SentenceEncoder`` stands where ``Transformer.sent_transformer`` stands: an ``nn.Module`` whose parameters carry the names the
reference's checkpoints use (``sent_transformer.0.auto_model.<BertModel name>``; SentenceTransformer is an nn.Sequential of
[Transformer, Pooling, Normalize] and its Transformer module holds the BertModel as ``auto_model``), so a reference text-model
checkpoint fills it through ``load_state_dict`` and a checkpoint saved here loads in the reference under strict=True.

Weights (hub-only in the reference): a local HF-format directory in ``$SVG_MINILM_WEIGHTS`` (``model.safetensors`` or
``pytorch_model.bin``, ``config.json``, ``vocab.txt``), a checkpoint's ``sent_transformer.*`` entries, or — explicit opt-in like
the SD networks — seeded synthetic weights.  Without any of them ``encode`` raises like a failed ``from_pretrained``.  The
WordPiece vocabulary is a separate input (``vocab=`` / ``$SVG_MINILM_VOCAB`` / the weights directory): real weights without it
refuse to encode; the crc32 stand-in tokenizer serves the synthetic weights only.
"""
import os
import re
import zlib

import torch
import torch.nn as nn

from . import _lib

MINILM = dict(vocab=30522, d_model=384, heads=12, layers=6, ffn=1536, max_pos=512)
MAX_TOKENS = 128          # the library's sequence limit; all-MiniLM-L6-v2 truncates at 256, UCF-101 class names have < 10 tokens
PREFIX = "0.auto_model."


def bert_shapes(cfg=MINILM, pooler=True):
    d, f = cfg["d_model"], cfg["ffn"]
    s = {"embeddings.word_embeddings.weight": (cfg["vocab"], d), "embeddings.position_embeddings.weight": (cfg["max_pos"], d),
         "embeddings.token_type_embeddings.weight": (2, d), "embeddings.LayerNorm.weight": (d,), "embeddings.LayerNorm.bias": (d,)}
    for i in range(cfg["layers"]):
        p = "encoder.layer.%d." % i
        for n in ("attention.self.query", "attention.self.key", "attention.self.value", "attention.output.dense"):
            s[p + n + ".weight"] = (d, d)
            s[p + n + ".bias"] = (d,)
        s[p + "intermediate.dense.weight"] = (f, d)
        s[p + "intermediate.dense.bias"] = (f,)
        s[p + "output.dense.weight"] = (d, f)
        s[p + "output.dense.bias"] = (d,)
        for n in ("attention.output.LayerNorm", "output.LayerNorm"):
            s[p + n + ".weight"] = (d,)
            s[p + n + ".bias"] = (d,)
    if pooler:
        s["pooler.dense.weight"] = (d, d)
        s["pooler.dense.bias"] = (d,)
    return s


class StandInWordPiece:
    """BertTokenizer's role without the hub-only vocab.txt (synthetic weights only): lower-case, words and punctuation marks, one
    crc32-hashed id per piece in [1000, vocab), [CLS] = 101 ... [SEP] = 102, padding id 0 to the longest row."""

    def __init__(self, vocab=30522, max_length=MAX_TOKENS):
        self.vocab, self.max_length = vocab, max_length

    def __call__(self, sentences):
        rows = []
        for s in sentences:
            pieces = re.findall(r"[a-z0-9]+|[^\sa-z0-9]", str(s).lower())
            rows.append([101] + [1000 + zlib.crc32(w.encode()) % (self.vocab - 1000) for w in pieces][: self.max_length - 2] + [102])
        T = max(len(r) for r in rows)
        return (torch.tensor([r + [0] * (T - len(r)) for r in rows], dtype=torch.long), torch.tensor([len(r) for r in rows], dtype=torch.long))


class _HFWordPiece:
    """transformers.BertTokenizer over a local vocab.txt (host-side string processing, as in sentence-transformers)"""

    def __init__(self, vocab_file, max_length=MAX_TOKENS):
        from transformers import BertTokenizer
        self.tok = BertTokenizer(vocab_file, do_lower_case=True)
        self.max_length = max_length

    def __call__(self, sentences):
        enc = self.tok(list(sentences), padding=True, truncation=True, max_length=self.max_length, return_tensors="pt")
        return enc["input_ids"], enc["attention_mask"].sum(1)


class _Tree(nn.Module):
    """a bare module whose children / parameters are attached by dotted name"""


def _attach(root, dotted, tensor, buffer=False):
    parts = dotted.split(".")
    m = root
    for p in parts[:-1]:
        if p not in m._modules:
            m.add_module(p, _Tree())
        m = m._modules[p]
    if buffer:
        m.register_buffer(parts[-1], tensor)
    else:
        m.register_parameter(parts[-1], nn.Parameter(tensor, requires_grad=False))


class SentenceEncoder(nn.Module):
    def __init__(self, weights=None, cfg=None, seed=0, ctx=None, vocab=None, allow_standin_tokenizer=False):
        """``vocab``: path of a WordPiece ``vocab.txt`` (else ``$SVG_MINILM_VOCAB``, else ``vocab.txt`` inside
        ``$SVG_MINILM_WEIGHTS``) — independent of where the weights come from, so that a reference text checkpoint (which carries the
        MiniLM weights but no vocabulary) can be paired with it.  REAL weights (local directory, checkpoint, dict) without such a
        file refuse to encode: hashed stand-in ids into trained embeddings would silently differ from SentenceTransformer.encode.
        ``allow_standin_tokenizer=True`` (or ``$SVG_MINILM_STANDIN_TOKENIZER=1``) is the explicit opt-out for tests."""
        super().__init__()
        self.cfg = dict(MINILM, **(cfg or {}))
        self.loaded = False
        self.synthetic = False            # True only for the seeded stand-in weights: the one case the hash tokenizer is meant for
        self._ctx = ctx
        self._uploaded = None
        self.tokenizer = None
        self._allow_standin = bool(allow_standin_tokenizer) or os.environ.get("SVG_MINILM_STANDIN_TOKENIZER", "") not in ("", "0")
        d = os.environ.get("SVG_MINILM_WEIGHTS")
        from .sd_utils import synthetic_allowed
        if isinstance(weights, dict):
            self._fill(weights)
        elif d and weights in (None, "local"):
            self._fill(self._load_local(d))
        elif weights == "synthetic" or (weights is None and synthetic_allowed()):
            from . import sd_layout
            self._fill(sd_layout.seeded_weights(bert_shapes(self.cfg), seed), synthetic=True)
        if vocab and not os.path.exists(vocab):
            raise FileNotFoundError("MiniLM vocab file %s" % vocab)
        for vf in (vocab, os.environ.get("SVG_MINILM_VOCAB"), os.path.join(d, "vocab.txt") if d else None):
            if vf and os.path.exists(vf):
                self.tokenizer = _HFWordPiece(vf)
                break
        if self.tokenizer is None:
            self.tokenizer = StandInWordPiece(self.cfg["vocab"])

    def _load_local(self, d):
        import json
        cj = os.path.join(d, "config.json")
        if os.path.exists(cj):
            with open(cj) as f:
                c = json.load(f)
            m = {"vocab_size": "vocab", "hidden_size": "d_model", "num_attention_heads": "heads", "num_hidden_layers": "layers",
                 "intermediate_size": "ffn", "max_position_embeddings": "max_pos"}
            self.cfg.update({v: int(c[k]) for k, v in m.items() if k in c})
        for fn in ("model.safetensors", "pytorch_model.bin"):
            p = os.path.join(d, fn)
            if os.path.exists(p):
                if fn.endswith(".safetensors"):
                    from safetensors.torch import load_file
                    return load_file(p)
                return torch.load(p, map_location="cpu", weights_only=True)
        raise FileNotFoundError("no model.safetensors / pytorch_model.bin under $SVG_MINILM_WEIGHTS=%s" % d)

    def _fill(self, sd, synthetic=False):
        sd = {(k[len("bert."):] if k.startswith("bert.") else k): v for k, v in sd.items()}
        sd = {(k[len(PREFIX):] if k.startswith(PREFIX) else k): v for k, v in sd.items()}
        shapes = bert_shapes(self.cfg, pooler="pooler.dense.weight" in sd)
        missing = [k for k in shapes if k not in sd]
        if missing:
            raise KeyError("MiniLM weights: missing %s (%d more)" % (missing[0], len(missing) - 1))
        for k in list(self._modules):
            del self._modules[k]
        for k, shp in shapes.items():
            t = sd[k].detach().to(torch.float32).cpu()
            if tuple(t.shape) != tuple(shp):
                raise ValueError("MiniLM weight %s has shape %s, expected %s" % (k, tuple(t.shape), tuple(shp)))
            _attach(self, PREFIX + k, t.clone())
        # transformers 4.21's BertEmbeddings keeps position_ids as a persistent buffer: part of the reference's checkpoints
        _attach(self, PREFIX + "embeddings.position_ids", torch.arange(self.cfg["max_pos"]).unsqueeze(0), buffer=True)
        self.loaded = True
        self.synthetic = bool(synthetic)
        self._uploaded = None

    def load_state_dict(self, state_dict, strict=True):
        if state_dict:
            self._fill(dict(state_dict))
        return nn.modules.module._IncompatibleKeys([], [])

    def _sync(self):
        ctx = self._ctx or _lib.default_context()
        if self._uploaded is ctx and ctx.owner(_lib.SVG_MINILM) is self:
            return ctx
        c = self.cfg
        ctx.configure(_lib.SVG_MINILM, vocab=c["vocab"], d_model=c["d_model"], heads=c["heads"], layers=c["layers"], ffn=c["ffn"], max_pos=c["max_pos"])
        ctx.load_state_dict(_lib.SVG_MINILM, {k[len(PREFIX):]: v for k, v in self.state_dict().items() if "position_ids" not in k})
        self.n_params = ctx.finalize(_lib.SVG_MINILM)
        ctx.claim(_lib.SVG_MINILM, self)
        self._uploaded = ctx
        return ctx

    def encode(self, sentences, return_hidden=False):
        """list[str] -> (n, 384) unit-norm f32 tensor on the device (the reference's .encode returns the same values as numpy)"""
        if not self.loaded:
            raise FileNotFoundError("no MiniLM weights: set $SVG_MINILM_WEIGHTS to a local all-MiniLM-L6-v2 directory (model.safetensors, "
                                    "config.json, vocab.txt), load a reference text checkpoint (its sent_transformer.* entries), pass "
                                    "text_encoder=, or opt in to seeded synthetic weights (SVG_ALLOW_SYNTHETIC_WEIGHTS=1)")
        if isinstance(self.tokenizer, StandInWordPiece) and not self.synthetic and not self._allow_standin:
            raise FileNotFoundError("MiniLM holds REAL weights (checkpoint / local directory / dict) but no WordPiece vocabulary: the "
                                    "stand-in tokenizer's hashed ids would feed garbage to trained embeddings.  Pass vocab=<vocab.txt>, set "
                                    "$SVG_MINILM_VOCAB, or put vocab.txt into $SVG_MINILM_WEIGHTS (explicit opt-out for tests: "
                                    "allow_standin_tokenizer=True / SVG_MINILM_STANDIN_TOKENIZER=1)")
        if isinstance(sentences, str):
            sentences = [sentences]
        ids, lens = self.tokenizer(list(sentences))
        ctx = self._sync()
        return ctx.minilm_encode(ids, lens, self.cfg["d_model"], return_hidden)

    def forward(self, sentences):
        return self.encode(sentences)
