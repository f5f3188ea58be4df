"""Generates tests/golden/transformer_pad.pt by IMPORTING the live reference Transformer (read-only at /root/reference):
key-padding masks through models/transformer.py:64 (src_key_padding_mask / tgt_key_padding_mask of nn.Transformer).
The tiny model is the one of transformer_tiny.pt (its state_dict is loaded into the reference module), so the fixture
holds inputs, masks and the reference's outputs only.  Run:  python oracle/gen_golden_pad.py
"""
import os
import sys

import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def main():
    tiny = torch.load(os.path.join(OUT, "transformer_tiny.pt"), weights_only=False)
    os.chdir(REF)
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    sys.argv = ["gen_golden_pad", "--dataset", "ball", "--config", "model_10_26"]
    from models.transformer import Transformer
    m = Transformer(dim_model=32, num_heads=4, num_encoder_layers=1, num_decoder_layers=2).eval()
    m.load_state_dict(tiny["state_dict"])
    g = torch.Generator().manual_seed(4321)
    src = torch.randn(3, 6, 256, generator=g)
    tgt = torch.randn(3, 5, 256, generator=g)
    # token "ids" as create_pad_mask sees them (transformer.py:91-94): 0 = pad
    src_ids = torch.tensor([[5, 3, 8, 1, 0, 0], [7, 7, 2, 0, 0, 0], [1, 2, 3, 4, 5, 6]])
    tgt_ids = torch.tensor([[5, 3, 8, 0, 0], [7, 7, 2, 9, 0], [1, 2, 3, 4, 5]])
    sp = m.create_pad_mask(src_ids, 0)
    tp = m.create_pad_mask(tgt_ids, 0)
    with torch.no_grad():
        both = m(src, tgt, m.get_tgt_mask(5), sp, tp)
        only_src = m(src, tgt, m.get_tgt_mask(5), sp, None)
        only_tgt = m(src, tgt, None, None, tp)
        # float key-padding masks are added to the scores (nn.MultiheadAttention): a finite bias
        fsp = torch.zeros(3, 6).masked_fill(sp, -2.5)
        float_src = m(src, tgt, m.get_tgt_mask(5), fsp, None)
    torch.save({"src": src, "tgt": tgt, "src_ids": src_ids, "tgt_ids": tgt_ids, "src_pad": sp, "tgt_pad": tp, "float_src_pad": fsp,
                "out_both": both, "out_src": only_src, "out_tgt": only_tgt, "out_float_src": float_src},
               os.path.join(OUT, "transformer_pad.pt"))
    print("transformer_pad.pt written:", [float(x.abs().mean()) for x in (both, only_src, only_tgt, float_src)])


if __name__ == "__main__":
    main()
