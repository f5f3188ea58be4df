"""Generates tests/golden/i3d_fvd.pt by IMPORTING the live reference modules evaluation/pytorch_i3d.py and evaluation/fvd_2.py
(read-only at /root/reference; they need torch + numpy only): the reference InceptionI3d with seeded weights
(oracle/i3d_oracle.seeded_i3d_weights — the Kinetics checkpoint is not in the tree) on a seeded 16-frame clip, the feature map after
Mixed_3c of the same run, the reference's preprocess on a seeded uint8 video, and its frechet_distance on seeded embeddings.
The fixture holds seeds, small inputs and the reference's outputs only.   python oracle/gen_golden_i3d.py"""
import os
import sys

import torch

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import i3d_oracle as IO  # noqa: E402


def main():
    sys.dont_write_bytecode = True
    sys.path.insert(0, os.path.join(REF, "evaluation"))
    from pytorch_i3d import InceptionI3d
    import fvd_2
    torch.set_num_threads(8)
    m = InceptionI3d(400, in_channels=3).eval()
    sd = IO.seeded_i3d_weights(17)
    full = dict(m.state_dict())
    full.update(sd)                                   # num_batches_tracked stays as constructed
    m.load_state_dict(full)
    g = torch.Generator().manual_seed(5)
    vid = torch.randint(0, 256, (2, 16, 48, 64, 3), dtype=torch.uint8, generator=g)          # (b,t,h,w,c)
    x = fvd_2.preprocess(vid.numpy())
    feats = {}
    m._modules["Mixed_3c"].register_forward_hook(lambda mod, i, o: feats.__setitem__("Mixed_3c", o))
    with torch.no_grad():
        logits = m(x)
    e1, e2, e3 = IO.fvd_test_embeddings(5)
    rec = {"w_seed": 17, "video": vid, "pre_stats": (float(x.mean()), float(x.std()), float(x.min()), float(x.max())),
           "pre_slice": x[:, :, ::5, ::37, ::41].clone(), "logits": logits.clone(),
           "mixed3c_slice": feats["Mixed_3c"][:, ::17, :, ::5, ::6].clone(), "emb_seed": 5,
           "fd_12": float(fvd_2.frechet_distance(e1.clone(), e2.clone())), "fd_11": float(fvd_2.frechet_distance(e1.clone(), e1.clone())),
           "fd_13": float(fvd_2.frechet_distance(e1.clone(), e3.clone())), "fd_33": float(fvd_2.frechet_distance(e3.clone(), e3[:300].clone()))}
    torch.save(rec, os.path.join(ROOT, "tests", "golden", "i3d_fvd.pt"))
    print("i3d_fvd.pt:", tuple(logits.shape), float(logits.abs().mean()), rec["fd_12"], rec["fd_11"], rec["fd_13"], rec["fd_33"])


if __name__ == "__main__":
    main()
