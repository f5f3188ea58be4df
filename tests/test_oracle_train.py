"""The training oracle (oracle/train_oracle.py + the drop-aware oracle/transformer_oracle.py) against the LIVE reference:
tests/golden/train_tiny.pt holds losses, gradients and two torch.optim.Adam steps of the reference Transformer in train mode
(dropout_p = 0) with the reference's BiPatchNCE (written by oracle/gen_golden_train.py)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import train_oracle as TR  # noqa: E402
from oracle import transformer_oracle as TO  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def rel(a, b):
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def test_two_training_steps_match_the_live_reference():
    fx = torch.load(os.path.join(GOLD, "train_tiny.pt"))
    sd = TR.leaf_state(torch.load(os.path.join(GOLD, "transformer_tiny.pt"))["state_dict"])
    names = [k for k, v in sorted(sd.items()) if v.requires_grad]
    opt = torch.optim.Adam([sd[k] for k in names], lr=fx["lr"])
    for st in fx["steps"]:
        total, terms = TR.loss(sd, 4, fx["new_batch"], fx["frames_to_predict"], fx["feat"], **fx["weights"])
        assert abs(float(total) - float(st["total"])) <= 2e-6 * abs(float(st["total"]))
        for k in ("mse", "l1", "gdl", "contrastive"):
            assert abs(float(terms[k]) - float(st["terms"][k])) <= 3e-6 * max(abs(float(st["terms"][k])), 1e-3), k
        opt.zero_grad()
        total.backward()
        for k in names:
            assert abs(float(sd[k].grad.norm()) - float(st["grad_norms"][k])) <= 2e-5 * float(st["grad_norms"][k]) + 1e-9, k
        for k, g in st["grads"].items():
            assert rel(sd[k].grad, g) < 2e-5, (k, rel(sd[k].grad, g))
        opt.step()
        # Adam's update is lr * m / (sqrt(v) + eps): where a gradient is rounding noise (e.g. the key bias of an attention, whose true
        # gradient is zero: softmax ignores a constant shift of a row) the ratio turns the noise into +-lr with a random sign.  Compare
        # the parameters where the gradient is meaningful.
        for k, p in st["params_after"].items():
            ok = st["grads"][k].abs() > 1e-5 * st["grads"][k].abs().max()
            assert rel(sd[k].detach()[ok], p[ok]) < 1e-5, k


def test_bipatchnce_gradient_structure():
    """contrastive_loss.py:41-49: direction 1 lets the gradient reach pred only through the diagonal scores (negatives are
    detached); direction 2 through every score of its row."""
    torch.manual_seed(0)
    p = torch.randn(2, 3, 4, 4, 4, requires_grad=True)
    g = torch.randn(2, 3, 4, 4, 4)
    TR.bi_patch_nce(p, g, 0.5).backward()
    # closed form used by the HIP kernel
    P = p.detach().reshape(6, 4, 16).transpose(1, 2)
    G = g.reshape(6, 4, 16).transpose(1, 2)
    S = torch.matmul(G, P.transpose(1, 2)) / 0.5                      # S[i][j] = <gt_i, pred_j> / tau
    sm_row = torch.softmax(S, dim=2)
    d1 = (torch.diagonal(sm_row, dim1=1, dim2=2) - 1).unsqueeze(-1) * G
    sm_col = torch.softmax(S.transpose(1, 2), dim=2)                  # rows of S^T
    d2 = torch.matmul(sm_col, G) - G
    want = (0.5 / 0.5 / (6 * 16)) * (d1 + d2)
    got = p.grad.reshape(6, 4, 16).transpose(1, 2)
    assert rel(got, want) < 1e-5


def test_drop_sites_are_visited_in_execution_order():
    sd = torch.load(os.path.join(GOLD, "transformer_tiny.pt"))["state_dict"]
    shapes = []

    def drop(x):
        shapes.append(tuple(x.shape))
        return x
    X = torch.randn(2, 6, 256)
    TO.forward(sd, X, X[:, :-1], 4, TO.get_tgt_mask(5), drop=drop)
    # embed(src), embed(tgt); 1 encoder layer: P, drop1, ff-inner, drop2; 2 decoder layers: P, drop1, P, drop2, ff-inner, drop3
    assert len(shapes) == 2 + 4 + 2 * 6
    assert shapes[0] == (6, 2, 32) and shapes[1] == (5, 2, 32)
    assert shapes[2] == (2 * 4, 6, 6) and shapes[4] == (6, 2, 2048)
    assert shapes[6] == (2 * 4, 5, 5) and shapes[8] == (2 * 4, 5, 6) and shapes[10] == (5, 2, 2048)
