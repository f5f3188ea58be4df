"""ORACLE — TEST INFRASTRUCTURE ONLY.  Not shipped, not measured, never imported by the product path (only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this).

CPU fp32 restatement of one optimisation step of the reference trainer, for the latent Transformer:

  * trainers/trainer.py:120-164   train_loop body: y_input = new_batch[:, :-1]; y_expected = new_batch[:, 1:].permute(1,0,2);
                                  pred = model(new_batch, y_input, tgt_mask); loss = loss_fn(pred[-F:], y_expected[-F:]);
                                  opt.zero_grad(); loss.backward(); opt.step()
  * trainers/trainer.py:65-109    gradient_difference_loss and criterion (use_* flags x lambda_*)
  * models/contrastive_loss.py:7-60   BiPatchNCE (bidirectional patch-wise InfoNCE with detached negatives in direction 1)
  * trainers/trainer.py:365       optim.Adam(model.parameters(), lr=lr)

The forward is oracle/transformer_oracle.py (pinned against the live reference module); gradients come from torch autograd
over those explicit ops.  Parity pin of this file: tests/golden/train_tiny.pt — losses, gradients and two Adam steps of the
LIVE reference Transformer (train mode, dropout_p = 0) with the live reference BiPatchNCE, written by oracle/gen_golden_train.py;
tests/test_oracle_train.py checks this restatement against it.  (trainers/trainer.py itself cannot be imported here — cv2,
wandb, diffusers are missing — so gradient_difference_loss / criterion are restated from the text.)
"""
import torch
import torch.nn.functional as F

from . import transformer_oracle as TO


def gradient_difference_loss(x, y, alpha=1):
    """trainers/trainer.py:65-86.  x, y (T, B, D_lat) -> scalar."""
    v = int((x.shape[-1] // 4) ** 0.5)
    fx = x.reshape(x.shape[0], x.shape[1], 4, v, v)
    fy = y.reshape(y.shape[0], x.shape[1], 4, v, v)
    vx = fx[:, :, :, 1:, :] - fx[:, :, :, :-1, :]
    vy = fy[:, :, :, 1:, :] - fy[:, :, :, :-1, :]
    hx = fx[:, :, :, :, 1:] - fx[:, :, :, :, :-1]
    hy = fy[:, :, :, :, 1:] - fy[:, :, :, :, :-1]
    lv = torch.abs(torch.abs(vx) - torch.abs(vy))
    lh = torch.abs(torch.abs(hx) - torch.abs(hy))
    return (torch.sum(torch.pow(lv, alpha)) + torch.sum(torch.pow(lh, alpha))) / x.numel()


def bi_patch_nce(pred_f, gt_f, temperature=0.07):
    """models/contrastive_loss.py:29-60.  pred_f, gt_f (N, T, C, h, w) -> scalar."""
    N, T, C, h, w = pred_f.shape
    mask = torch.eye(h * w).unsqueeze(0).repeat(N * T, 1, 1)
    g = gt_f.reshape(N * T, C, h * w).transpose(1, 2)
    p = pred_f.reshape(N * T, C, h * w).transpose(1, 2)
    s1 = (torch.matmul(g, p.transpose(1, 2)) * mask + torch.matmul(g, p.detach().transpose(1, 2)) * (1.0 - mask)) / temperature
    s2 = (torch.matmul(p, g.transpose(1, 2)) * mask + torch.matmul(p, g.detach().transpose(1, 2)) * (1.0 - mask)) / temperature
    target = torch.arange(h * w).repeat(N * T)
    return 0.5 * (F.cross_entropy(s1.flatten(0, 1), target) + F.cross_entropy(s2.flatten(0, 1), target))


def criterion(x, y, frames_to_predict, feat, w_mse=0.0, w_l1=0.0, w_gdl=0.0, alpha=1, w_contrastive=0.0, temperature=0.07):
    """trainers/trainer.py:91-109 on x = pred[-F:], y = y_expected[-F:] (F, B, D_lat).  -> (total, dict of the terms)."""
    terms = {"mse": F.mse_loss(x, y), "l1": F.l1_loss(x, y), "gdl": gradient_difference_loss(x, y, alpha)}
    if w_contrastive:
        px = x.permute(1, 0, 2).reshape(-1, frames_to_predict, 4, feat, feat)
        py = y.permute(1, 0, 2).reshape(-1, frames_to_predict, 4, feat, feat)
        terms["contrastive"] = bi_patch_nce(px, py, temperature)
    else:
        terms["contrastive"] = torch.zeros(())
    total = w_mse * terms["mse"] + w_l1 * terms["l1"] + w_gdl * terms["gdl"] + w_contrastive * terms["contrastive"]
    return total, terms


def loss(sd, num_heads, new_batch, frames_to_predict, feat, drop=None, txt=None, **weights):
    """The loss of one train_loop iteration for the encoded batch `new_batch` (B, T, D_lat) (trainer.py:124-145)."""
    y_input = new_batch[:, :-1]
    y_expected = new_batch[:, 1:].permute(1, 0, 2)
    mask = TO.get_tgt_mask(y_input.size(1))
    pred = TO.forward(sd, new_batch, y_input, num_heads, mask, txt, drop=drop)
    return criterion(pred[-frames_to_predict:], y_expected[-frames_to_predict:], frames_to_predict, feat, **weights)


def leaf_state(sd):
    """state_dict -> the same tensors as autograd leaves (buffers untouched)."""
    return {k: (v.clone().requires_grad_(True) if k != "positional_encoder.pos_encoding" else v.clone()) for k, v in sd.items()}


def params_of(sd):
    return [v for k, v in sorted(sd.items()) if v.requires_grad]
