"""Generates tests/golden/train_tiny.pt by IMPORTING the live reference (read-only at /root/reference) in the build
container: one training iteration of the reference Transformer (train mode, dropout_p = 0 so that it is deterministic) on the
weights of tests/golden/transformer_tiny.pt, with the reference's own BiPatchNCE, and two torch.optim.Adam steps.
Fixtures are data only.  Run:  python oracle/gen_golden_train.py

trainers/trainer.py cannot be imported here (cv2 / wandb / diffusers are missing), so the criterion around the imported
pieces follows oracle/train_oracle.py; what this fixture pins is (a) the train-mode forward and the autograd gradients of
the live module against the oracle's explicit ops, (b) the live BiPatchNCE against its restatement, (c) torch's Adam.
"""
import os
import sys

import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "tests", "golden")


def main():
    root = os.path.abspath(os.path.join(HERE, ".."))
    sys.path.insert(0, root)
    from oracle import train_oracle as TR
    tiny = torch.load(os.path.join(OUT, "transformer_tiny.pt"))
    # from here on `models.*` / `utils.*` must resolve to the REFERENCE, not to this repo's same-named shim packages
    sys.path[:] = [p for p in sys.path if os.path.abspath(p or ".") != root]
    for k in [k for k in sys.modules if k.split(".")[0] in ("models", "utils", "prediction", "trainers", "loaders")]:
        del sys.modules[k]
    os.chdir(REF)
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    sys.argv = ["gen_golden_train", "--dataset", "ball", "--config", "model_10_26"]
    from models.transformer import Transformer
    from models.contrastive_loss import BiPatchNCE
    import models.transformer as _mt
    assert _mt.__file__.startswith(REF), _mt.__file__
    torch.manual_seed(99)
    m = Transformer(dim_model=32, num_heads=4, num_encoder_layers=1, num_decoder_layers=2, dropout_p=0.0)
    m.load_state_dict(tiny["state_dict"])
    m.train()
    B, T, F, feat = 3, 7, 3, 8
    new_batch = torch.cat([2.0 * torch.ones(B, 1, 256), torch.randn(B, T - 1, 256)], dim=1)
    y_input = new_batch[:, :-1]
    y_expected = new_batch[:, 1:].permute(1, 0, 2)
    w = dict(w_mse=1.0, w_l1=0.5, w_gdl=0.7, alpha=2, w_contrastive=0.1, temperature=0.07)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    out = {"new_batch": new_batch, "frames_to_predict": F, "feat": feat, "weights": w, "lr": 1e-3, "steps": []}
    for step in range(2):
        pred = m(new_batch, y_input, m.get_tgt_mask(y_input.size(1)))
        x, y = pred[-F:], y_expected[-F:]
        nce = BiPatchNCE(N=B, T=F, h=feat, w=feat, temperature=w["temperature"])(
            x.permute(1, 0, 2).reshape(-1, F, 4, feat, feat), y.permute(1, 0, 2).reshape(-1, F, 4, feat, feat))
        terms = {"mse": torch.nn.MSELoss()(x, y), "l1": torch.nn.L1Loss()(x, y),
                 "gdl": TR.gradient_difference_loss(x, y, w["alpha"]), "contrastive": nce}
        total = w["w_mse"] * terms["mse"] + w["w_l1"] * terms["l1"] + w["w_gdl"] * terms["gdl"] + w["w_contrastive"] * terms["contrastive"]
        opt.zero_grad()
        total.backward()
        grads = {k: p.grad.clone() for k, p in m.named_parameters()}
        opt.step()
        out["steps"].append({"total": total.detach(), "terms": {k: v.detach() for k, v in terms.items()}, "grads": grads,
                             "params_after": {k: p.detach().clone() for k, p in m.named_parameters()}})
    # keep the fixture small: the norm of every gradient, the full tensor for one of each kind
    keep = ("embedding.weight", "embedding.bias", "out.weight", "out.bias",
            "transformer.encoder.layers.0.self_attn.in_proj_weight", "transformer.encoder.layers.0.self_attn.out_proj.bias",
            "transformer.encoder.layers.0.linear1.weight", "transformer.encoder.layers.0.norm1.weight", "transformer.encoder.norm.bias",
            "transformer.decoder.layers.0.self_attn.in_proj_bias", "transformer.decoder.layers.1.multihead_attn.in_proj_weight",
            "transformer.decoder.layers.1.linear2.weight", "transformer.decoder.layers.1.norm3.bias", "transformer.decoder.norm.weight")
    for st in out["steps"]:
        st["grad_norms"] = {k: v.norm() for k, v in st["grads"].items()}
        st["grads"] = {k: st["grads"][k] for k in keep}
        st["param_norms_after"] = {k: v.norm() for k, v in st["params_after"].items()}
        st["params_after"] = {k: st["params_after"][k] for k in keep}
    torch.save(out, os.path.join(OUT, "train_tiny.pt"))
    print("wrote train_tiny.pt: total loss", [float(s["total"]) for s in out["steps"]], {k: float(v) for k, v in out["steps"][0]["terms"].items()})


if __name__ == "__main__":
    main()
