"""Generates tests/golden/*.pt by IMPORTING the live reference (read-only at /root/reference)
in the build container.  Fixtures are data only (inputs + the reference's outputs); nothing of
the reference's source travels.  Run:  python oracle/gen_golden.py

G1 transformer_tiny.pt   full state_dict of a tiny reference Transformer + X/outputs (T=6 and T=5)
G2 transformer_spot.pt   full-size spot checks: seed, sha256(state_dict), X, pred — the weights are
                         regenerated from the seed with torch's own nn modules in the reference's
                         construction order (transformer.py:33-45)
G3 pe_quirk.pt           PositionalEncoding(256,0.1,64).eval()(zeros(3,6,256))
G4 tgt_masks.pt          get_tgt_mask(5), (6)
G5 loop_trace.pt         predict.py:136-196 loop shapes driven through the reference Transformer
"""
import hashlib
import os
import sys

import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def sd_hash(sd):
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(sd[k].detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()


def build_ref(config_name, **kw):
    sys.argv = ["gen_golden", "--dataset", "ball", "--config", config_name]
    from models.transformer import Transformer
    return Transformer(**kw)


def main():
    os.makedirs(OUT, exist_ok=True)
    os.chdir(REF)
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    torch.set_num_threads(8)
    from models.positional_encoding import PositionalEncoding
    from prediction_shim import predict_ref  # noqa: F401  (defined below, registered in sys.modules)

    # ---- G1
    torch.manual_seed(1234)
    m = build_ref("model_10_26", dim_model=32, num_heads=4, num_encoder_layers=1, num_decoder_layers=2).eval()
    X6 = torch.randn(1, 6, 256)
    X5 = torch.randn(1, 5, 256)
    Xb = torch.randn(3, 5, 256)          # batch 3: exercises PE-by-batch-index
    with torch.no_grad():
        o6 = m(X6, X6, m.get_tgt_mask(6))
        o5 = m(X5, X5, m.get_tgt_mask(5))
        ob = m(Xb, Xb, m.get_tgt_mask(5))
        onomask = m(X5, X6, None)         # src != tgt lengths, no mask
        p6 = predict_ref(m, X6)
    torch.save({"state_dict": {k: v.clone() for k, v in m.state_dict().items()}, "num_heads": 4,
                "X6": X6, "X5": X5, "Xb": Xb, "out6": o6, "out5": o5, "outb": ob, "out_nomask": onomask,
                "pred6": p6}, os.path.join(OUT, "transformer_tiny.pt"))

    # ---- G5 (same tiny model) — loop plumbing of predict.py:136-196 with random latents for the VAE
    torch.manual_seed(77)
    nb = torch.cat([2.0 * torch.ones(1, 1, 256), torch.randn(1, 5, 256)], dim=1)
    X = nb
    inputs = nb[:, 1:]
    preds = torch.zeros(1, 0, 256)
    trace = []
    for _ in range(4):
        sin = tuple(X.shape)
        pred = predict_ref(m, X)
        preds = torch.cat((preds, pred.unsqueeze(0).unsqueeze(0)), dim=1)
        all_latents = torch.cat([inputs[:, :-1], preds], dim=1)
        X = all_latents[:, -5:]
        trace.append((sin, tuple(all_latents.shape)))
    torch.save({"new_batch": nb, "all_latents": all_latents, "trace": trace},
               os.path.join(OUT, "loop_trace.pt"))

    # ---- G2 full-size spot checks
    spots = {}
    for cfg, kw, dlat, seed in [
        ("config_test", dict(dim_model=256, num_heads=8, num_encoder_layers=6, num_decoder_layers=6, dropout_p=0.1), 1024, 11),
        ("1_16_kitti_L1_64", dict(dim_model=2048, num_heads=8, num_encoder_layers=4, num_decoder_layers=8, dropout_p=0.1), 256, 12),
        # BASELINE configs[3]: UCF-101, F=128 -> D_lat 1024, d=2048
        ("11_27_ucf_final", dict(dim_model=2048, num_heads=8, num_encoder_layers=4, num_decoder_layers=8, dropout_p=0.1), 1024, 13),
    ]:
        for mod in [k for k in sys.modules if k.startswith("models") or k.startswith("utils")]:
            del sys.modules[mod]
        torch.manual_seed(seed)
        m2 = build_ref(cfg, **kw).eval()
        n_params = sum(p.numel() for p in m2.parameters())
        g = torch.Generator().manual_seed(seed + 1000)
        X = torch.randn(1, 6, dlat, generator=g)
        with torch.no_grad():
            pred = predict_ref(m2, X)
        spots[cfg] = {"seed": seed, "kw": kw, "d_lat": dlat, "n_params": n_params,
                      "sha256": sd_hash(m2.state_dict()), "X": X, "pred": pred}
        print(cfg, n_params, spots[cfg]["sha256"][:16], float(pred.abs().mean()))
        del m2
    torch.save(spots, os.path.join(OUT, "transformer_spot.pt"))

    # ---- G3 / G4
    pe = PositionalEncoding(256, 0.1, 64).eval()
    torch.save({"out": pe(torch.zeros(3, 6, 256))}, os.path.join(OUT, "pe_quirk.pt"))
    m = build_ref("model_10_26", dim_model=32, num_heads=4, num_encoder_layers=1, num_decoder_layers=1)
    torch.save({"m5": m.get_tgt_mask(5), "m6": m.get_tgt_mask(6)}, os.path.join(OUT, "tgt_masks.pt"))

    # ---- config surface: YAML values of the configs named in BASELINE.json, as the reference parses them
    import yaml
    cfgs = {}
    for name in ["config_test", "1_16_kitti_L1_64", "1_19_ball_complex_L1_64", "11_27_ucf_final",
                 "11_27_ucf_text_final", "model_10_26"]:
        with open(os.path.join(REF, "config", name + ".yml")) as f:
            cfgs[name] = yaml.safe_load(f)
    torch.save(cfgs, os.path.join(OUT, "config_values.pt"))
    print("golden fixtures written to", os.path.abspath(OUT))


# prediction/predict.py cannot be imported here (cv2 / torchvision missing: ordinary ModuleNotFoundError),
# so its `predict()` (predict.py:16-42) is driven through the imported reference Transformer like this:
import types  # noqa: E402

_shim = types.ModuleType("prediction_shim")


def _predict_ref(model, input_sequence):
    model.eval()
    with torch.no_grad():
        tgt_mask = model.get_tgt_mask(input_sequence.size(1))
        pred = model(input_sequence, input_sequence, tgt_mask).permute(1, 0, 2)
    return pred[0, -1]


_shim.predict_ref = _predict_ref
sys.modules["prediction_shim"] = _shim

if __name__ == "__main__":
    main()
