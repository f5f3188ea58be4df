"""Host side of the training step — mirror of the reference's ``trainers/trainer.py`` (Trainer: :39-300, main: :303-480).

What runs where: ``encode_batch`` is the HIP VAE encoder of the sampling path (frozen, like the reference's SD models); the
latent Transformer's train-mode forward, the criterion, the backward pass and Adam run inside libsvg_hip.so
(``svg_transformer_loss`` / ``svg_transformer_adam_step``, include/svg_hip.h) on the library's copy of the weights.  The
objects the reference's loop passes around keep their roles:

  * ``criterion(...)``      trainer.py:91-109 — returns the loss description the library evaluates (a ``Criterion``), with the
                            same keyword arguments; the invalid use_mse + use_L1 combination prints and returns None like there
  * ``Adam(model, lr)``     stands for ``optim.Adam(model.parameters(), lr=lr)`` (trainer.py:365): ``zero_grad()`` / ``step()``
  * ``train_loop`` / ``validation_loop`` / ``fit``   trainer.py:111-190, :192-260, :262-273 (same arguments, same returns)
  * ``main()``              trainer.py:303-480 without wandb (absent here): hyper-parameters come from the YAML config
                            (first entry of each list, exactly the values a one-point wandb sweep would deliver), the log is
                            JSON lines on stdout; checkpoints ``./checkpoints/<config>_<index>_{train,test}.pt`` as at :469-480.

Deviations, logging only: the per-term losses logged are those of the F predicted positions (the reference logs the GDL of all
positions, trainer.py:176) and the contrastive term is reported directly instead of as ``loss - mse - gdl`` (:178).
"""
import json
import os
import time

import torch

from . import _lib
from .config import parse_config_args


class Criterion:
    """the loss of trainer.py:91-109 as the library evaluates it: w_mse*MSE + w_l1*L1 + w_gdl*GDL(alpha) + w_con*BiPatchNCE(tau)"""

    def __init__(self, use_mse, use_L1, use_gdl, lambda_gdl, alpha, use_contrastive, temperature, lambda_contrastive, feat):
        self.w_mse = float(bool(use_mse))
        self.w_l1 = float(bool(use_L1))
        self.w_gdl = float(bool(use_gdl)) * float(lambda_gdl)
        self.alpha = float(alpha)
        self.w_contrastive = float(bool(use_contrastive)) * float(lambda_contrastive)
        self.temperature = float(temperature)
        self.feat = int(feat)

    def cfg(self, frames_to_predict, dropout_p=0.0, seed=0):
        return _lib.TrainCfg(frames_to_predict=int(frames_to_predict), feat_h=self.feat, feat_w=self.feat, w_mse=self.w_mse, w_l1=self.w_l1,
                             w_gdl=self.w_gdl, gdl_alpha=self.alpha, w_contrastive=self.w_contrastive, temperature=self.temperature,
                             dropout_p=float(dropout_p), seed=int(seed) & (2 ** 64 - 1))


class Adam:
    """``optim.Adam(model.parameters(), lr=lr)`` for a model whose gradients live in the library (trainer.py:365)."""

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.model, self.lr, self.betas, self.eps = model, lr, betas, eps

    def zero_grad(self):
        pass            # svg_transformer_loss(backward=1) overwrites the gradients: zero_grad + backward in one

    def step(self):
        self.model.adam_step(self.lr, self.betas, self.eps)


class Trainer:
    def __init__(self, sd_utils=None):
        self.config, self.args = parse_config_args()
        os.makedirs("./checkpoints", exist_ok=True)
        # trainer.py:43: the run index counts the checkpoints that carry this config's name
        self.index = len([name for name in os.listdir("./checkpoints") if self.args.config in name])
        self.device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        if sd_utils is None:
            from .sd_utils import SDUtils
            sd_utils = SDUtils()
        self.sd_utils = sd_utils
        self.SOS_token = torch.ones((1, 1, self.config.FRAME_SIZE ** 2 // 64 * 4), dtype=torch.float32, device=self.device) * 2
        # dropout: one fresh seed per training iteration.  The masks are a pure function of (seed, site, element), so the starting
        # point carries the entropy: torch's initial seed (torch.manual_seed makes a run reproducible, as in the reference), the
        # rank and the run index — a --resume run, the next sweep point and every replica rank get their own mask sequence
        self.seed = (int(torch.initial_seed()) * 1000003 + int(os.environ.get("RANK", "0")) * 7919 + self.index * 104729) & (2 ** 62 - 1)
        self._stream = torch.cuda.Stream() if torch.cuda.is_available() else None   # capturable: the library replays a step as one hipGraph
        self.log = lambda rec: print(json.dumps(rec), flush=True)

    def criterion(self, use_mse=True, use_L1=False, use_gdl=True, lambda_gdl=1, alpha=2, use_contrastive=True, temperature=0.07,
                  lambda_contrastive=0.1):
        if use_mse and use_L1:
            print("Invalid loss function combination")      # trainer.py:107-109
            return None
        return Criterion(use_mse, use_L1, use_gdl, lambda_gdl, alpha, use_contrastive, temperature, lambda_contrastive,
                         self.config.FRAME_SIZE // 8)

    def _loop(self, model, loss_fn, dataloader, frames_to_predict, opt):
        sums = {"total": 0.0, "mse": 0.0, "l1": 0.0, "gdl": 0.0, "contrastive": 0.0}
        n = 0
        for index_list, batch in dataloader:
            new_batch = self.sd_utils.encode_batch(batch, use_sos=True)                  # trainer.py:123
            new_batch = torch.as_tensor(new_batch).to(self.device)
            train = opt is not None
            self.seed += 1
            cfg = loss_fn.cfg(frames_to_predict, model.positional_encoder.dropout_p if train else 0.0, self.seed)
            if train:
                opt.zero_grad()
            torch.cuda.current_stream().synchronize()                                      # the encoded batch is ready
            with torch.cuda.stream(self._stream):
                terms = model.training_loss(cfg, new_batch, backward=train)                # :141-145 (+ loss.backward(), :164)
                if train:
                    opt.step()                                                             # :165
            self._stream.synchronize()
            for k in sums:
                sums[k] += terms[k]
            n += 1
        return {k: v / max(n, 1) for k, v in sums.items()}

    def train_loop(self, model, opt, scheduler, loss_fn, dataloader, frames_to_predict):
        model.train()
        avg = self._loop(model, loss_fn, dataloader, frames_to_predict, opt)
        self.log({"train_loss": avg["total"], "mse_train": avg["mse"], "L1_train": avg["l1"], "gdl_train": avg["gdl"],
                  "contrastive_train": avg["contrastive"]})
        return avg["total"]

    def validation_loop(self, model, loss_fn, dataloader, frames_to_predict):
        model.eval()
        avg = self._loop(model, loss_fn, dataloader, frames_to_predict, None)
        self.log({"val_loss": avg["total"], "mse_val": avg["mse"], "L1_val": avg["l1"], "gdl_val": avg["gdl"],
                  "contrastive_val": avg["contrastive"]})
        return avg["total"]

    def fit(self, model, opt, scheduler, loss_fn, train_dataloader, val_dataloader, frames_to_predict):
        print("Training and validating model")
        train_loss = self.train_loop(model, opt, scheduler, loss_fn, train_dataloader, frames_to_predict)
        validation_loss = self.validation_loop(model, loss_fn, val_dataloader, frames_to_predict)
        print(f"Training loss: {train_loss:.4f}")
        print(f"Validation loss: {validation_loss:.4f}")
        return train_loss, validation_loss


def _first(v):
    return v[0] if isinstance(v, (list, tuple)) else v


def make_loaders(args, config, frames_per_clip, frames_to_predict, stride, batch_size, epoch_ratio, num_workers):
    """trainer.py:372-447.  UCF-101 needs torchvision / PyAV, which this image does not have."""
    from torch.utils.data import DataLoader, RandomSampler
    from .loaders import BouncingBall, Kitti
    if args.dataset == "ball":
        mk = lambda stage: BouncingBall(num_frames=5, stride=stride, dir=args.folder, stage=stage, shuffle=True)
    elif args.dataset == "kitti":
        mk = lambda stage: Kitti(num_frames=(frames_per_clip + frames_to_predict), stride=1, dir=args.folder, stage=stage, shuffle=True)
    elif "ucf" in args.dataset:
        raise RuntimeError("the UCF-101 loader of the reference is torchvision.datasets.UCF101 (PyAV); neither is installed here")
    else:
        raise ValueError("Invalid dataset name")
    out = []
    for stage in ("train", "test"):
        ds = mk(stage)
        sampler = RandomSampler(ds, replacement=False, num_samples=max(1, int(len(ds) * epoch_ratio)))
        out.append(DataLoader(ds, batch_size=batch_size, shuffle=False, sampler=sampler, num_workers=num_workers, pin_memory=True))
    return out


def main():
    config, args = parse_config_args()
    frames_per_clip, frames_to_predict = _first(config.FRAMES_PER_CLIP), _first(config.FRAMES_TO_PREDICT)
    stride, batch_size, epoch_ratio = _first(config.STRIDE), _first(config.BATCH_SIZE), _first(config.EPOCH_RATIO)
    epochs, lr, num_workers = _first(config.EPOCHS), _first(config.LR), _first(config.NUM_WORKERS)
    from .transformer import Transformer
    trainer = Trainer()
    model = Transformer(num_tokens=0, dim_model=_first(config.DIM_MODEL), num_heads=_first(config.NUM_HEADS),
                        num_encoder_layers=_first(config.NUM_ENCODER_LAYERS), num_decoder_layers=_first(config.NUM_DECODER_LAYERS),
                        dropout_p=_first(config.DROPOUT_P))
    print("number of parameters: ", sum(p.numel() for p in model.parameters() if p.requires_grad))
    if args.resume:
        model.load_state_dict(torch.load("./checkpoints/" + args.old_name + ".pt", weights_only=True))
    opt = Adam(model, lr=lr)
    loss_fn = trainer.criterion(use_mse=_first(config.USE_MSE), use_L1=_first(getattr(config, "USE_L1", False)), use_gdl=_first(config.USE_GDL),
                                lambda_gdl=_first(config.LAMBDA_GDL), alpha=_first(config.ALPHA),
                                use_contrastive=_first(getattr(config, "USE_CONTRASTIVE", False)),
                                lambda_contrastive=_first(getattr(config, "LAMBDA_CONTRASTIVE", 0.0)))
    if loss_fn is None:
        raise ValueError("Invalid loss function combination")
    train_loader, test_loader = make_loaders(args, config, frames_per_clip, frames_to_predict, stride, batch_size, epoch_ratio, num_workers)
    stem = "./checkpoints/" + args.config + "_" + str(trainer.index)
    best_train_loss = best_val_loss = 1e10
    for epoch in range(1, epochs + 1):
        print("-" * 25, f"Epoch {epoch}", "-" * 25)
        t0 = time.time()
        train_loss, validation_loss = trainer.fit(model=model, opt=opt, scheduler=None, loss_fn=loss_fn, train_dataloader=train_loader,
                                                  val_dataloader=test_loader, frames_to_predict=frames_to_predict)
        trainer.log({"epoch": epoch, "seconds": time.time() - t0})
        if args.save_best:                                   # trainer.py:469-477
            if train_loss < best_train_loss:
                best_train_loss = train_loss
                torch.save(model.state_dict(), stem + "_train.pt")
                print("model saved as " + args.config + "_" + str(trainer.index) + "_train.pt (best train loss)")
            if validation_loss < best_val_loss:
                best_val_loss = validation_loss
                torch.save(model.state_dict(), stem + "_test.pt")
                print("model saved as " + args.config + "_" + str(trainer.index) + "_test.pt (best test loss)")
        else:                                                # :478-480
            torch.save(model.state_dict(), stem + "_test.pt")
            print("model saved as " + args.config + "_" + str(trainer.index) + "_test.pt")


if __name__ == "__main__":
    main()
