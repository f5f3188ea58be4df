"""`python -m prediction.predict_text` — the reference's text-conditioned sampling + FVD script (prediction/predict_text.py:76-321):
load the text Transformer, walk a test set clip by clip (UCF-101: class name per clip), generate ``--pred_frames`` frames per clip
with `predict(model, X, cls_list)` (optionally through the denoise round trip), collect I3D logits of the real clips and of the
generated ones 16 clips at a time, and print the Fréchet distance between the two sets.

Same flags, same flow and quirks:
  * UCF test clips have 16 frames (predict_text.py:133), train clips 5 (:127); ``frame_rate=3``; up to 2048 random clips (:137);
  * ``real`` / ``fake`` stacks go through ``(x * 255).astype('uint8')`` on tensors that already ARE uint8 (:164,:283): the
    multiplication wraps modulo 256 — reproduced (both sets get the same map, so the distance is between "negated" videos);
  * the generated frames enter the FVD stack only under ``--save_output`` (:260-288);
  * a group of 16 clips is embedded as soon as it is complete; an incomplete tail is dropped (:163-167,:281-287).
Differences: clips come from pre-extracted frames (loaders.UCF101Frames: no PyAV offline), the per-clip loop is the clip-batched
`sample_clips` (one clip per call here: results are batch-invariant), ranks shard the clips and `fvd.all_gather` reassembles the
logits (fvd_2.py:103-107), images are written by the PNG writer of predict.py instead of cv2.
"""
import os

import numpy as np
import torch


def splitClassNames(classes):
    """predict_text.py:18-32"""
    from .loaders import split_class_names
    return split_class_names(classes)


def find_classes(directory):
    """predict_text.py:34-46"""
    from .loaders import find_classes as fc
    return fc(directory)


def fvd_from_stacks(real_groups, fake_groups, i3d):
    """real_groups / fake_groups: lists of (16,T,H,W,3) uint8 tensors -> Fréchet distance of their I3D logits (predict_text.py:308-315).
    The decision "are there enough clips" is COLLECTIVE: every rank embeds what it holds (possibly nothing: a (0, 400) block),
    every rank enters both all_gathers, and only then the gathered row counts are tested — a rank whose shard held fewer than 16
    clips must not leave before the collective the other ranks are waiting in.  Returns (None, real, fake) when either set is empty."""
    from . import fvd
    dev = i3d.ctx.device

    def embed(groups):
        if groups:
            return torch.cat([fvd.get_fvd_logits(g, i3d) for g in groups])
        return torch.zeros((0, i3d.num_classes), dtype=torch.float32, device=dev)
    real, fake = fvd.all_gather(embed(real_groups)), fvd.all_gather(embed(fake_groups))
    print("fake_embeddings shape", tuple(fake.shape))
    print("real_embeddings shape", tuple(real.shape))
    if real.shape[0] == 0 or fake.shape[0] == 0:
        return None, real, fake
    return fvd.frechet_distance(fake.clone(), real), real, fake


def main(argv=None):
    from . import config as svg_config, fvd, sharding
    from .loaders import BouncingBall, UCF101Frames, ucf_dirs, ucf_transform
    from .predict import sample_clips, save_frames
    from .sd_utils import SDUtils, synthetic_allowed
    from .transformer_text import Transformer
    if argv is not None:
        svg_config.set_args(argv)
    config, args = svg_config.parse_config_args()
    if "RANK" in os.environ and int(os.environ.get("WORLD_SIZE", "1")) > 1:
        import torch.distributed as dist
        torch.cuda.set_device(int(os.environ.get("SVG_DEVICE_OVERRIDE", os.environ.get("LOCAL_RANK", "0"))))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(os.environ.get("SVG_DIST_BACKEND", "nccl"))
    rank, ws = sharding.world()
    sd_utils = SDUtils(verbose=(rank == 0))
    device = sd_utils.device
    model = Transformer(num_tokens=0, dim_model=config.DIM_MODEL[0], num_heads=config.NUM_HEADS[0], num_encoder_layers=config.NUM_ENCODER_LAYERS[0],
                        num_decoder_layers=config.NUM_DECODER_LAYERS[0], dropout_p=config.DROPOUT_P[0])
    ckpt = "./checkpoints/" + str(args.config) + "_" + str(args.index) + "_" + str(args.mode) + ".pt"          # predict_text.py:83
    if os.path.exists(ckpt):
        model.load_state_dict(torch.load(ckpt, map_location="cpu", weights_only=True))
    elif not synthetic_allowed():
        raise FileNotFoundError(ckpt)
    model.eval()
    i3d = fvd.load_i3d_pretrained(device, ctx=sd_utils.ctx)                                                    # :86
    F = config.FRAME_SIZE
    idx_to_class = None
    if args.dataset == "ball":                                                                                 # :90-92
        ds = BouncingBall(num_frames=5, stride=1, dir=args.folder, stage="test", shuffle=False)
        keep = [i for i in range(len(ds)) if len(ds.dataset[i]) == 5]
        n_items = len(keep)

        def load_item(j):
            i = keep[j]
            return torch.tensor(ds.indices[i]), torch.from_numpy(ds[i][1])
    elif "ucf" in args.dataset:                                                                                # :94-138
        ucf_data_dir, ucf_label_dir = ucf_dirs(args.dataset)
        if args.folder:                       # pre-extracted frames live elsewhere: <folder>/<same relative layout>
            ucf_data_dir, ucf_label_dir = os.path.join(args.folder, ucf_data_dir), os.path.join(args.folder, ucf_label_dir)
        _, idx_to_class = find_classes(ucf_data_dir)
        print("Loading UCF dataset from", ucf_data_dir)
        train = args.mode == "train"
        ucf = UCF101Frames(ucf_data_dir, ucf_label_dir, frames_per_clip=5 if train else 16, train=train, transform=ucf_transform(F), frame_rate=3)
        # RandomSampler(num_samples=2048) (:137) draws WITHOUT replacement permutation after permutation until it has 2048
        # indices: a set smaller than 2048 is visited several times.  Same here, seeded; $SVG_FVD_CLIPS shortens the walk.
        n = int(os.environ.get("SVG_FVD_CLIPS", "2048"))
        g = torch.Generator().manual_seed(0)
        order = []
        while len(order) < n and len(ucf) > 0:
            order += torch.randperm(len(ucf), generator=g).tolist()
        order = order[:n]
        n_items = len(order)

        def load_item(j):                      # one decode + transform per clip, and only for the clips of this rank's shard
            v, _, l = ucf[order[j]]
            return torch.tensor([l]), torch.from_numpy(v)
    else:
        raise ValueError("Invalid dataset name")
    a, b = sharding.shard_range(n_items, rank, ws)
    real_embeddings, fake_embeddings = [], []
    real_input, fake_input = None, None
    out_tag = str(args.config) + "_" + str(args.index) + "_" + str(args.mode)
    with torch.no_grad():
        for ind in range(a, b):
            index_list, clip = load_item(ind)
            batch = clip.unsqueeze(0)                                              # (1,T,F,F,3) uint8 BGR, like the DataLoader's batch
            cls_list = [idx_to_class[int(i)] for i in index_list.tolist()] if idx_to_class is not None else None
            real_input = batch if real_input is None else torch.cat((real_input, batch), 0)                   # :157-161
            if real_input.shape[0] >= 16:
                real_embeddings.append((real_input * 255).to(torch.uint8))                                     # :164 (wraps: the input is uint8)
                real_input = None
            kw = dict(cls_list=cls_list) if cls_list is not None else {}
            lat, frames = sample_clips(model, sd_utils, batch.to(device), args.pred_frames, denoise=bool(args.denoise),
                                       start_step=args.denoise_start_step, seeds=[ind], return_frames=True, **kw)  # :186-258
            if args.save_output:                                                                               # :260-288
                n_in = batch.shape[1] - 1
                is_pred = [False] * n_in + [True] * (frames.shape[1] - n_in)
                fr = frames[0].cpu()
                save_frames(fr.numpy(), [False] * len(is_pred), os.path.join("outputs_pred", out_tag, str(ind)))
                fake_curr = fr[n_in:].unsqueeze(0)                                 # the predicted frames only (:267-274)
                fake_input = fake_curr if fake_input is None else torch.cat((fake_input, fake_curr), 0)
                if fake_input.shape[0] >= 16:
                    fake_embeddings.append((fake_input * 255).to(torch.uint8))                                 # :283
                    fake_input = None
        # every rank takes part in the gathers, whatever its shard produced (a rank that returned early here left the others
        # hanging in all_gather: e.g. 31 clips on 2 ranks = one complete group on rank 0, none on rank 1)
        fvd_value, real_all, fake_all = fvd_from_stacks(real_embeddings, fake_embeddings, i3d)
        if fvd_value is None:
            if rank == 0:
                print("FVD needs at least 16 real clips and, under --save_output, 16 generated ones on some rank (got %d / %d rows)"
                      % (real_all.shape[0], fake_all.shape[0]))
            return None
    if rank == 0:
        print("FVD: ", fvd_value)                                                                              # :315
    return fvd_value
