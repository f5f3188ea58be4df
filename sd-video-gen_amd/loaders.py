"""Dataset front-end of the sampling path: the directory crawlers of the reference's
``loaders/bouncing_ball_loader.py:14-91`` and ``loaders/kitti_loader.py:15-100`` — same constructor arguments, same
``(indices, frames)`` items, same clip-grouping rules (quirks included), without cv2 / torchvision:

  * frames live under ``<dir>/<stage>/<parent>/<name>NNN.png``; a frame's index is ``int(parent + file[-7:-4])`` (folder
    digits followed by the three digits in front of ``.png``); frames are sorted by that number;
  * clips are taken every ``num_frames * stride`` frames; frame k of a clip is ``i + k * stride``; a clip stops at the first
    frame whose parent folder differs from the first frame's — ``BouncingBall`` keeps such a short clip (bouncing_ball_loader.py
    :66-77 appends whatever it collected), ``Kitti`` drops it (kitti_loader.py:77);
  * images are returned like ``cv2.imread``: uint8, HxWx3, **BGR**;
  * ``Kitti.transform`` (kitti_loader.py:87-100): centre square crop, then resize to FRAME_SIZE with bilinear interpolation
    at half-pixel centres (cv2.resize's default INTER_LINEAR geometry; cv2 computes it in 11-bit fixed point, so single
    pixels may differ by 1 LSB — cv2 is not installable here to pin that).
UCF-101 clips come from torchvision's video reader (PyAV) in the reference (prediction/predict.py:60-109): decoding video files
is not part of this path; extract frames to PNG folders and use ``--folder``.
"""
import os

import numpy as np
import torch.utils.data as data

from .config import parse_config_args


def _imread_bgr(path):
    from PIL import Image
    return np.ascontiguousarray(np.asarray(Image.open(path).convert("RGB"))[..., ::-1])


def _crawl(root, num_frames, stride, keep_short):
    img_names = []
    for d, _, files in os.walk(root):
        parent = d.split("/")[-1]
        for f in files:
            if f.endswith(".png"):
                img_names.append((int(parent + f[-7:-4]), os.path.join(d, f)))
    img_names = sorted(img_names, key=lambda x: x[0])
    indices, dataset = [], []
    step = num_frames * stride
    for i in range(0, len(img_names) - step + 1, step):
        for j in range(stride):
            if i % stride != j:
                continue
            index_list, frame_names = [], []
            first_parent = img_names[i][1].split("/")[-2]
            for k in range(num_frames):
                if img_names[i + k * stride][1].split("/")[-2] != first_parent:
                    break                                   # the clip must stay inside one folder
                index_list.append(img_names[i + k * stride][0])
                frame_names.append(img_names[i + k * stride][1])
            if keep_short or len(frame_names) == num_frames:
                indices.append(index_list)
                dataset.append(frame_names)
    return indices, dataset


class BouncingBall(data.Dataset):
    def __init__(self, num_frames=5, stride=1, dir="data/bouncing_ball", stage="raw", shuffle=True):
        self.stage = stage
        self.dir = os.path.join(dir, stage)
        self.num_frames = num_frames
        self.stride = stride
        self.indices, self.dataset = self.get_data(shuffle=shuffle)

    def get_data(self, shuffle):
        indices, dataset = _crawl(self.dir, self.num_frames, self.stride, keep_short=True)
        if shuffle:
            # like the reference, only the file lists are shuffled — `indices` keeps the crawl order (bouncing_ball_loader.py:86-89)
            np.random.shuffle(dataset)
        return indices, dataset

    def __getitem__(self, index):
        frames = np.stack([_imread_bgr(p) for p in self.dataset[index]], axis=0)
        return self.indices[index], frames

    def __len__(self):
        return len(self.dataset)


def resize_bilinear_u8(img, out_h, out_w):
    """cv2.resize(img, (out_w, out_h)) geometry (INTER_LINEAR, half-pixel centres, edge clamp) in float, rounded to uint8."""
    h, w = img.shape[:2]
    ys = (np.arange(out_h) + 0.5) * (h / out_h) - 0.5
    xs = (np.arange(out_w) + 0.5) * (w / out_w) - 0.5
    y0 = np.floor(ys).astype(np.int64); x0 = np.floor(xs).astype(np.int64)
    fy = (ys - y0)[:, None, None]; fx = (xs - x0)[None, :, None]
    y0c, y1c = np.clip(y0, 0, h - 1), np.clip(y0 + 1, 0, h - 1)
    x0c, x1c = np.clip(x0, 0, w - 1), np.clip(x0 + 1, 0, w - 1)
    f = img.astype(np.float64)
    top = f[y0c][:, x0c] * (1 - fx) + f[y0c][:, x1c] * fx
    bot = f[y1c][:, x0c] * (1 - fx) + f[y1c][:, x1c] * fx
    return np.clip(np.floor(top * (1 - fy) + bot * fy + 0.5), 0, 255).astype(np.uint8)


class Kitti(data.Dataset):
    def __init__(self, num_frames=5, stride=1, dir="data/kitti", stage="raw", shuffle=True):
        self.config, self.args = parse_config_args()
        self.stage = stage
        self.dir = os.path.join(dir, stage)
        self.num_frames = num_frames
        self.stride = stride
        self.indices, self.dataset = self.get_data(shuffle=shuffle)

    def get_data(self, shuffle):
        indices, dataset = _crawl(self.dir, self.num_frames, self.stride, keep_short=False)
        if shuffle:
            np.random.shuffle(dataset)
        return indices, dataset

    def transform(self, frame):
        h, w, _ = frame.shape
        if h < w:
            frame = frame[:, (w - h) // 2:(w - h) // 2 + h]
        else:
            frame = frame[(h - w) // 2:(h - w) // 2 + w, :]
        F = self.config.FRAME_SIZE
        return resize_bilinear_u8(frame, F, F)

    def __getitem__(self, index):
        frames = np.stack([self.transform(_imread_bgr(p)) for p in self.dataset[index]], axis=0)
        return self.indices[index], frames

    def __len__(self):
        return len(self.dataset)
