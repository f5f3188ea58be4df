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
UCF-101 (prediction/predict.py:60-109, prediction/predict_text.py:90-139): the reference builds ``torchvision.datasets.UCF101`` —
class folders, the ``ucfTrainTestlist`` fold files, ``VideoClips`` indexing with ``frames_per_clip`` / ``step_between_clips`` /
``frame_rate`` resampling — over ``.avi`` files decoded by PyAV.  ``UCF101Frames`` keeps that indexing (torchvision 0.12,
environment.yml:106, restated from its published ``video_utils.VideoClips``) over PRE-EXTRACTED frames: a video is a directory of
frame images or one ``.npy`` array (T,H,W,3 RGB uint8) named like the ``.avi`` it came from; video decoding itself is not part of
this path (no PyAV / torchvision offline).  ``ucf_transform`` is the reference's transform (nearest resize to FRAME_SIZE, RGB -> BGR).
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


# ---- UCF-101 ---------------------------------------------------------------------------------------------------------------------
UCF_FPS = 25.0        # every UCF-101 video is 25 fps (the value PyAV reports to VideoClips in the reference)


def ucf_dirs(dataset):
    """prediction/predict.py:60-72: dataset flag -> (video directory, fold-file directory); 'Invalid dataset name' otherwise"""
    if dataset.endswith("wallpushups"):
        d = "data/UCF-101/UCF-101-wallpushups"
    elif dataset.endswith("workout"):
        d = "data/UCF-101/UCF-101-workout"
    elif dataset.endswith("instruments"):
        d = "data/UCF-101/UCF-101-instruments"
    elif dataset == "ucf":
        d = "data/UCF-101/UCF-101"
    else:
        raise ValueError("Invalid dataset name")
    return d, "data/UCF101TrainTestSplits-RecognitionTask/ucfTrainTestlist"


def split_class_names(classes):
    """prediction/predict_text.py:18-32 splitClassNames: 'WallPushups' -> 'Wall Pushups' (a space in front of every capital)"""
    out = []
    for s in classes:
        words, cur = [], ""
        for ch in s:
            if ch.isupper():
                words.append(cur)
                cur = ch
            else:
                cur += ch
        words.append(cur)
        words.remove("")                     # as the reference: x.remove('') — raises for a name that starts lower-case
        out.append(" ".join(words))
    return out


def find_classes(directory):
    """prediction/predict_text.py:34-46 -> (class names split into words, {index: name})"""
    classes = sorted(e.name for e in os.scandir(directory) if e.is_dir())
    classes = split_class_names(classes)
    if not classes:
        raise FileNotFoundError("Couldn't find any class folder in %s." % directory)
    return classes, {i: c for i, c in enumerate(classes)}


def resample_video_idx(num_frames, original_fps, new_fps):
    """torchvision VideoClips._resample_video_idx: a slice when the step is an integer, else floor(arange(n) * step)"""
    step = float(original_fps) / new_fps
    if step.is_integer():
        return slice(None, None, int(step))
    return np.floor(np.arange(num_frames, dtype=np.float32) * np.float32(step)).astype(np.int64)


def clips_for_video(n_video_frames, frames_per_clip, step_between_clips, fps=UCF_FPS, frame_rate=None):
    """torchvision VideoClips.compute_clips_for_video on frame numbers: -> list of frame-index lists, one per clip"""
    import math
    if frame_rate is None:
        frame_rate = fps
    total = n_video_frames * (float(frame_rate) / fps)
    idxs = resample_video_idx(int(math.floor(total)), fps, frame_rate)
    pts = np.arange(n_video_frames)[idxs]
    if isinstance(idxs, np.ndarray):
        pts = pts[: len(idxs)]
    n = (len(pts) - frames_per_clip) // step_between_clips + 1
    return [pts[i * step_between_clips: i * step_between_clips + frames_per_clip].tolist() for i in range(max(n, 0))]


def ucf_transform(frame_size):
    """predict.py:76-87: (T,H,W,C) uint8 RGB -> nearest resize to FRAME_SIZE x FRAME_SIZE (F.interpolate's default mode on the
    uint8 tensor: src = floor(dst * in / out)) -> RGB to BGR"""
    def tf(video):
        T, H, W, _ = video.shape
        ys = np.minimum((np.arange(frame_size) * (H / frame_size)).astype(np.int64), H - 1)
        xs = np.minimum((np.arange(frame_size) * (W / frame_size)).astype(np.int64), W - 1)
        return np.ascontiguousarray(video[:, ys][:, :, xs][..., ::-1])
    return tf


class UCF101Frames(data.Dataset):
    """torchvision.datasets.UCF101(root, annotation_path, frames_per_clip, step_between_clips=1, frame_rate=None, fold=1, train=True,
    transform=None) over pre-extracted frames.  Items are (video (T,H,W,3) uint8 RGB [transformed], None, label) like torchvision's
    (video, audio, label); `collate` is the reference's custom_collate -> (label, video)."""
    IMG = (".png", ".jpg", ".jpeg")

    def __init__(self, root, annotation_path, frames_per_clip, step_between_clips=1, frame_rate=None, fold=1, train=True, transform=None,
                 fps=UCF_FPS, num_workers=0):
        if not 1 <= fold <= 3:
            raise ValueError("fold should be between 1 and 3, got %s" % fold)
        self.root, self.transform, self.frames_per_clip = root, transform, frames_per_clip
        self.classes = sorted(e.name for e in os.scandir(root) if e.is_dir())
        if not self.classes:
            raise FileNotFoundError("Couldn't find any class folder in %s." % root)
        self.class_to_idx = {c: i for i, c in enumerate(self.classes)}
        self.samples = []                      # (video path, label): every video of every class, sorted like make_dataset
        for c in self.classes:
            for name in sorted(os.listdir(os.path.join(root, c))):
                p = os.path.join(root, c, name)
                if os.path.isdir(p) or name.endswith(".npy"):
                    self.samples.append((p, self.class_to_idx[c]))
        avi = lambda p: os.path.join(os.path.basename(os.path.dirname(p)), os.path.splitext(os.path.basename(p))[0] + ".avi")
        fn = os.path.join(annotation_path, "%slist%02d.txt" % ("train" if train else "test", fold))
        with open(fn) as f:
            selected = {ln.strip().split(" ")[0] for ln in f if ln.strip()}
        self.indices = [i for i, (p, _) in enumerate(self.samples) if avi(p) in selected]          # UCF101._select_fold
        self.clips = []                        # (position in self.indices, frame numbers)
        for vi, i in enumerate(self.indices):
            for fr in clips_for_video(self._n_frames(self.samples[i][0]), frames_per_clip, step_between_clips, fps, frame_rate):
                self.clips.append((vi, fr))

    def _frame_files(self, p):
        return sorted(f for f in os.listdir(p) if f.lower().endswith(self.IMG))

    def _n_frames(self, p):
        if p.endswith(".npy"):
            return int(np.load(p, mmap_mode="r").shape[0])
        return len(self._frame_files(p))

    def __len__(self):
        return len(self.clips)

    def __getitem__(self, idx):
        vi, frames = self.clips[idx]
        p, label = self.samples[self.indices[vi]]
        if p.endswith(".npy"):
            video = np.ascontiguousarray(np.load(p, mmap_mode="r")[frames])
        else:
            from PIL import Image
            files = self._frame_files(p)
            video = np.stack([np.asarray(Image.open(os.path.join(p, files[k])).convert("RGB")) for k in frames])
        if self.transform is not None:
            video = self.transform(video)
        return video, None, label

    @staticmethod
    def collate(batch):
        """predict.py:90-95 custom_collate: [(video, _, label)] -> (labels (B,), videos (B,T,H,W,C))"""
        import torch
        return torch.tensor([b[2] for b in batch]), torch.from_numpy(np.stack([b[0] for b in batch]))
