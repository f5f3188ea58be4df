"""Clip sharding over the GPUs of a node (one process per GPU, torch.distributed; backend "nccl" is RCCL over
xGMI on ROCm).  The path shards by CLIP: inside a clip every frame depends on the previous one
(prediction/predict.py:193-196) and the DDIM loop is sequential in t, so there is no data-path collective —
clips are dealt to ranks, each rank samples its clips with its own full weight replica, and ONE all-gather
reassembles the generated clips at the end (<= 1 MB per clip: latency-bound, link bandwidth irrelevant).
Per-clip seeds (base + clip index) make results independent of the world size.
"""
import torch
import torch.distributed as dist


# tests: run the collective even in a one-rank group (an RCCL all_gather of one rank's device tensor is still an RCCL call)
_FORCE_COLLECTIVE = False


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_range(n_clips, rank, world_size):
    """Contiguous block of clip indices for `rank` (sizes differ by at most one)."""
    base, rem = divmod(n_clips, world_size)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def clip_seeds(base_seed, start, stop):
    return [base_seed + c for c in range(start, stop)]


def gather_clips(local, n_clips):
    """local: (c_local, ...) tensor of this rank's clips -> (n_clips, ...) on every rank, in clip order.
    One all_gather; ragged shards are padded to the largest shard and trimmed."""
    rank, ws = world()
    if ws == 1 and not _FORCE_COLLECTIVE:
        return local
    sizes = [shard_range(n_clips, r, ws) for r in range(ws)]
    cmax = max(b - a for a, b in sizes)
    pad = local
    if local.shape[0] < cmax:
        pad = torch.cat([local, local.new_zeros((cmax - local.shape[0],) + tuple(local.shape[1:]))])
    pad = pad.contiguous()
    dev = pad.device
    if dist.get_backend() == "gloo" and pad.is_cuda:
        pad = pad.cpu()          # rehearsal backend: gloo moves host memory; RCCL ("nccl") takes the device tensor as is
    out = [torch.empty_like(pad) for _ in range(ws)]
    dist.all_gather(out, pad)
    return torch.cat([o[: b - a] for o, (a, b) in zip(out, sizes)]).to(dev)


def gather_clips_packed(tensors, n_clips):
    """Several per-clip tensors (same leading clip dimension, any dtypes) in ONE all_gather: each clip's payloads are
    viewed as bytes and laid side by side, gathered once, and split back (predict.main: latents f32 + frames u8)."""
    rank, ws = world()
    if ws == 1 and not _FORCE_COLLECTIVE:
        return list(tensors)
    c = tensors[0].shape[0]
    widths = [int(torch.tensor(t.shape[1:]).prod()) * t.element_size() for t in tensors]
    flat = [t.contiguous().reshape(c, w // t.element_size()).view(torch.uint8) for t, w in zip(tensors, widths)]
    out = gather_clips(torch.cat(flat, dim=1), n_clips)
    res, off = [], 0
    for t, w in zip(tensors, widths):
        piece = out[:, off:off + w].contiguous().view(t.dtype).reshape((n_clips,) + tuple(t.shape[1:]))
        res.append(piece)
        off += w
    return res


def gather_rows(local):
    """evaluation/fvd_2.py:103-107 all_gather: every rank's (n, ...) rows concatenated in rank order (equal n per rank there;
    here ragged counts are allowed: one small all_gather of the counts, then the padded payloads)."""
    rank, ws = world()
    if ws == 1:
        return local
    dev = local.device
    host = dist.get_backend() == "gloo"
    n = torch.tensor([local.shape[0]], dtype=torch.int64, device="cpu" if host else dev)
    counts = [torch.zeros_like(n) for _ in range(ws)]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    cmax = max(counts)
    pad = local
    if local.shape[0] < cmax:
        pad = torch.cat([local, local.new_zeros((cmax - local.shape[0],) + tuple(local.shape[1:]))])
    pad = pad.contiguous()
    if host and pad.is_cuda:
        pad = pad.cpu()
    out = [torch.empty_like(pad) for _ in range(ws)]
    dist.all_gather(out, pad)
    return torch.cat([o[:c] for o, c in zip(out, counts)]).to(dev)
