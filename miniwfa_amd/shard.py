"""Multi-GPU sharding of a batch of independent pairs.

Pairs are independent, so ranks split them with no data-path collective: each rank aligns its own contiguous
slice on its own GPU.  The only communication is the final gather of the fixed-size per-pair records
(s: int32, n_iter: int64) — `all_gather` over RCCL/xGMI on GPUs (backend "nccl"), over gloo in the CPU tests.
CIGARs, when wanted, are variable-length and travel as a second padded all_gather sized from the gathered n_cigar.
"""
from __future__ import annotations

import numpy as np


def shard_bounds(n: int, rank: int, world: int) -> tuple[int, int]:
    """[begin, end) of the pairs rank `rank` owns; sizes differ by at most one, earlier ranks take the remainder."""
    base, rem = divmod(n, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def gather_records(dist, s_local, it_local, n_total: int, device=None):
    """All ranks end up with the full (s[n_total], n_iter[n_total]) in global pair order.

    s_local / it_local: this rank's results as torch tensors (any device the backend supports) in shard order."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = [shard_bounds(n_total, r, world)[1] - shard_bounds(n_total, r, world)[0] for r in range(world)]
    cap = max(sizes) if sizes else 0
    dev = device if device is not None else s_local.device
    pad_s = torch.full((cap,), -2, dtype=torch.int32, device=dev)
    pad_it = torch.zeros((cap,), dtype=torch.int64, device=dev)
    pad_s[:sizes[rank]] = s_local.to(dev)
    pad_it[:sizes[rank]] = it_local.to(dev)
    all_s = [torch.empty_like(pad_s) for _ in range(world)]
    all_it = [torch.empty_like(pad_it) for _ in range(world)]
    dist.all_gather(all_s, pad_s)
    dist.all_gather(all_it, pad_it)
    s = torch.cat([all_s[r][:sizes[r]] for r in range(world)])
    it = torch.cat([all_it[r][:sizes[r]] for r in range(world)])
    return s, it


def gather_cigars(dist, cigars_local, n_total: int, device=None):
    """Variable-length payload: list (shard order) of uint32 numpy arrays -> list for all n_total pairs on every rank."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = [shard_bounds(n_total, r, world)[1] - shard_bounds(n_total, r, world)[0] for r in range(world)]
    dev = device if device is not None else torch.device("cpu")
    lens_local = torch.tensor([len(c) for c in cigars_local], dtype=torch.int64, device=dev)
    cap = max(sizes) if sizes else 0
    pad_len = torch.zeros((cap,), dtype=torch.int64, device=dev)
    pad_len[:sizes[rank]] = lens_local
    all_len = [torch.empty_like(pad_len) for _ in range(world)]
    dist.all_gather(all_len, pad_len)
    words = [int(all_len[r][:sizes[r]].sum()) for r in range(world)]
    wcap = max(words) if words else 0
    flat = np.concatenate([np.asarray(c, dtype=np.uint32) for c in cigars_local]) if cigars_local and words[rank] else np.zeros(0, dtype=np.uint32)
    pad = torch.zeros((max(wcap, 1),), dtype=torch.int64, device=dev)
    pad[:words[rank]] = torch.from_numpy(flat.astype(np.int64)).to(dev)
    all_w = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(all_w, pad)
    out = []
    for r in range(world):
        off = 0
        w = all_w[r].cpu().numpy()
        for ln in all_len[r][:sizes[r]].cpu().numpy():
            out.append(w[off:off + int(ln)].astype(np.uint32))
            off += int(ln)
    return out
