"""Multi-GPU sharding of a batch of independent pairs.

Pairs are independent (reference main.c:67-72 is a loop without carried state), so ranks split them with no data-path
collective: each rank aligns its own share on its own GPU.  The deal is by work, not by position: pairs are sorted by
length and dealt longest-first to the rank with the least work so far (work ~ cells ~ (tl+ql)^2 at equal divergence) —
the same deal `mwf_wfa_batch_multi` makes across the devices of one process (miniwfa_amd/csrc/mwf_engine.cpp), so a
ragged batch does not leave one GPU with all the long pairs.  Every rank computes the same deal from the lengths alone.

The only communication is the final gather of the fixed-size per-pair records: ONE `all_gather_into_tensor` of
(s, n_iter) packed as two int64 per pair — RCCL over xGMI on GPUs (backend "nccl"), gloo in the CPU tests.  CIGARs,
when wanted, are variable-length: their lengths travel in one more fixed-record gather, the words in ONE grouped send/recv exchange,
each message exactly its sender's size (no padding to the largest payload, no per-rank broadcast loop).
"""
from __future__ import annotations

import numpy as np


def shard_bounds(n: int, rank: int, world: int) -> tuple[int, int]:
    """[begin, end) of a contiguous split (uniform batches: bench.py's weak-scaling workload generates its own share)."""
    base, rem = divmod(n, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def deal_pairs(lengths, world: int) -> list[np.ndarray]:
    """Work-balanced deal: lengths[i] = tl+ql of pair i -> for every rank the (ascending) ids of its pairs.

    Longest first, each to the rank with the least work so far; ties go to the lowest rank.  Deterministic."""
    lengths = np.asarray(lengths, dtype=np.int64)
    order = np.argsort(-lengths, kind="stable")
    load = np.zeros(world, dtype=np.float64)
    share: list[list[int]] = [[] for _ in range(world)]
    for i in order:
        r = int(np.argmin(load))
        w = float(lengths[i]) + 1.0
        load[r] += w * w
        share[r].append(int(i))
    return [np.array(sorted(s), dtype=np.int64) for s in share]


def gather_records(dist, s_local, it_local, n_total: int, device=None, deal=None):
    """All ranks end up with the full (s[n_total], n_iter[n_total]) in global pair order.

    s_local / it_local: this rank's results (torch tensors on any device the backend supports), in the order of its
    share — `deal[rank]` if a deal is given, else the contiguous `shard_bounds` slice.  One collective."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    if deal is None:
        ids = [np.arange(*shard_bounds(n_total, r, world), dtype=np.int64) for r in range(world)]
    else:
        ids = [np.asarray(d, dtype=np.int64) for d in deal]
    sizes = [len(x) for x in ids]
    cap = max(sizes) if sizes else 0
    dev = device if device is not None else s_local.device
    rec = torch.zeros((max(cap, 1), 2), dtype=torch.int64, device=dev)
    rec[:, 0] = -2                                   # padding: "not a result"
    if sizes[rank]:
        rec[:sizes[rank], 0] = s_local.to(dev).to(torch.int64)
        rec[:sizes[rank], 1] = it_local.to(dev)
    out = torch.empty((world * max(cap, 1), 2), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(out, rec)
    out = out.view(world, max(cap, 1), 2)
    s = torch.empty((n_total,), dtype=torch.int32, device=dev)
    it = torch.empty((n_total,), dtype=torch.int64, device=dev)
    for r in range(world):
        if sizes[r]:
            idx = torch.from_numpy(ids[r]).to(dev)
            s[idx] = out[r, :sizes[r], 0].to(torch.int32)
            it[idx] = out[r, :sizes[r], 1]
    return s, it


def gather_cigars(dist, cigars_local, n_total: int, device=None, deal=None, dst=None):
    """Variable-length payload: list (share order) of uint32 numpy arrays -> list for all n_total pairs (on every rank, or — `dst` given —
    on rank `dst` only; the others get None).

    Two steps (fixed records first, payloads "via grouped send/recv sized from the gathered n_cigar"), neither padded to the
    largest payload:
      1. the per-pair lengths in ONE all_gather of fixed-size records (one int64 per pair, padded to the largest share — a few bytes per pair);
      2. the words in ONE grouped point-to-point exchange (`batch_isend_irecv`: on RCCL a single ncclGroupStart/End of sends and receives over
         xGMI, each exactly as long as its sender's payload; gloo runs the same ops in the CPU tests).  Every rank posts its sends and receives
         at once — no rank-by-rank broadcast loop, no host synchronisation between peers: one wait, one copy to the host at the end."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    if deal is None:
        ids = [np.arange(*shard_bounds(n_total, r, world), dtype=np.int64) for r in range(world)]
    else:
        ids = [np.asarray(d, dtype=np.int64) for d in deal]
    sizes = [len(x) for x in ids]
    dev = device if device is not None else torch.device("cpu")
    cap = max(max(sizes) if sizes else 0, 1)
    pad_len = torch.zeros((cap,), dtype=torch.int64, device=dev)
    if sizes[rank]:
        pad_len[:sizes[rank]] = torch.tensor([len(c) for c in cigars_local], dtype=torch.int64, device=dev)
    all_len = torch.empty((world * cap,), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(all_len, pad_len)
    all_len = all_len.view(world, cap).cpu().numpy()          # (the one host read: the receive buffers are sized from it)
    words = [int(all_len[r, :sizes[r]].sum()) for r in range(world)]
    flat = np.concatenate([np.asarray(c, dtype=np.uint32) for c in cigars_local]) if words[rank] else np.zeros(0, dtype=np.uint32)
    mine = torch.from_numpy(flat.view(np.int32).copy()).to(dev)
    receivers = list(range(world)) if dst is None else [int(dst)]
    bufs = {rank: mine}
    ops = []
    if rank in receivers:
        for r in range(world):
            if r != rank and words[r]:
                bufs[r] = torch.empty((words[r],), dtype=torch.int32, device=dev)
                ops.append(dist.P2POp(dist.irecv, bufs[r], r))
    if words[rank]:
        for r in receivers:
            if r != rank:
                ops.append(dist.P2POp(dist.isend, mine, r))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    if rank not in receivers:
        return None
    payload = [bufs[r].cpu().numpy().view(np.uint32) if words[r] else np.zeros(0, dtype=np.uint32) for r in range(world)]
    out: list = [None] * n_total
    for r in range(world):
        off = 0
        for j, ln in enumerate(all_len[r, :sizes[r]]):
            out[int(ids[r][j])] = payload[r][off:off + int(ln)].copy()
            off += int(ln)
    return out
