#!/usr/bin/env python3
"""bench.py — aligned Gbp/s of the exact WFA hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[2], the batch the metric is quoted on): per GPU, 1024 synthetic pairs,
10 kb target, query = target mutated at 5 % (60/20/20 sub/ins/del, geometric indels), score-only
mwf_wfa_exact semantics, default penalties.  One "step" = one pass of the hot path over that batch, with
the packed sequences already resident in HBM (torch tensors wrapped zero-copy by the C ABI).  Pairs are
independent, so ranks shard them with no data-path collective (weak scaling); the only collective is the
final RCCL all_gather of the fixed 12-byte (s, n_iter) records, inside the timed region.

Printed JSON (one line, rank 0): the driver contract fields plus
  roofline     — 48 algorithmic bytes per (penalty,diagonal) cell (7 int32 loads + 5 stores, reference
                 miniwfa.c:269-276) x cells per launch / HIP-event kernel time, against 8 TB/s
  cpu_baseline — the compiled reference (oracle/_ref, kind "reference") or our C restatement (kind "port")
                 on the host cores, bounded sample of the same batch (rank 0, N=1 only)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_CELL = 48          # score-only; 49 with traceback, 97 in the low-memory first pass
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def host_cores() -> int:
    """CPUs this process may actually use: the affinity mask, capped by the cgroup CPU quota (on the GPU boxes the
    container sees 256 logical CPUs but is throttled to 16; 256 threads then run slower than 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, -(-q // per)))
        except Exception:
            pass
    return max(1, n)


class _DevPtr:
    """Zero-copy torch view of a device buffer owned by the C library."""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pairs", type=int, default=1024, help="pairs per GPU")
    ap.add_argument("--len", type=int, default=10000, dest="tl")
    ap.add_argument("--div", type=float, default=0.05)
    ap.add_argument("--cigar", action="store_true", help="score+CIGAR (high-memory) instead of score-only")
    ap.add_argument("--block", type=int, default=0)
    ap.add_argument("--slots-per-cu", type=int, default=0)
    ap.add_argument("--cpu-sample", type=int, default=1024, help="pairs in the cpu_baseline sample (0: skip)")
    ap.add_argument("--long-pairs", type=int, default=1, help="also time the single-pair configs (C4-like 150 kb, MHC-like 5 Mb) on rank 0 at N=1")
    ap.add_argument("--seed", type=int, default=50000)
    args = ap.parse_args()

    import torch
    import miniwfa_amd as mw
    from miniwfa_amd.synth import synth_pair, PackedBatch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)  # "nccl" is RCCL on ROCm

    # ---- synthetic batch of this rank, resident in HBM before anything is timed
    pairs = [synth_pair(args.seed + rank * args.pairs + i, args.tl, args.div) for i in range(args.pairs)]
    pk = PackedBatch(pairs)
    d_seqs = torch.from_numpy(pk.seqs.copy()).to(dev)
    d_toff, d_qoff = torch.from_numpy(pk.t_off).to(dev), torch.from_numpy(pk.q_off).to(dev)
    d_tl, d_ql = torch.from_numpy(pk.tl).to(dev), torch.from_numpy(pk.ql).to(dev)
    stream = torch.cuda.current_stream(dev)
    eng = mw.Engine(local_rank, stream.cuda_stream)
    if args.block:
        eng.set("block", args.block)
    if args.slots_per_cu:
        eng.set("slots_per_cu", args.slots_per_cu)
    batch = eng.wrap(pk.n, d_seqs.data_ptr(), pk.total, d_toff.data_ptr(), d_tl.data_ptr(), d_qoff.data_ptr(), d_ql.data_ptr(),
                     pk.tl, pk.ql, keep=(d_seqs, d_toff, d_qoff, d_tl, d_ql))
    opt = mw.opt_init(flag=mw.MWF_F_CIGAR if args.cigar else 0)
    d_s = torch.as_tensor(_DevPtr(batch.dev_scores_ptr(), pk.n, "<i4"), device=dev)
    d_it = torch.as_tensor(_DevPtr(batch.dev_iters_ptr(), pk.n, "<i8"), device=dev)
    from miniwfa_amd.shard import gather_records

    kernel_ms = []

    def step(record: bool):
        batch.align(opt)                       # kernels enqueued on torch's current stream
        if args.cigar:
            batch.results()                    # CIGAR mode may have to retry pairs: needs the host in the loop
        if world > 1:                          # the result gather of the multi-GPU job (RCCL all_gather over xGMI)
            gather_records(dist, d_s, d_it, world * pk.n, device=dev)
        if record:
            torch.cuda.synchronize(dev)
            kernel_ms.append(eng.stats().kernel_ms)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step(False)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(False)
    fence()
    elapsed = time.perf_counter() - t0
    # kernel-only timing (HIP events on the launch stream), outside the wall-clock region so the per-step
    # event sync does not perturb it
    for _ in range(max(3, min(args.steps, 10))):
        step(True)
    s, n_iter, _ = batch.results()
    cells = int(n_iter.sum())
    assert (s >= 0).all(), "some pairs did not finish"

    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        tot = torch.tensor([pk.bases, cells], dtype=torch.int64, device=dev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        total_bases, total_cells = int(tot[0].item()), int(tot[1].item())
    else:
        total_bases, total_cells = pk.bases, cells

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    k_ms = float(np.mean(kernel_ms))
    bytes_per_cell = 49 if args.cigar else ALGO_BYTES_PER_CELL
    achieved = bytes_per_cell * cells / (k_ms * 1e-3) / 1e9
    out = {
        "metric": "aligned Gbp/s (q+t)",
        "value": total_bases * args.steps / elapsed / 1e9,
        "unit": "Gbp/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "int32", "data": "synthetic",
        "config": {
            "workload": f"{args.pairs} pairs/GPU x {args.tl} bp, {args.div:g} divergence, "
                        f"{'score+CIGAR high-mem' if args.cigar else 'score-only'} mwf_wfa_exact, default penalties (BASELINE configs[2])",
            "pairs_per_gpu": args.pairs, "target_len": args.tl, "divergence": args.div,
            "bases_per_gpu": pk.bases, "cells_per_gpu": cells, "mean_s": float(s.mean()),
            "kernel": {0: "wfa_batch_kernel (generic: one workgroup per pair, ring in HBM)", 1: "wfa_coop_kernel (one pair across the device)",
                       2: "wfa_band_kernel (one workgroup per pair, E/F in registers, H rows prefetched)"}.get(eng.stats().kernel_kind, "?"),
            "grid": eng.stats().grid, "block": eng.stats().block,
            "parallelism": f"pairs sharded over {world} GPU(s), RCCL all_gather of (s,n_iter)",
        },
        "gcells_per_s": total_cells * args.steps / elapsed / 1e9,
        "roofline": {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": None, "bytes_per_cell": bytes_per_cell, "cells_per_launch": cells, "kernel_ms": k_ms,
        },
    }
    if eng.stats().kernel_kind == 2:
        # The band kernel keeps E1/F1/E2/F2 in registers: of the 48 algorithmic bytes per cell only H crosses HBM (three
        # 4-byte loads + one 4-byte store, +1 traceback byte).  `achieved`/`frac` above follow SURVEY 8(d)'s definition and
        # can therefore exceed the HBM peak; the figures below are the kernel's own floor and what the PMC counters saw.
        kb = 17 if args.cigar else 16
        out["roofline"]["kernel_bytes_per_cell"] = kb
        out["roofline"]["achieved_kernel_bytes"] = kb * cells / (k_ms * 1e-3) / 1e9
        out["roofline"]["frac_kernel_bytes"] = out["roofline"]["achieved_kernel_bytes"] / HBM_PEAK_GBS
        out["roofline"]["note"] = ("frac uses SURVEY 8(d)'s 48 B/cell (the reference's 7 loads + 5 stores); this kernel moves only H "
                                   "(kernel_bytes_per_cell) and is bound by instruction issue + one barrier per penalty, not by HBM: DESIGN.md 4.2")
    traffic_file = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(traffic_file):
        try:
            tr = json.load(open(traffic_file))
            key = f"{args.pairs}x{args.tl}@{args.div:g}{'c' if args.cigar else 's'}"
            if key in tr:
                out["roofline"]["traffic"] = tr[key]["hbm_bytes_per_launch"]
                out["roofline"]["traffic_gbs"] = tr[key]["hbm_bytes_per_launch"] / (k_ms * 1e-3) / 1e9
                out["roofline"]["traffic_source"] = tr[key].get("source")
        except Exception:
            pass

    # ---- PCIe-inclusive rate (host buffers in, host results out) — reported beside, never as `value`
    t1 = time.perf_counter()
    b2 = eng.upload(pk)
    b2.align(opt)
    b2.results()
    out["pcie_inclusive_gbps"] = pk.bases / (time.perf_counter() - t1) / 1e9
    b2.free()

    # ---- CPU baseline: same pairs (a bounded sample), host cores of this box
    if world == 1 and args.cpu_sample > 0:
        from oracle.pyoracle import Oracle, Reference, make_opt
        n = min(args.cpu_sample, pk.n)
        cores = host_cores()
        o = make_opt(flag=1 if args.cigar else 0)
        orc = Oracle()
        arena = None
        if Reference.available():
            ref = Reference()
            fn, kind, arena = ref.exact_addr(), "reference", ref.arena_addrs()   # one pair per thread, private arena
        else:
            fn, kind = None, "port"
        threads = min(cores, n)
        cs, cit, sec = orc.batch(pk, o, threads, exact_fn=fn, n=n, arena=arena)   # pthread pool in C, one pair per thread at a time
        ok = bool((cs == s[:n]).all() and (cit == n_iter[:n]).all())
        sb = int(pk.tl[:n].sum() + pk.ql[:n].sum())
        out["cpu_baseline"] = {
            "value": sb / sec / 1e9, "unit": "Gbp/s", "cores": threads, "kind": kind,
            "sample": f"first {n} of the {pk.n} pairs, pthread pool of {threads} threads (= usable CPUs: affinity mask capped by the cgroup quota; os.cpu_count() = {os.cpu_count()}), "
                      f"{'lh3/miniwfa compiled -O3 -msse4.2 (oracle/_ref), a private kalloc arena per thread' if kind == 'reference' else 'oracle/mwf_oracle.c'}, {sec:.3f} s wall",
            "gcells_per_s": float(cit.sum()) / sec / 1e9,
            "gpu_matches_cpu_on_sample": ok,
        }
    # ---- the single-pair configs of BASELINE.json (configs[1] and configs[3]); stand-ins, see SURVEY.md §8d
    if world == 1 and args.long_pairs:
        lp = {}
        try:
            batch.free()
            for name, seed, tl_, p_, nl, lm, modes in (
                    ("c4_like_150kb", 2001, 150000, 0.035, 0, 0, (("score", {}), ("cigar_highmem", {"flag": 1}))),
                    ("mhc_like_5Mb", 2002, 5000000, 0.008, 3, 15000, (("cigar_lowmem_p5000", {"flag": 1, "step": 5000}),))):
                t_, q_ = synth_pair(seed, tl_, p_, nl, lm)
                bb = eng.upload(PackedBatch([(t_, q_)]))
                for label, kw in modes:
                    o_ = mw.opt_init(**kw)
                    bb.align(o_)
                    bb.results()               # first call also sizes the workspace (55 GB traceback arena for the 5 Mb pair)
                    bb.align(o_)
                    s_, it_, nc_ = bb.results()
                    st_ = eng.stats()
                    rec = {"s": int(s_[0]), "n_iter": int(it_[0]), "kernel_s": st_.kernel_ms * 1e-3, "cells_pass1": int(st_.cells_pass1),
                           "gbp_s": (len(t_) + len(q_)) / (st_.kernel_ms * 1e-3) / 1e9}
                    if kw.get("flag"):
                        cg = bb.cigar(0, int(nc_[0])).tolist()
                        rec["cigar_rescored_ok"] = mw.cigar2score(mw.opt_init(), cg) == (int(s_[0]), len(t_), len(q_))
                    lp.setdefault(name, {"tl": len(t_), "ql": len(q_)})[label] = rec
                bb.free()
        except Exception as e:  # never lose the headline line over the extras
            lp["error"] = repr(e)
        out["long_pairs"] = lp
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
