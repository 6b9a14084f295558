#!/usr/bin/env python3
"""bench.py — aligned Gbp/s of the exact WFA hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W                 # BASELINE configs[2]: 1024 x 10 kb per GPU, weak scaling
    python bench.py --config 5 --gpus N --steps K --warmup W      # BASELINE configs[4]: 10 000 x 50 kb in total, strong scaling
    N > 1: one process per GPU.  Either launched by  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment), or bare — `python bench.py --gpus N` without WORLD_SIZE
    re-executes itself under torch.distributed.run on 127.0.0.1 with a free port (launcher_argv()).
    MWF_BENCH_BACKEND=gloo: a DRY RUN of the N-rank plumbing on CPU (deal, per-rank step, ONE all_gather of the records, max-over-ranks
    timing, the JSON line) on a tiny workload with the oracle standing in for the GPU; the line says "dry_run": true — it is a test of
    the launcher and the gather (tests/test_bench_launch.py), never a measurement.

Default workload (the batch BASELINE.json's metric is quoted on): per GPU, 1024 synthetic pairs, 10 kb target, query =
target mutated at 5 % (60/20/20 sub/ins/del, geometric indels), score-only mwf_wfa_exact semantics, default
penalties.  One "step" = one pass of the hot path over that batch: the alignment kernels on sequences already
resident in HBM (torch tensors wrapped zero-copy by the C ABI), every pair's (s, n_iter) record back on the host
(mwf_gpu_batch_results: one device-to-host copy, and any pair that needs a re-run gets it inside the step), and — at
N > 1 — the one RCCL collective that gathers the records of all ranks.  `value` = bases aligned by all ranks / wall.
The host-buffers-in to host-results-out rate of the same batch (PCIe both ways, warmed, pooled allocations) is
printed beside it as `end_to_end_gbps`; it is never `value`.

Printed JSON (one line, rank 0): the driver contract fields plus
  roofline     — ONE definition everywhere in the line (top level and long_pairs.*): achieved = HBM bytes per launch by the PMC
                 counters (profiles/traffic.json: 2048 B x FETCH_SIZE + 1024 B x WRITE_SIZE, units calibrated on this device) /
                 the kernel time measured live with HIP events on the launch stream; frac = achieved / 8 TB/s.  Named side fields:
                 `own_bytes_frac` (the bytes the kernel itself must move per cell), `nominal_48B` (the reference's 48 /
                 49 / 97 B per cell x cells / time — exceeds 1 for kernels that keep four of the five arrays on chip), issue fractions
  cpu_baseline — the compiled reference (oracle/_ref, kind "reference") or our C restatement (kind "port")
                 on the host cores, bounded sample of the same batch (rank 0, N=1 only)
  call_latency_us, chain_mode, short_reads, long_pairs, peak_device_bytes, n_retries — extras (rank 0, N=1 only)
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TRAFFIC = {}
try:
    TRAFFIC = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
except Exception:
    pass

ALGO_BYTES_PER_CELL = 48          # score-only; 49 with traceback, 97 in the low-memory first pass
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# VALU issue peak: 256 CUs x 4 SIMD-32 per CU x 2.4 GHz, a wave64 VALU instruction occupies its SIMD for 2 cycles
# (MI355X_MICROARCH.md: "issues each VALU instruction over 2 cycles (32 lanes/cycle x 2)", v_fma_f32 row: 2 cyc)
VALU_PEAK_GINST = 256 * 4 * 2.4 / 2
ISSUE_PEAK_GINST = 256 * 4 * 2.4 / 1.28   # what 1024 SIMDs issue with four ready waves each: one instruction per 1.28 cycles (measured, any VALU/SALU mix)


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


KERNEL_NAMES = {0: "wfa_batch_kernel (generic: one workgroup per pair, ring in HBM)", 1: "wfa_sys_kernel (one pair across the device, systolic hand-offs)",
                2: "wfa_band_kernel (one workgroup per pair, E/F in registers, 32-bit H rows in HBM)",
                3: "wfa_band2_kernel (one workgroup per pair, recurrence in packed int16 (two columns per VOP3P instruction), E/F in registers, 16-bit H rows in HBM, sequences in LDS at 2 bits per base)"}


def call_latency(mw, synth_pair, reps=40):
    """Per-call time of the drop-in mwf_wfa_exact (host strings in, mwf_rst_t out) next to the compiled reference's."""
    out = {}
    try:
        from oracle.pyoracle import Reference, make_opt
        ref = Reference(arena=True) if Reference.available() else None
    except Exception:
        ref = None
    for tl in (200, 1000, 2000, 10000):
        t, q = synth_pair(123, tl, 0.05)
        for label, flag in (("score", 0), ("cigar", 1)):
            o = mw.opt_init(flag=flag)
            for _ in range(3):
                mw.wfa_exact(t, q, o)
            t0 = time.perf_counter()
            for _ in range(reps):
                mw.wfa_exact(t, q, o)
            rec = {"gpu_us": (time.perf_counter() - t0) / reps * 1e6}
            if ref is not None:
                ro = make_opt(flag=flag)
                ref.align(t, q, ro)
                n = max(3, reps // 4)
                t0 = time.perf_counter()
                for _ in range(n):
                    ref.align(t, q, ro)
                rec["cpu_reference_us"] = (time.perf_counter() - t0) / n * 1e6
            out[f"{tl}bp_{label}"] = rec
    # The caller that keeps the reference's one-pair-per-call shape (main.c:67-72) but has SEVERAL host threads: every thread takes its own
    # pooled engine (stream + device pools), so their single-pair kernels share the device — the call RATE of 16 threads looping
    # mwf_wfa_exact on 1 kb pairs against one thread's (ctypes drops the GIL inside the call; the Python between calls is ~10 us of a 300 us call).
    try:
        pairs = [synth_pair(5000 + i, 1000, 0.05) for i in range(16)]
        o = mw.opt_init()

        def loop(k, n_calls, box):
            t, q = pairs[k]
            for _ in range(n_calls):
                box.append(mw.wfa_exact(t, q, o)[0])

        def rate(n_threads, n_calls):
            box = []
            th = [threading.Thread(target=loop, args=(k, n_calls, box)) for k in range(n_threads)]
            t0 = time.perf_counter()
            for x in th:
                x.start()
            for x in th:
                x.join()
            return n_threads * n_calls / (time.perf_counter() - t0)

        rate(16, 5)   # sixteen warm engines
        r1, r16 = rate(1, 100), rate(16, 100)
        out["threads16_1000bp_score"] = {"calls_per_s_one_thread": r1, "calls_per_s_16_threads": r16, "ratio": r16 / r1,
                                         "what": "16 host threads each looping mwf_wfa_exact on its own 1 kb pair (one pooled engine per thread, kernels of different threads share the device)"}
        # the same loop with "submit" as its body and the waits behind it (mwf_wfa_submit / mwf_wfa_wait: one dispatcher thread, one batch launch per ~100 us of submissions)
        many = [pairs[i % 16] for i in range(1600)]
        for _ in range(2):
            t0 = time.perf_counter()
            jobs = [mw.wfa_submit(t, q, o) for t, q in many]
            res = [j.wait() for j in jobs]
            w = time.perf_counter() - t0
        out["submit_wait_1000bp_score"] = {"calls_per_s": len(many) / w, "ratio_to_one_thread_calls": len(many) / w / r1, "ok": all(r[0] == res[i % 16][0] for i, r in enumerate(res)),
                                           "what": "one host thread: 1600 x mwf_wfa_submit, then 1600 x mwf_wfa_wait (Python binding overhead included)"}
        # ... and the same sixteen looping threads with MWF_COALESCE_US=150 (read once per process: a child process): calls of different threads that
        # arrive within 150 us share one batch launch (mwf_async.cpp) — no change to the calling program at all
        try:
            import subprocess
            code = ("import sys, threading, time, json\nsys.path.insert(0, %r)\nimport torch\nimport miniwfa_amd as mw\nfrom miniwfa_amd.synth import synth_pair\n"
                    "pairs = [synth_pair(5000 + i, 1000, 0.05) for i in range(16)]\no = mw.opt_init()\n"
                    "def loop(k, n):\n    t, q = pairs[k]\n    for _ in range(n): mw.wfa_exact(t, q, o)\n"
                    "def rate(nt, n):\n    th = [threading.Thread(target=loop, args=(k, n)) for k in range(nt)]\n    t0 = time.perf_counter()\n    [x.start() for x in th]; [x.join() for x in th]\n    return nt * n / (time.perf_counter() - t0)\n"
                    "rate(16, 5)\nprint(json.dumps({'calls_per_s_16_threads': rate(16, 100), 'async_stats': mw.async_stats()}))\n") % ROOT
            r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=dict(os.environ, MWF_COALESCE_US="150"))
            rec = json.loads(r.stdout.strip().splitlines()[-1])
            rec["ratio_to_one_thread_calls"] = rec["calls_per_s_16_threads"] / r1
            rec["what"] = "the same sixteen threads with MWF_COALESCE_US=150 in the environment: concurrent mwf_wfa_exact calls share batch launches (async_stats: batches run, calls in them)"
            out["threads16_coalesced_1000bp_score"] = rec
        except Exception as e:
            out["threads16_coalesced_1000bp_score"] = {"error": repr(e)}
    except Exception as e:
        out["threads16_1000bp_score"] = {"error": repr(e)}
    return out


def chain_mode(mw, synth_pair, reps=8):
    """Per-call time of the drop-in mwf_wfa_chain (reference miniwfa.c:850-896: k-mer chaining on the host, every gap fill in one device batch) with
    CIGAR on a warm engine, next to the compiled reference's; answers compared."""
    out = {}
    try:
        from oracle.pyoracle import Reference, make_opt
        ref = Reference(arena=True) if Reference.available() else None
    except Exception:
        ref = None
    for tl, p in ((5000, 0.05), (30000, 0.04), (100000, 0.03)):
        t, q = synth_pair(4242, tl, p, 2, 800)
        o = mw.opt_init(flag=1)
        for _ in range(3):
            s, _, cig = mw.wfa_chain(t, q, o)
        t0 = time.perf_counter()
        for _ in range(reps):
            mw.wfa_chain(t, q, o)
        rec = {"gpu_ms": (time.perf_counter() - t0) / reps * 1e3, "s": int(s)}
        if ref is not None:
            ro = make_opt(flag=1)
            es, _, ecig = ref.chain(t, q, ro)
            t0 = time.perf_counter()
            for _ in range(max(2, reps // 2)):
                ref.chain(t, q, ro)
            rec["cpu_reference_ms"] = (time.perf_counter() - t0) / max(2, reps // 2) * 1e3
            rec["matches_reference"] = bool(s == es and (None if cig is None else list(cig)) == ecig)
        out[f"{tl}bp@{p}"] = rec
    # many records in chain mode (the reference's test program loops mwf_wfa_chain over its records, main.c:67-72): one call per pair against mwf_wfa_chain_batch —
    # the chaining of the pairs on a few host threads, the gap fills of ALL pairs in one device batch
    try:
        for n, tl, p in ((200, 5000, 0.05), (100, 30000, 0.04)):
            pairs = [synth_pair(9000 + i, tl, p, 2, 800) for i in range(n)]
            o = mw.opt_init(flag=1)
            mw.wfa_chain_batch(pairs[:4], o)
            t0 = time.perf_counter()
            a = [mw.wfa_chain(t, q, o) for t, q in pairs]
            t1 = time.perf_counter()
            b = mw.wfa_chain_batch(pairs, o)
            t2 = time.perf_counter()
            rec = {"per_pair_call_ms": (t1 - t0) / n * 1e3, "chain_batch_ms_per_pair": (t2 - t1) / n * 1e3, "same_answers": a == b}
            if ref is not None:
                ro = make_opt(flag=1)
                t0 = time.perf_counter()
                for t, q in pairs[:20]:
                    ref.chain(t, q, ro)
                rec["cpu_reference_ms_per_pair"] = (time.perf_counter() - t0) / 20 * 1e3
            out[f"batch_{n}x{tl}bp@{p}"] = rec
    except Exception as e:  # noqa: BLE001
        out["batch_error"] = repr(e)
    return out


def short_reads(mw, synth_pair, PackedBatch, reps=5):
    """Batches of read-sized pairs at 5 % (score-only) on the one-wave-per-pair lane kernel (mwf_lane.hip): `kernel_gbps` from the
    library's HIP events around the kernels of one align call, `step_gbps` from the wall clock around align() + results() (what a
    caller sees with the batch resident: host work, launches, the re-runs of pairs whose window outgrew the kernel, records back)."""
    out = {}
    for n, tl in ((40000, 150), (20000, 250)):
        pairs = [synth_pair(7000 + i, tl, 0.05) for i in range(n)]
        bp = sum(len(t) + len(q) for t, q in pairs)
        eng = mw.Engine(0)
        b = eng.upload(PackedBatch(pairs))
        o = mw.opt_init()
        ms, wall = [], []
        for it in range(reps + 2):
            t0 = time.perf_counter()
            b.align(o)
            b.results()
            t1 = time.perf_counter()
            if it >= 2:
                ms.append(eng.stats().kernel_ms)
                wall.append((t1 - t0) * 1e3)
        st = eng.stats()
        # kernel_*: HIP events around the first launches of an align call (pairs re-run afterwards are outside them); step_*: wall clock of
        # align() + results() with the batch resident — host classification, launches, every re-run, the records on the host
        rec = {"kernel_gbps": bp / (sum(ms) / len(ms)) / 1e6, "kernel_ms": sum(ms) / len(ms), "step_gbps": bp / (sum(wall) / len(wall)) / 1e6,
               "step_ms": sum(wall) / len(wall), "re_run": int(st.n_retries)}
        b.free()
        # one shot, host to host: a FRESH batch on the warm engine — pack + H2D, its first align (plan and all), the records back, free: what a
        # user who aligns a batch once pays (round 4 reported only the repeated aligns of a resident batch)
        shots, firsts, reruns = [], [], 0
        for r in range(4):
            pk = PackedBatch([synth_pair(7000 + 100000 * (r + 1) + i, tl, 0.05) for i in range(n)])
            t0 = time.perf_counter()
            b = eng.upload(pk)
            t1 = time.perf_counter()
            b.align(o)
            b.results()
            t2 = time.perf_counter()
            b.free()
            t3 = time.perf_counter()
            if r:
                shots.append((t3 - t0) * 1e3), firsts.append((t2 - t1) * 1e3)
                reruns += int(eng.stats().n_retries)
        rec.update({"one_shot_ms": sum(shots) / len(shots), "one_shot_gbps": bp / (sum(shots) / len(shots)) / 1e6, "first_align_ms": sum(firsts) / len(firsts),
                    "one_shot_re_run_mean": reruns / len(shots)})
        out[f"{n}x{tl}bp"] = rec
        eng.close()
    return out


def divergence_sweep(mw, synth_pair, PackedBatch, n=1024, tl=2000):
    """The size classes are chosen from the lengths for ~5 % divergence; what the chooser costs away from that: Gbp/s (align + results, batch
    resident, second align) and the share of pairs that were run twice, at 1 / 5 / 15 / 30 %."""
    out = {}
    eng = mw.Engine(0)
    for div in (0.01, 0.05, 0.15, 0.30):
        pk = PackedBatch([synth_pair(33000 + i, tl, div) for i in range(n)])
        b = eng.upload(pk)
        o = mw.opt_init()
        b.align(o); b.results()
        first_rr = int(eng.stats().n_retries)
        t0 = time.perf_counter()
        b.align(o)
        b.results()
        w = time.perf_counter() - t0
        out[f"{div:g}"] = {"gbps": pk.bases / w / 1e9, "step_ms": w * 1e3, "re_run_frac": int(eng.stats().n_retries) / n, "first_align_re_run_frac": first_rr / n}
        b.free()
        # the same batch DEVICE-RESIDENT (torch tensors wrapped zero-copy — the path `value` is timed on): its classes follow an 8-mer sketch
        # computed on the device when the batch is wrapped (round 5 classified such a batch by length alone: ~every pair of a 15 % batch twice)
        try:
            import torch
            bw = eng.wrap_packed(pk, torch.device("cuda", 0))
            bw.align(o); bw.results()
            first_rr = int(eng.stats().n_retries)
            t0 = time.perf_counter()
            bw.align(o)
            bw.results()
            w = time.perf_counter() - t0
            out[f"{div:g}"]["wrapped"] = {"gbps": pk.bases / w / 1e9, "step_ms": w * 1e3, "re_run_frac": int(eng.stats().n_retries) / n, "first_align_re_run_frac": first_rr / n}
            bw.free()
        except Exception as e:
            out[f"{div:g}"]["wrapped"] = {"error": repr(e)}
    eng.close()
    return out


def long_batches(mw, synth_pair, PackedBatch, n=1250, tl=50000, div=0.03):
    """One GPU's share of BASELINE configs[4] (10 000 x 50 kb @ 3 % over 8 GPUs = 1250 pairs), score-only, on the default kernel choice
    (the packed band kernel's 1024-thread span geometry; round 3-4: the generic kernel with 16-bit ring rows): one warm-up align, one timed."""
    pairs = [synth_pair(60000 + i, tl, div) for i in range(n)]
    pk = PackedBatch(pairs)
    eng = mw.Engine(0)
    b = eng.upload(pk)
    o = mw.opt_init()
    b.align(o); b.results()
    t0 = time.perf_counter()
    b.align(o)
    s, it, _ = b.results()
    wall = time.perf_counter() - t0
    st = eng.stats()
    cells = int(it.sum())
    ks = st.kernel_ms * 1e-3
    kb = 8 if (st.kernel_kind == 2 and st.packed) else 16 if st.packed == 16 else 32   # packed band kernel: only the 16-bit H rows cross HBM (three loads + one store)
    rec = {"workload": f"{n} x {tl} bp @ {div:g}, score-only (one GPU's share of configs[4])", "kernel_ms": st.kernel_ms, "wall_ms": wall * 1e3,
           "gbp_s": pk.bases / wall / 1e9, "gcells_per_s": cells / wall / 1e9, "cells": cells, "n_retries": int(st.n_retries), "mean_s": float(s.mean()),
           "kernel_kind": int(st.kernel_kind), "block": int(st.block), "ring_bits": 16 if (st.packed == 16 or st.kernel_kind == 2) else 32,
           "kernel": KERNEL_NAMES.get(3 if (st.kernel_kind == 2 and st.packed) else st.kernel_kind, "?"), "kernel_bytes_per_cell": kb,
           "roofline": counter_roofline(ks, TRAFFIC.get(f"{n}x{tl}@{div:g}s", {}), float(kb) * cells, 48.0 * cells, "48 B x cells", ran_symbol(st))}
    b.free()
    eng.close()
    return rec


LONG_SPECS = (("c4_like_150kb", 2001, 150000, 0.035, 0, 0,
               (("score", {}, "c4-score"), ("cigar_highmem", {"flag": 1}, "c4-cigar"), ("cigar_lowmem_p5000", {"flag": 1, "step": 5000}, "c4-lowmem"))),
              ("mhc_like_5Mb", 2002, 5000000, 0.008, 3, 15000,
               (("cigar_lowmem_p5000", {"flag": 1, "step": 5000}, "mhc-lowmem"), ("score", {}, "mhc-score"))))


def profile_matches_tree(prof: dict, ran_symbol: str | None) -> dict:
    """Does the counter profile describe the code that was just timed?  traffic.json entries carry the symbols of the kernels they were
    collected over and a fingerprint of those kernels' source file (profiles/summarize.py records it on the GPU box, profiles/make_traffic.py
    stores it): compared here with the tree bench.py runs from and with the kernel the library says it launched.  `frac_stale` true = the
    traffic figure (and so `frac`) belongs to other code — re-run profiles/r06_profiles.sh + r06_collect.sh."""
    if not prof.get("hbm_bytes_per_launch"):
        return {}
    try:
        from miniwfa_amd.build import kernel_fingerprint
        kerns = prof.get("kernels") or []
        now = kernel_fingerprint(kerns[0]) if kerns else None
        same_src = bool(kerns) and prof.get("kernel_fingerprint") is not None and now == prof.get("kernel_fingerprint")
        same_kernel = None if ran_symbol is None else any(k.replace(" ", "").startswith(ran_symbol.replace(" ", "")) for k in kerns)
        return {"frac_stale": not (same_src and same_kernel is not False),
                "traffic_check": {"profiled_kernels": kerns, "timed_kernel": ran_symbol, "kernel_matches": same_kernel,
                                  "profiled_source_fingerprint": prof.get("kernel_fingerprint"), "tree_source_fingerprint": now,
                                  "source_matches": same_src, "collected_at_commit": prof.get("collected_at_commit")}}
    except Exception as e:  # pragma: no cover
        return {"frac_stale": True, "traffic_check": {"error": repr(e)}}


def ran_symbol(st) -> str | None:
    """Start of the symbol of the kernel the library launched last, from its statistics."""
    if st.kernel_kind == 2 and st.packed and st.block >= 64:
        return f"wfa_band2_kernel<{st.block},"
    if st.kernel_kind == 1:
        return "wfa_sys"
    if st.kernel_kind == 0:
        return "wfa_batch_kernel<"
    return None


def counter_roofline(kernel_s: float, prof: dict, own_bytes: float, nominal_bytes: float, nominal_label: str, ran: str | None = None) -> dict:
    """THE roofline block of this line, one definition for every kernel: achieved = HBM bytes per launch measured by the PMC counters
    (profiles/traffic.json; 2048 B x FETCH_SIZE + 1024 B x WRITE_SIZE, the units calibrated by profiles/micro/fetch_calib.hip) / the kernel
    time measured live with HIP events; frac = achieved / 8 TB/s.  Without a counter profile of the workload (non-default arguments) the
    kernel's own bytes stand in and `frac_source` says so.  The other figures are named side fields, never `frac`."""
    ks = max(kernel_s, 1e-12)
    traffic = prof.get("hbm_bytes_per_launch")
    used = traffic if traffic else own_bytes
    rf = {"bound": "hbm", "achieved": used / ks / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": used / ks / 1e9 / HBM_PEAK_GBS,
          "traffic": traffic,
          "frac_source": ("PMC counter traffic per launch (profiles/traffic.json) / live HIP-event kernel time / 8 TB/s" if traffic else
                          "no counter profile for this workload: the bytes the kernel itself must move / live kernel time / 8 TB/s"),
          "traffic_source": prof.get("source"),
          "kernel_s": ks,
          "own_bytes": own_bytes, "own_bytes_frac": own_bytes / ks / 1e9 / HBM_PEAK_GBS,
          "nominal_48B": {"what": nominal_label, "bytes": nominal_bytes, "gbs": nominal_bytes / ks / 1e9, "frac": nominal_bytes / ks / 1e9 / HBM_PEAK_GBS,
                          "note": "SURVEY 8(d)'s byte model of the REFERENCE's loops; a kernel that keeps E1/F1/E2/F2 on chip and H in 16 bits moves a fraction "
                                  "of it, so this figure can exceed 1 — it is not a roofline fraction"}}
    if prof.get("valu_insts_per_launch"):
        vi = prof["valu_insts_per_launch"]
        rf["valu_issue_frac"] = vi / ks / 1e9 / VALU_PEAK_GINST
        ai = vi + (prof.get("salu_insts_per_launch") or 0) + (prof.get("lds_insts_per_launch") or 0) + (prof.get("vmem_insts_per_launch") or 0)
        rf["all_issue_frac"] = ai / ks / 1e9 / ISSUE_PEAK_GINST
        rf["issue_source"] = prof.get("valu_source")
    if prof.get("wait_any_over_wave_cycles") is not None:
        rf["wait_any_over_wave_cycles"] = prof["wait_any_over_wave_cycles"]
    rf.update(profile_matches_tree(prof, ran))
    return rf


class BackgroundReference:
    """The compiled reference (oracle/_ref) on ONE host thread beside the GPU work of this run — the same-run CPU baseline of the 5 Mb
    pair (configs[3]; about four minutes of one core).  ctypes drops the GIL inside the call."""

    def __init__(self, t, q, kw):
        from oracle.pyoracle import Reference, make_opt
        self.ref = Reference(arena=True)
        self.t, self.q, self.opt = t, q, make_opt(**kw)
        self.result, self.seconds, self.error = None, None, None
        self.th = threading.Thread(target=self._run, daemon=True)
        self.th.start()

    def _run(self):
        try:
            t0 = time.perf_counter()
            self.result = self.ref.align(self.t, self.q, self.opt)
            self.seconds = time.perf_counter() - t0
        except Exception as e:  # pragma: no cover
            self.error = repr(e)

    def join(self):
        self.th.join()
        return self


def read_first_fasta(path: str) -> bytes:
    """First record of a FASTA / FASTQ file (plain or .gz), sequence bytes as they stand in the file (the reference compares bytes verbatim, main.c:67-72)."""
    import gzip
    op = gzip.open if path.endswith(".gz") else open
    seq, started = [], False
    with op(path, "rb") as f:
        first = f.readline()
        fastq = first.startswith(b"@")
        for line in f:
            if line.startswith(b">") or (fastq and line.startswith(b"+")):
                break
            seq.append(line.strip())
    return b"".join(seq)


# the reference's own evaluation pairs (Zenodo record 6056061, README.md:82-88,146) and the penalties its README publishes for them with the DEFAULT costs
REAL_PAIRS = {"c4_like_150kb": ("--c4", 26917), "mhc_like_5Mb": ("--mhc", 229868)}


def long_pairs(mw, synth_pair, PackedBatch, cpu: bool, mhc_cpu: bool, real=None):
    """BASELINE configs[1] and configs[3] (stand-ins): one pair on the whole device, each mode on a fresh
    engine so that `peak_device_bytes` is that mode's own need; the compiled reference timed beside it in the same run:
    inline for the 150 kb pair (seconds), on a host thread that runs while the GPU legs do for the 5 Mb pair (minutes)."""
    lp = {}
    gold = {}
    try:
        for line in open(os.path.join(ROOT, "tests", "golden", "long_pairs.jsonl")):
            v = json.loads(line)
            gold[v["id"]] = v
    except Exception:
        pass
    ref = None
    if cpu:
        try:
            from oracle.pyoracle import Reference, make_opt
            ref = Reference(arena=True) if Reference.available() else None
        except Exception:
            ref = None
    seqs = {name: synth_pair(seed, tl_, p_, nl, lm) for name, seed, tl_, p_, nl, lm, _ in LONG_SPECS}
    real = real or {}
    for name, files in real.items():   # the real pair instead of its synthetic stand-in, the moment the files are there
        seqs[name] = (read_first_fasta(files[0]), read_first_fasta(files[1]))
    bg = None
    if ref is not None and mhc_cpu:   # configs[3]'s CPU baseline: started before the GPU legs, joined behind them
        t_, q_ = seqs["mhc_like_5Mb"]
        bg = BackgroundReference(t_, q_, {"flag": 1, "step": 5000})
    for name, seed, tl_, p_, nl, lm, modes in LONG_SPECS:
        t_, q_ = seqs[name]
        for label, kw, gid in modes:
            eng = mw.Engine(0)
            bb = eng.upload(PackedBatch([(t_, q_)]))
            o_ = mw.opt_init(**kw)
            bb.align(o_)
            bb.results()               # first call also sizes the workspace
            t0 = time.perf_counter()
            bb.align(o_)
            s_, it_, nc_ = bb.results()
            wall = time.perf_counter() - t0
            st_ = eng.stats()
            rec = {"s": int(s_[0]), "n_iter": int(it_[0]), "kernel_s": st_.kernel_ms * 1e-3, "wall_s": wall, "cells_pass1": int(st_.cells_pass1),
                   "gbp_s": (len(t_) + len(q_)) / wall / 1e9, "peak_device_bytes": int(st_.dev_bytes_peak), "n_retries": int(st_.n_retries),
                   "lowmem_two_pass": int(st_.lowmem_two_pass)}
            # SURVEY 8(d) bytes: 48 per cell score-only, 49 with traceback, 97 in the low-memory first pass (+ 49 per cell of the second)
            if kw.get("step"):
                nominal = 97 * int(st_.cells_pass1) + 49 * int(it_[0])
                own = (16 + 1) * (int(st_.cells_pass1) + int(it_[0]))       # what the kernel itself moves: 32-bit H (three loads, one store) + the traceback byte, both passes
            else:
                nominal = (49 if kw.get("flag") else 48) * int(it_[0])
                own = (16 + (1 if kw.get("flag") else 0)) * int(it_[0])
            ks = max(st_.kernel_ms * 1e-3, 1e-9)
            rec["roofline"] = counter_roofline(ks, TRAFFIC.get(f"{name}:{label}", {}), own, nominal,
                                               "97 B x first-pass cells + 49 B x second-pass cells" if kw.get("step") else ("49 B x cells" if kw.get("flag") else "48 B x cells"), ran_symbol(st_))
            rec["roofline"]["us_per_penalty"] = ks / max(1, int(s_[0])) * 1e6 / (2 if kw.get("step") else 1)
            rec["roofline"]["binding"] = "per-penalty latency of ONE sequential chain of penalties (hand-offs between chunk slots + single-wave issue), not bytes"
            if kw.get("flag"):
                cg = bb.cigar(0, int(nc_[0])).tolist()
                rec["cigar_rescored_ok"] = mw.cigar2score(mw.opt_init(), cg) == (int(s_[0]), len(t_), len(q_))
            if name in real:   # the README's published penalty for this pair under the default costs (README.md:83-86) must come out
                rec["real_data"] = {"files": list(real[name]), "readme_s": REAL_PAIRS[name][1], "s_matches_readme": int(s_[0]) == REAL_PAIRS[name][1]}
            g = None if name in real else gold.get(gid)
            if g:
                rec["matches_reference_golden"] = (int(s_[0]), int(it_[0])) == (g["expect"]["s"], g["expect"]["n_iter"])
                rec["cpu_reference_s_build_container"] = g.get("reference_wall_s")
            if ref is not None and tl_ <= 200000:
                ro = make_opt(**kw)
                t0 = time.perf_counter()
                rs = ref.align(t_, q_, ro)
                rec["cpu_reference_s"] = time.perf_counter() - t0
                rec["cpu_reference_matches"] = (rs[0], rs[1]) == (int(s_[0]), int(it_[0]))
                rec["cpu_reference_how"] = "lh3/miniwfa (oracle/_ref), one host thread, same run, inline"
            if bg is not None and name == "mhc_like_5Mb" and label == "cigar_lowmem_p5000":
                rec["_gpu_answer"] = (int(s_[0]), int(it_[0]), cg)
            lp.setdefault(name, {"tl": len(t_), "ql": len(q_)})[label] = rec
            bb.free()
            eng.close()
    if bg is not None:
        bg.join()
        rec = lp["mhc_like_5Mb"]["cigar_lowmem_p5000"]
        gs, git, gcg = rec.pop("_gpu_answer")
        if bg.error is None and bg.result is not None:
            rec["cpu_reference_s"] = bg.seconds
            rec["cpu_reference_matches"] = (bg.result[0], bg.result[1], bg.result[2]) == (gs, git, gcg)   # s, n_iter and every CIGAR word
            rec["cpu_reference_how"] = ("lh3/miniwfa (oracle/_ref), one host thread, same run: started before the GPU legs of long_pairs and joined behind "
                                        "them (it shares the host with the other legs' inline reference calls, each on its own core)")
            rec["speedup_vs_cpu_reference_same_run"] = bg.seconds / max(rec["wall_s"], 1e-9)
        else:
            rec["cpu_reference_error"] = bg.error
    return lp


def launcher_argv(n: int, argv, port: int):
    """The command a bare `python bench.py --gpus N` becomes: one process per GPU under torch.distributed.run, rendezvous on 127.0.0.1."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def free_port() -> int:
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", type=int, default=3, choices=(3, 5), help="3: 1024 x 10 kb per GPU (weak); 5: 10 000 x 50 kb in total (strong)")
    ap.add_argument("--pairs", type=int, default=None, help="pairs per GPU (config 3) / in total (config 5)")
    ap.add_argument("--len", type=int, default=None, dest="tl")
    ap.add_argument("--div", type=float, default=None)
    ap.add_argument("--cigar", action="store_true", help="score+CIGAR (high-memory) instead of score-only")
    ap.add_argument("--block", type=int, default=0)
    ap.add_argument("--band-pack", type=int, default=-1, help="band kernel: 1 forces the int16-packed variants where the forced block has both")
    ap.add_argument("--cpu-sample", type=int, default=None, help="pairs in the cpu_baseline sample (0: skip)")
    ap.add_argument("--long-pairs", type=int, default=1, help="also time the single-pair configs (C4-like 150 kb, MHC-like 5 Mb) on rank 0 at N=1")
    ap.add_argument("--mhc-cpu", type=int, default=1, help="1: time the compiled reference on the 5 Mb pair in this run (one host thread beside the GPU legs, ~4 min); 0: skip")
    ap.add_argument("--long-batches", type=int, default=1, help="also time one GPU's share of configs[4] (1250 x 50 kb, score-only) on rank 0 at N=1")
    ap.add_argument("--extras", type=int, default=1, help="0: skip end_to_end / call latency / long pairs (profiling runs)")
    ap.add_argument("--c4", nargs=2, metavar=("TARGET.fa", "QUERY.fa"), default=None, help="the reference's NA19240 C4A / C4B pair (Zenodo 6056061) instead of the synthetic 150 kb stand-in: s must be 26917 (README.md:83)")
    ap.add_argument("--mhc", nargs=2, metavar=("TARGET.fa", "QUERY.fa"), default=None, help="the reference's GRCh38 / CHM13 MHC pair instead of the synthetic 5 Mb stand-in: s must be 229868 (README.md:86)")
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--seeds", type=int, default=4, help="batches of the headline shape rotated through the timed steps (config 3): value is the mean over them")
    args = ap.parse_args(argv)
    args.dry = os.environ.get("MWF_BENCH_BACKEND", "").lower() == "gloo"
    strong = args.config == 5
    if args.dry:     # the plumbing test: a few short pairs per rank
        defaults = dict(pairs=40, tl=600, div=0.03, seed=60000, steps=2, warmup=1, cpu_sample=0) if strong else \
                   dict(pairs=16, tl=300, div=0.05, seed=50000, steps=2, warmup=1, cpu_sample=0)
    elif strong:
        defaults = dict(pairs=10000, tl=50000, div=0.03, seed=60000, steps=2, warmup=1, cpu_sample=32)
    else:
        defaults = dict(pairs=1024, tl=10000, div=0.05, seed=50000, steps=10, warmup=2, cpu_sample=1024)
    for k, v in defaults.items():
        if getattr(args, k) is None:
            setattr(args, k, v)
    return args


def dry_run(args, rank, world):
    """MWF_BENCH_BACKEND=gloo: the multi-rank plumbing of this file on CPU — deal, per-rank step, ONE all_gather of the records, barrier +
    max-over-ranks timing, the JSON line — with the oracle standing in for the GPU on a tiny workload.  Not a measurement."""
    import torch
    import torch.distributed as dist
    from miniwfa_amd.synth import synth_pair
    from miniwfa_amd.shard import gather_records, deal_pairs
    from oracle.pyoracle import Oracle, make_opt
    strong = args.config == 5
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    if strong:
        n_total = args.pairs
        deal = deal_pairs([2 * args.tl] * n_total, world)
    else:
        n_total = args.pairs * world
        deal = [np.arange(r * args.pairs, (r + 1) * args.pairs) for r in range(world)]
    pairs = [synth_pair(args.seed + int(i), args.tl, args.div) for i in deal[rank]]
    orc, o = Oracle(), make_opt(flag=1 if args.cigar else 0)
    bases = sum(len(t) + len(q) for t, q in pairs)
    gathered = [None]

    def step():
        res = [orc.align(t, q, o) for t, q in pairs]
        s_loc = torch.tensor([r[0] for r in res], dtype=torch.int32)
        it_loc = torch.tensor([r[1] for r in res], dtype=torch.int64)
        if world > 1:
            gathered[0] = gather_records(dist, s_loc, it_loc, n_total, deal=deal)
        else:
            gathered[0] = (s_loc, it_loc)

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    total_bases = bases
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        tot = torch.tensor([bases], dtype=torch.int64)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        total_bases = int(tot.item())
    ok = True
    if rank == 0:   # every pair's record arrived, in global order
        s_all, it_all = gathered[0]
        for i in range(n_total):
            es, eit, _ = orc.align(*synth_pair(args.seed + i, args.tl, args.div), o)
            ok = ok and (int(s_all[i]), int(it_all[i])) == (es, eit)
        print(json.dumps({
            "metric": "aligned Gbp/s (q+t)", "value": total_bases * args.steps / elapsed / 1e9, "unit": "Gbp/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "int32", "data": "synthetic", "dry_run": True,
            "config": {"workload": f"DRY RUN on CPU (MWF_BENCH_BACKEND=gloo): {n_total} pairs x {args.tl} bp dealt over {world} rank(s), the oracle standing in "
                                   "for the GPU — exercises the launcher, the deal and the one all_gather of the records; NOT a measurement",
                       "pairs_total": n_total, "pairs_this_rank": len(pairs)},
            "roofline": None, "cpu_baseline": None, "gathered_records_match_oracle": ok}))
    if world > 1:
        dist.destroy_process_group()
    if not ok:
        raise SystemExit("dry run: gathered records differ from the oracle's")


def main():
    args = parse_args()
    strong = args.config == 5
    # ---- N > 1 without a launcher: become `python -m torch.distributed.run --nproc-per-node N ... bench.py <same arguments>`
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        cmd = launcher_argv(args.gpus, sys.argv[1:], free_port())
        sys.stdout.flush()
        os.execv(cmd[0], cmd)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's --nproc-per-node and --gpus must agree")
    if args.dry:
        return dry_run(args, rank, world)

    import torch
    import miniwfa_amd as mw
    from miniwfa_amd.synth import synth_pair, PackedBatch
    from miniwfa_amd.shard import gather_records, deal_pairs

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback (MWF_BENCH_BACKEND=gloo dry-runs the multi-rank plumbing on CPU)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)  # "nccl" is RCCL on ROCm

    # ---- synthetic batch of this rank, resident in HBM before anything is timed
    if strong:   # fixed total work: the pairs are dealt by work (uniform lengths here: an even deal), every rank generates its own
        n_total = args.pairs
        deal = deal_pairs([2 * args.tl] * n_total, world)
        my_ids = deal[rank].tolist()
    else:        # fixed work per GPU: contiguous seeds per rank
        n_total = args.pairs * world
        deal = [np.arange(r * args.pairs, (r + 1) * args.pairs) for r in range(world)]
        my_ids = deal[rank].tolist()
    # The weak-scaled headline rotates FOUR batches of the same shape through the timed steps (seed, seed + 10000, ...): three of four seeds of this
    # shape hold a pair that outgrows the three-slot 512-thread geometry (round 4 timed the one that does not) — `value` is the mean over them.
    n_seeds = 1 if strong else max(1, args.seeds)
    stream = torch.cuda.current_stream(dev)
    eng = mw.Engine(local_rank, stream.cuda_stream)
    rot = []
    for k in range(n_seeds):
        pk_k = PackedBatch([synth_pair(args.seed + 10000 * k + i, args.tl, args.div) for i in my_ids])
        t_seqs = torch.from_numpy(pk_k.seqs.copy()).to(dev)
        t_toff, t_qoff = torch.from_numpy(pk_k.t_off).to(dev), torch.from_numpy(pk_k.q_off).to(dev)
        t_tl, t_ql = torch.from_numpy(pk_k.tl).to(dev), torch.from_numpy(pk_k.ql).to(dev)
        rot.append((pk_k, (t_seqs, t_toff, t_qoff, t_tl, t_ql)))
    pk = rot[0][0]
    if args.block:
        eng.set("block", args.block)
    if args.band_pack >= 0:
        eng.set("band_pack", args.band_pack)
    batches = []
    for pk_k, (t_seqs, t_toff, t_qoff, t_tl, t_ql) in rot:
        bk = eng.wrap(pk_k.n, t_seqs.data_ptr(), pk_k.total, t_toff.data_ptr(), t_tl.data_ptr(), t_qoff.data_ptr(), t_ql.data_ptr(),
                      pk_k.tl, pk_k.ql, keep=(t_seqs, t_toff, t_qoff, t_tl, t_ql))
        batches.append((bk, torch.as_tensor(_DevPtr(bk.dev_scores_ptr(), pk_k.n, "<i4"), device=dev), torch.as_tensor(_DevPtr(bk.dev_iters_ptr(), pk_k.n, "<i8"), device=dev)))
    batch = batches[0][0]
    opt = mw.opt_init(flag=mw.MWF_F_CIGAR if args.cigar else 0)

    kernel_ms, retries, step_no = [], 0, 0
    per_seed_kernel = [[] for _ in batches]

    def step(record: bool):
        nonlocal retries, step_no
        k = step_no % len(batches)
        step_no += 1
        bk, d_s, d_it = batches[k]
        bk.align(opt)                          # kernels enqueued on torch's current stream
        res = bk.results()                     # (s, n_iter) records on the host; re-runs of pairs that did not fit happen here
        retries += eng.stats().n_retries
        if world > 1:                          # the result gather of the multi-GPU job: one RCCL all_gather over xGMI
            gather_records(dist, d_s, d_it, n_total, device=dev, deal=deal)
        if record:
            kernel_ms.append(eng.stats().kernel_ms)
            per_seed_kernel[k].append(eng.stats().kernel_ms)
        return res

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(max(args.warmup, 2 * len(batches)) if len(batches) > 1 else args.warmup):   # (every batch of the rotation has been aligned before the clock starts)
        step(False)
    fence()
    retries, step_no = 0, 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(False)
    fence()
    elapsed = time.perf_counter() - t0
    timed_retries = retries
    timed_bases = sum(rot[i % len(rot)][0].bases for i in range(args.steps))
    # kernel-only timing (HIP events recorded by the library on the launch stream), outside the wall-clock region
    step_no = 0
    cells_seed = [0] * len(batches)
    s = n_iter = None
    for i in range(max(3, min(args.steps, 10)) // len(batches) * len(batches) or len(batches)):
        s_i, n_iter_i, _ = step(True)
        cells_seed[i % len(batches)] = int(n_iter_i.sum())
        assert (s_i >= 0).all(), "some pairs did not finish"
        if i % len(batches) == 0:
            s, n_iter = s_i.copy(), n_iter_i.copy()   # the first batch of the rotation: what cpu_baseline compares with
    cells = int(round(sum(cells_seed) / len(cells_seed)))   # per launch, mean over the rotation
    st = eng.stats()

    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        tot = torch.tensor([timed_bases, cells, timed_retries], dtype=torch.int64, device=dev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        total_timed_bases, total_cells, timed_retries = int(tot[0].item()), int(tot[1].item()), int(tot[2].item())
    else:
        total_timed_bases, total_cells = timed_bases, cells

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    k_ms = float(np.mean(kernel_ms))
    bytes_per_cell = 49 if args.cigar else ALGO_BYTES_PER_CELL
    mode = "score+CIGAR high-mem" if args.cigar else "score-only"
    if strong:
        workload = (f"{n_total} pairs in total x {args.tl} bp, {args.div:g} divergence, {mode} mwf_wfa_exact, default penalties, dealt over "
                    f"{world} GPU(s) (BASELINE configs[4])")
    else:
        workload = (f"{args.pairs} pairs/GPU x {args.tl} bp, {args.div:g} divergence, {mode} mwf_wfa_exact, default penalties (BASELINE configs[2])")
    # What the kernel itself must move per cell (the floor of ITS traffic): the band kernels keep E1/F1/E2/F2 in registers,
    # so only H crosses HBM (three loads + one store per cell: 16 bytes, 8 with the packed kernel's 16-bit rows), +1 traceback byte;
    # the generic kernel with E2/F2 in LDS moves 32 of the 48, 16 with its 16-bit ring rows.
    if st.kernel_kind == 2:
        kb = (8 if st.packed else 16) + (1 if args.cigar else 0)
    elif st.kernel_kind == 0:
        kb = (16 if st.packed == 16 else 32) + (1 if args.cigar else 0)   # 16-bit ring rows: H (three loads, one store) + E1/F1 (load + store each) at 2 bytes
    else:
        kb = 16 + (1 if args.cigar else 0)
    key = f"{pk.n}x{args.tl}@{args.div:g}{'c' if args.cigar else 's'}"
    rf = counter_roofline(k_ms * 1e-3, TRAFFIC.get(key, {}), float(kb) * cells, float(bytes_per_cell) * cells, f"{bytes_per_cell} B x cells", ran_symbol(st))
    rf.update({"kernel_bytes_per_cell": kb, "cells_per_launch": cells, "kernel_ms": k_ms,
               "note": "frac = counter traffic / kernel time / 8 TB/s (the same definition in long_pairs.*.roofline and long_batches.roofline). Nothing physical binds "
                       "this kernel: what is left is per-penalty synchronisation (wait_any_over_wave_cycles) and single-wave issue latency (DESIGN.md section 4)."})
    out = {
        "metric": "aligned Gbp/s (q+t)",
        "value": total_timed_bases / elapsed / 1e9,
        "unit": "Gbp/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        # Which reading of BASELINE's metric `value` is: the measurement contract's (inputs already resident in HBM when the timed region
        # starts; results on the host when it ends).  SURVEY 8(d)'s reading — host buffers in to host results out, H2D + kernels + D2H — is
        # the top-level `end_to_end` block of this line (same batches, same run), never `value`.
        "value_definition": "resident: sequences in HBM before the clock starts, alignment kernels + (s, n_iter) records on the host (+ the RCCL gather at N > 1) inside it; "
                            "SURVEY 8(d)'s host-to-host figure is `end_to_end.gbps`",
        "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
        # the arithmetic the timed kernel computes in (results are bit-exact to the reference's int32 either way)
        "dtype": ("int16x2 packed, range-guarded (bit-exact to the reference's int32)" if (st.kernel_kind == 2 and st.packed) else
                  "int16 ring rows, int32 arithmetic" if (st.kernel_kind == 0 and st.packed == 16) else "int32"),
        "data": "synthetic",
        "config": {
            "workload": workload,
            "pairs_this_gpu": pk.n, "pairs_total": n_total, "target_len": args.tl, "divergence": args.div,
            "bases_this_gpu": pk.bases, "cells_this_gpu": cells, "mean_s": float(s.mean()),
            "step": "alignment kernels on sequences already resident in HBM when the timed region starts (the measurement contract's `value`; the host-buffers-in "
                    "to host-results-out rate of the same batch is `end_to_end_gbps`) + (s, n_iter) records to the host" + (" + RCCL all_gather of the records" if world > 1 else ""),
            "kernel": KERNEL_NAMES.get(3 if (st.kernel_kind == 2 and st.packed) else st.kernel_kind, "?"), "grid": st.grid, "block": st.block,
            "parallelism": f"pairs dealt over {world} GPU(s), no data-path collective, one RCCL all_gather of (s,n_iter)",
        },
        "gcells_per_s": total_cells * args.steps / elapsed / 1e9,
        "seeds": {"n": len(batches), "base_seeds": [args.seed + 10000 * k for k in range(len(batches))],
                  "kernel_ms_per_seed": [float(np.mean(x)) if x else None for x in per_seed_kernel],
                  "note": "the timed steps rotate through these batches: `value`, `ms_per_step` and roofline.kernel_ms are means over them"},
        "kernel_gbps": pk.bases / (k_ms * 1e-3) / 1e9,
        "n_retries": timed_retries,
        "roofline": rf,
    }

    if world == 1 and args.extras:
        out["peak_device_bytes"] = int(st.dev_bytes_peak)
        # ---- host buffers in -> host results out (PCIe both ways), warmed, pooled allocations — reported beside, never as `value`
        try:
            for _ in range(2):
                b2 = eng.upload(pk); b2.align(opt); b2.results(); b2.free()
            reps = max(3, min(args.steps, 10))
            shots, firsts = [], []
            t1 = time.perf_counter()
            for i in range(reps):
                pk2 = rot[i % len(rot)][0]
                ta = time.perf_counter(); b2 = eng.upload(pk2); tb = time.perf_counter(); b2.align(opt); b2.results(); tc = time.perf_counter(); b2.free()
                shots.append((time.perf_counter() - ta) * 1e3), firsts.append((tc - tb) * 1e3)
            out["end_to_end_gbps"] = sum(rot[i % len(rot)][0].bases for i in range(reps)) / (time.perf_counter() - t1) / 1e9
            # one shot = a fresh batch each time: pack + H2D + FIRST align (its plan, the wide class on four chunk slots) + records + free
            out["one_shot"] = {"ms": float(np.mean(shots)), "gbps": pk.bases / float(np.mean(shots)) / 1e6, "first_align_ms": float(np.mean(firsts)),
                               "what": "host buffers in -> host results out of a batch aligned ONCE on a warm engine (SURVEY 8(d)'s reading of the metric)"}
            out["end_to_end"] = {"metric": "aligned Gbp/s (q+t)", "gbps": out["one_shot"]["gbps"], "ms_per_batch": out["one_shot"]["ms"], "unit": "Gbp/s",
                                 "definition": "SURVEY 8(d): host buffers in -> host results out (pack + H2D + first align of a FRESH batch, plan and all, + D2H of the records + free), "
                                               "warm engine, mean over the rotation's batches", "ratio_to_value": out["one_shot"]["gbps"] / max(out["value"], 1e-12)}
        except Exception as e:
            out["end_to_end_gbps"] = repr(e)

    # ---- CPU baseline: same pairs (a bounded sample), host cores of this box
    if world == 1 and args.cpu_sample > 0:
        from oracle.pyoracle import Oracle, Reference, make_opt
        n = min(args.cpu_sample, pk.n)
        cores = host_cores()
        o = make_opt(flag=1 if args.cigar else 0)
        orc = Oracle()
        threads = min(cores, n)
        n1 = min(n, 48)   # the one-thread leg: a bounded sample (~1.5 s of one core)
        sb = int(pk.tl[:n].sum() + pk.ql[:n].sum())
        sb1 = int(pk.tl[:n1].sum() + pk.ql[:n1].sum())

        def timed(fn, arena, label):
            """N threads over the n pairs and ONE thread over the first n1 (both figures), answers compared with the GPU's."""
            cs, cit, sec = orc.batch(pk, o, threads, exact_fn=fn, n=n, arena=arena)   # pthread pool in C, one pair per thread at a time
            c1, i1, sec1 = orc.batch(pk, o, 1, exact_fn=fn, n=n1, arena=arena)
            return {"build": label, "threads": threads, "gbps": sb / sec / 1e9, "gcells_per_s": float(cit.sum()) / sec / 1e9, "wall_s": sec,
                    "one_thread_gbps": sb1 / sec1 / 1e9, "one_thread_pairs": n1, "one_thread_wall_s": sec1,
                    "matches_gpu": bool((cs == s[:n]).all() and (cit == n_iter[:n]).all() and (c1 == s[:n1]).all() and (i1 == n_iter[:n1]).all())}

        builds = {}
        if Reference.available():
            kind = "reference"
            # the README's recommended build and the widest vector build this host can run ("-msse4 and -march=native" — native of the
            # build container would not be portable to this box, so the x86-64-v3 / v4 levels are built there and checked against /proc/cpuinfo here)
            for variant in ("sse4.2", "v3", "v4"):
                if not Reference.variant_usable(variant):
                    continue
                try:
                    ref = Reference(variant=variant)
                    builds[variant] = timed(ref.exact_addr(), ref.arena_addrs(), f"lh3/miniwfa {ref.flags} (oracle/_ref), a private kalloc arena per thread")
                except Exception as e:  # pragma: no cover
                    builds[variant] = {"error": repr(e)}
        else:
            kind = "port"
            builds["port"] = timed(None, None, "oracle/mwf_oracle.c (this repo's C restatement of the path), -O3 -msse4.2")
        good = {k: v for k, v in builds.items() if "gbps" in v}
        best = max(good, key=lambda k: good[k]["gbps"])
        out["cpu_baseline"] = {
            "value": good[best]["gbps"], "unit": "Gbp/s", "cores": threads, "kind": kind,
            "sample": f"first {n} of the {pk.n} pairs, pthread pool of {threads} threads (= usable CPUs: affinity mask capped by the cgroup quota; os.cpu_count() = {os.cpu_count()}), "
                      f"{good[best]['build']}, {good[best]['wall_s']:.3f} s wall; the fastest of the builds listed under `builds` (each also timed on ONE thread over the first {n1} pairs)",
            "gcells_per_s": good[best]["gcells_per_s"],
            "one_thread_gbps": good[best]["one_thread_gbps"],
            "gpu_matches_cpu_on_sample": all(v["matches_gpu"] for v in good.values()),
            "builds": builds,
        }
        if kind != "reference":
            out["cpu_baseline"]["fallback"] = ("oracle/_ref/libmwf_ref.so is absent (a fresh clone: the compiled reference is a git-ignored binary that only a build container "
                                               "with /root/reference produces) — this repo's own C restatement was timed instead, hence kind = 'port'")
    for bk, _, _ in batches:
        bk.free()
    eng.close()
    if world == 1 and args.extras:
        try:
            out["call_latency_us"] = call_latency(mw, synth_pair)
        except Exception as e:  # never lose the headline line over the extras
            out["call_latency_us"] = {"error": repr(e)}
        try:
            out["chain_mode"] = chain_mode(mw, synth_pair)
        except Exception as e:
            out["chain_mode"] = {"error": repr(e)}
        try:
            out["short_reads"] = short_reads(mw, synth_pair, PackedBatch)
        except Exception as e:
            out["short_reads"] = {"error": repr(e)}
        try:
            out["divergence_sweep"] = divergence_sweep(mw, synth_pair, PackedBatch)
        except Exception as e:
            out["divergence_sweep"] = {"error": repr(e)}
        if args.long_batches and not strong:
            try:
                out["long_batches"] = long_batches(mw, synth_pair, PackedBatch)
            except Exception as e:
                out["long_batches"] = {"error": repr(e)}
        if args.long_pairs and not strong:
            try:
                out["long_pairs"] = long_pairs(mw, synth_pair, PackedBatch, cpu=args.cpu_sample > 0, mhc_cpu=bool(args.mhc_cpu),
                                               real={k: v for k, v in (("c4_like_150kb", args.c4), ("mhc_like_5Mb", args.mhc)) if v})
            except Exception as e:
                out["long_pairs"] = {"error": repr(e)}
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
