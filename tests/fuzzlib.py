"""The randomised cross-checks of the HIP path — one function per fuzzer, each returning the list of mismatches it found.

Round 5 ran these as builder scripts under profiles/ (where they found two bit-exactness bugs no golden vector saw); they now live
here so that tests/test_gpu_fuzz.py runs them with fixed seeds inside the driver's `-m gpu` suite.  profiles/fuzz_*.py remain as
command-line wrappers for more seeds.  Checker = oracle/mwf_oracle.c (pinned to the compiled reference by the golden fixtures), or one
kernel form against another where both are on the device.  s, n_iter and CIGAR words are compared for equality: integer work, no tolerance.
"""
from __future__ import annotations

import os
import time

import numpy as np

import miniwfa_amd as mw
from miniwfa_amd.synth import PackedBatch, fuzz_pairs, synth_pair, skewed_pairs
from oracle.pyoracle import Oracle, make_opt

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
ORACLE_THREADS = max(1, min(8, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 4))


def oracle_many(orc, pairs, o):
    """[(s, n_iter, cigar)] from the oracle on a few host threads (ctypes drops the GIL inside the call)."""
    return orc.align_many(pairs, o, threads=ORACLE_THREADS)[0]


def run_engine(pk, opt_kw, tunables=(), want_cigar=None):
    """One align of a packed batch on a fresh engine with the given tunables: (s, n_iter, cigars | None, stats)."""
    eng = mw.Engine(0)
    try:
        for k, v in tunables:
            eng.set(k, v)
        o = mw.opt_init(**opt_kw)
        b = eng.upload(pk)
        b.align(o)
        s, it, nc = b.results()
        want_cigar = bool(o.flag & 1) if want_cigar is None else want_cigar
        cig = [b.cigar(i, int(nc[i])).tolist() for i in range(pk.n)] if want_cigar else None
        st = eng.stats()
        out = (np.array(s).copy(), np.array(it).copy(), cig, st)
        b.free()
        return out
    finally:
        eng.close()


def compare(got, exp, label, pairs, bad, log, check_cigar=True):
    """Append (label, pair index, lengths, got, expected) for every pair whose (s, n_iter[, CIGAR]) differs from the checker's."""
    s, it, cig, st = got
    n_bad = 0
    for i, (es, eit, ecig) in enumerate(exp):
        ok = (int(s[i]), int(it[i])) == (es, eit) and (not check_cigar or cig is None or cig[i] == (ecig or []))
        if not ok:
            n_bad += 1
            bad.append((label, i, len(pairs[i][0]), len(pairs[i][1]), (int(s[i]), int(it[i])), (es, eit)))
    if log:
        print(f"   {label}: {len(exp)} pairs, mismatches {n_bad}, re-runs {st.n_retries}", flush=True)
    return n_bad


# ---- profiles/fuzz_fold.py: folded against unfolded form of the packed band kernel, and both against the oracle -------------------------
def fuzz_fold(seed=1, n=400, log=False, long_sets=True, penalty_sets=None):
    """Shapes that move the window's start up (length-skewed and unrelated pairs, long gaps), fuzz pairs, 10-20 kb pairs; penalty sets
    with o1 == x (the fold's condition, mwf_band2.hip FOLD; reference recurrence miniwfa.c:267-278)."""
    orc = Oracle()
    sets = {
        "skewed 200-3000": skewed_pairs(seed, n, 200, 3000),
        "fuzz": fuzz_pairs(seed, n, 3000),
        "skewed 4-9 kb": skewed_pairs(seed + 7, max(n // 8, 16), 4000, 9000),
    }
    if long_sets:
        sets["10 kb @ 5 %"] = [synth_pair(seed * 1000 + i, 10000, 0.05) for i in range(64)]
        sets["16 kb @ 4 %"] = [synth_pair(seed * 1000 + 500 + i, 16000, 0.04) for i in range(32)]
    penalty_sets = penalty_sets or (dict(), dict(x=2, o1=2, e1=2, o2=12, e2=1), dict(x=6, o1=6, e1=1, o2=30, e2=1), dict(x=3, o1=3, e1=2, o2=9, e2=2))
    bad = []
    for name, pairs in sets.items():
        pk = PackedBatch(pairs)
        for kw in penalty_sets:
            if max(len(t) + len(q) for t, q in pairs) < 7000:  # the oracle finishes these in seconds: score-only and CIGAR under the default routing
                exp = oracle_many(orc, pairs, make_opt(flag=1, **kw))
                for flag in (0, 1):
                    compare(run_engine(pk, dict(flag=flag, **kw)), exp, f"fold seed {seed} {name} {kw} flag {flag} default routing", pairs, bad, log)
            for block in (0, 512, 1024):
                res = {}
                for fold in (1, 0):
                    tun = [("band_fold", fold)]
                    if block == 1024:
                        tun.append(("band_span", 2))   # the span geometry: 1024 threads x 5 slots
                    elif block:
                        tun += [("force_kind", 2), ("block", block), ("band_pack", 1)]
                    res[fold] = run_engine(pk, dict(**kw), tun)
                diff = np.nonzero((res[0][0] != res[1][0]) | (res[0][1] != res[1][1]))[0]
                for i in diff:
                    bad.append((f"fold seed {seed} {name} {kw} block {block}", int(i), len(pairs[i][0]), len(pairs[i][1]),
                                (int(res[1][0][i]), int(res[1][1][i])), (int(res[0][0][i]), int(res[0][1][i]))))
                if log:
                    print(f"   fold seed {seed} {name} {kw} block {block or 'auto'}: {pk.n} pairs, folded != unfolded on {len(diff)}", flush=True)
    return bad


# ---- profiles/fuzz_default_routing.py: mixed batches under the DEFAULT routing against the oracle -------------------------------------
def fuzz_default_routing(seed=1, scale=1.0, log=False, penalty_sets=None):
    """What a caller of mwf_wfa_batch gets: thousands of read-length pairs at 0-25 % (lane kernel, device-side re-runs, mid kernel), medium
    pairs, length-skewed and unrelated pairs, a few 5-12 kb pairs; folded and unfolded penalty sets; score and CIGAR."""
    rng = np.random.default_rng(seed)
    orc = Oracle()
    pairs = []
    for i in range(int(1500 * scale)):
        pairs.append(synth_pair(seed * 100000 + i, int(rng.integers(50, 400)), float(rng.choice([0.0, 0.02, 0.05, 0.1, 0.25]))))
    for i in range(int(300 * scale)):
        pairs.append(synth_pair(seed * 100000 + 50000 + i, int(rng.integers(400, 3000)), float(rng.choice([0.01, 0.05, 0.15]))))
    pairs += skewed_pairs(seed, int(60 * scale), 200, 3000)
    pairs += fuzz_pairs(seed, int(60 * scale), 2500)
    for i in range(int(8 * scale)):
        pairs.append(synth_pair(seed * 100000 + 90000 + i, int(rng.integers(5000, 12000)), float(rng.choice([0.03, 0.05, 0.1]))))
    order = rng.permutation(len(pairs))
    pairs = [pairs[i] for i in order]
    pk = PackedBatch(pairs)
    bad = []
    penalty_sets = penalty_sets or (dict(), dict(x=2, o1=2, e1=2, o2=12, e2=1), dict(x=6, o1=2, e1=2, o2=20, e2=1), dict(x=1, o1=0, e1=1, o2=0, e2=1))
    for kw in penalty_sets:
        exp = oracle_many(orc, pairs, make_opt(flag=1, **kw))
        for flag in (0, 1):
            compare(run_engine(pk, dict(flag=flag, **kw)), exp, f"routing seed {seed} {kw} flag {flag}", pairs, bad, log)
    return bad


# ---- profiles/fuzz_all_kernels_oracle.py: the fuzz pairs through the OTHER kernels -----------------------------------------------------
ALL_KERNEL_CONFIGS = [
    ("generic, one column per lane", dict(force_kind=0, scalar_generic=1)),
    ("generic, four columns per lane", dict(force_kind=0)),
    ("generic, 16-bit ring rows (packed recurrence)", dict(force_kind=0, ring16=2)),
    ("band, unpacked 256", dict(force_kind=2, block=256, band_pack=0)),
    ("band, unpacked 768", dict(force_kind=2, block=768, band_pack=0)),
    ("whole-device", dict(force_kind=1)),
]


def fuzz_all_kernels(seed=1, n_pairs=120, log=False, wd_pairs=12, modes=None):
    """Generic kernel (one / four columns per lane, 32- and 16-bit ring rows, low-memory two-pass mode), forced band geometries and the
    whole-device (systolic) kernel — score, CIGAR and low-memory (miniwfa.c:551-601) — against the oracle."""
    pairs = fuzz_pairs(seed, n_pairs, 3000)
    # two longer pairs (one of them unrelated): a batch whose longest pair exceeds 8 kb takes the generic kernel's wide form (E2/F2 in LDS), the one with 16-bit rows
    rng = np.random.default_rng(seed + 1000)
    pairs.append((ACGT[rng.integers(0, 4, 5200)].tobytes(), ACGT[rng.integers(0, 4, 3900)].tobytes()))
    t = ACGT[rng.integers(0, 4, 6000)]
    q = np.delete(t.copy(), rng.integers(0, 6000, 250))
    q[rng.integers(0, len(q), 200)] = ACGT[rng.integers(0, 4, 200)]
    pairs.append((t.tobytes(), q.tobytes()))
    orc = Oracle()
    bad = []
    for kw in (modes or (dict(), dict(flag=1), dict(flag=1, step=97))):
        exp = oracle_many(orc, pairs, make_opt(**kw))
        for name, sets in ALL_KERNEL_CONFIGS:
            if name == "whole-device":
                sub = [i for i in range(len(pairs)) if len(pairs[i][0]) + len(pairs[i][1]) > 600][:wd_pairs]   # (one launch per pair or group: a few)
            else:
                sub = list(range(len(pairs)))
            sub_pairs = [pairs[i] for i in sub]
            try:
                got = run_engine(PackedBatch(sub_pairs), kw, list(sets.items()))
            except Exception as ex:  # a tunable this build does not accept
                if log:
                    print("   (skipped:", name, ex, ")")
                continue
            compare(got, [exp[i] for i in sub], f"kernels seed {seed} {kw} {name}", sub_pairs, bad, log)
    return bad


# ---- profiles/fuzz_band2_oracle.py: the packed band kernel's geometries against the oracle ------------------------------------------------
def fuzz_band2(seed=1, n_pairs=240, log=False, blocks=(0, 64, 512, 768), modes=None):
    pairs = fuzz_pairs(seed, n_pairs)
    pk = PackedBatch(pairs)
    orc = Oracle()
    bad = []
    for kw in (modes or (dict(), dict(flag=1), dict(flag=1, o2=4, e2=2), dict(flag=1, x=1, o1=0, e1=1, o2=0, e2=1), dict(flag=0, x=6, o1=2, e1=2, o2=20, e2=1))):
        exp = oracle_many(orc, pairs, make_opt(**kw))
        for block in blocks:
            tun = [("force_kind", 2), ("block", block), ("band_pack", 1)] if block else []
            compare(run_engine(pk, kw, tun), exp, f"band2 seed {seed} {kw} block {block or 'auto'}", pairs, bad, log)
    return bad


# ---- profiles/fuzz_ring16.py: the generic kernel's 16-bit ring rows against its 32-bit rows ---------------------------------------------
def fuzz_ring16(seed=1, n_pairs=40, log=False, modes=None):
    rng = np.random.default_rng(seed)
    pairs = []
    for i in range(n_pairs):
        tl = int(rng.integers(4000, 40000))
        t, q = synth_pair(900000 + 1000 * seed + i, tl, float(rng.choice([0.0, 0.005, 0.02, 0.06, 0.15])))
        if i % 5 == 0:
            q = q[: max(1, len(q) - int(rng.integers(0, 3000)))]          # much shorter query
        if i % 7 == 0:
            t = t[:1000] + t[1000:2000] * 3 + t[2000:]                   # a tandem repeat in the target
        if i % 11 == 0:
            t, q = b"A" * 5000 + t, b"A" * 4990 + q                      # a long homopolymer run
        pairs.append((t, q))
    pk = PackedBatch(pairs)
    bad = []
    for kw in (modes or (dict(), dict(flag=1), dict(flag=1, x=6, o1=2, e1=2, o2=20, e2=1), dict(flag=0, max_s=2000))):
        a = run_engine(pk, kw, [("ring16", 0), ("force_kind", 0)])
        c = run_engine(pk, kw, [("ring16", 2), ("force_kind", 0)])
        for i in range(pk.n):
            if a[0][i] != c[0][i] or a[1][i] != c[1][i] or (a[2] is not None and a[2][i] != c[2][i]):
                bad.append((f"ring16 seed {seed} {kw}", i, len(pairs[i][0]), len(pairs[i][1]), (int(c[0][i]), int(c[1][i])), (int(a[0][i]), int(a[1][i]))))
        if log:
            print(f"   ring16 seed {seed} {kw}: {pk.n} pairs, 16-bit rows (stats.packed {c[3].packed}) vs 32-bit, max s {int(a[0].max())}", flush=True)
    return bad


# ---- profiles/fuzz_seq2.py: the 2-bit sequence copy against the byte-wise copy ---------------------------------------------------------
def fuzz_seq2(seed=1, n_pairs=600, log=False, modes=None):
    rng = np.random.default_rng(seed)

    def rand_seq(n, kind):
        if kind == 0:
            return ACGT[rng.integers(0, 4, n)]
        if kind == 1:
            return np.full(n, ACGT[rng.integers(0, 4)], dtype=np.uint8)                       # homopolymer
        if kind == 2:
            return np.resize(ACGT[rng.integers(0, 4, rng.integers(1, 40))], n)                 # tandem repeat
        return ACGT[rng.choice(4, n, p=[0.85, 0.05, 0.05, 0.05])]                             # low complexity

    def mutate(t, p):
        out = []
        for b in t:
            r = rng.random()
            if r < p / 3:
                continue
            if r < 2 * p / 3:
                out.append(ACGT[rng.integers(0, 4)])
            if r < p:
                out.append(ACGT[rng.integers(0, 4)])
                continue
            out.append(b)
        return np.array(out, dtype=np.uint8)

    pairs = []
    for i in range(n_pairs):
        n = int(rng.choice([0, 1, 15, 16, 17, 31, 32, 33, 63, 64, 65, 255, 256, 257, 1023, 1024, 1025])) if i % 3 == 0 else int(rng.integers(0, 3000))
        t = rand_seq(n, i % 4)
        q = mutate(t, float(rng.choice([0.0, 0.01, 0.05, 0.2]))) if i % 7 else rand_seq(int(rng.integers(0, 3000)), (i + 1) % 4)
        pairs.append((t.tobytes(), q.tobytes()))
    pk = PackedBatch(pairs)
    bad = []
    for kw in (modes or (dict(), dict(flag=1), dict(flag=1, o2=4, e2=2), dict(flag=1, x=1, o1=0, e1=1, o2=0, e2=1), dict(flag=0, x=6, o1=2, e1=2, o2=20, e2=1), dict(flag=1, max_s=300))):
        a = run_engine(pk, kw, [("seq2bit", 0)])
        c = run_engine(pk, kw, [("seq2bit", 1)])
        for i in range(pk.n):
            if a[0][i] != c[0][i] or a[1][i] != c[1][i] or (a[2] is not None and a[2][i] != c[2][i]):
                bad.append((f"seq2 seed {seed} {kw}", i, len(pairs[i][0]), len(pairs[i][1]), (int(c[0][i]), int(c[1][i])), (int(a[0][i]), int(a[1][i]))))
        if log:
            print(f"   seq2 seed {seed} {kw}: {pk.n} pairs, 2-bit vs bytes", flush=True)
    return bad


# ---- profiles/fuzz_chain.py: chain mode and mwf_wfa_auto against the COMPILED reference ----------------------------------------------
CHAIN_OPTS = [dict(flag=1), dict(flag=0), dict(flag=1, kmer=11, max_occ=3, min_len=20), dict(flag=1, kmer=15, max_occ=1, min_len=40), dict(flag=1, step=200),
              dict(flag=1, kmer=9, max_occ=5, min_len=10), dict(flag=1, x=2, o1=2, e1=2, o2=12, e2=1), dict(flag=1, x=6, o1=2, e1=2, o2=20, e2=1)]


def chain_fuzz_pair(rng):
    """One pair with structural variation (reference miniwfa.c:850-896 chains through it): substitutions at 1-20 %, long insertions,
    deletions, tandem duplications, blocks that do not align, low-complexity stretches."""
    def rnd(n):
        return rng.integers(0, 4, n).astype(np.uint8)
    tl = int(rng.choice([300, 1500, 5000, 12000, 30000]))
    t = rnd(tl)
    if rng.random() < 0.3:  # a low-complexity stretch and a tandem repeat in the target
        a = int(rng.integers(0, tl // 2))
        t[a:a + tl // 10] = t[a]
        unit = rnd(int(rng.integers(2, 40)))
        b = int(rng.integers(tl // 2, tl - 1))
        n = min(tl - b, len(unit) * 30)
        t[b:b + n] = np.resize(unit, n)
    q = t.copy()
    p = float(rng.choice([0.01, 0.04, 0.1, 0.2]))
    flip = rng.random(len(q)) < p
    q[flip] = (q[flip] + rng.integers(1, 4, int(flip.sum()))) & 3
    for _ in range(int(rng.integers(0, 4))):  # structural events
        kind = int(rng.integers(0, 4))
        at = int(rng.integers(0, max(1, len(q) - 1)))
        ln = int(rng.choice([50, 400, 2500, 11000]))
        if kind == 0:
            q = np.concatenate([q[:at], q[at + ln:]])                       # deletion
        elif kind == 1:
            q = np.concatenate([q[:at], rnd(ln), q[at:]])                   # insertion
        elif kind == 2:
            q = np.concatenate([q[:at], q[max(0, at - ln):at], q[at:]])     # tandem duplication
        else:
            q = np.concatenate([q[:at], rnd(ln), q[at + ln:]])              # a block that does not align
    if len(q) == 0:
        q = rnd(10)
    return ACGT[t].tobytes(), ACGT[q].tobytes()


def fuzz_chain(seed=1, n_pairs=40, log=False):
    """Needs oracle/_ref/libmwf_ref.so (the compiled reference travels with the snapshot as a git-ignored binary); the stored chain
    answers of tests/golden/chain_fresh.jsonl cover the case where it did not."""
    from oracle.pyoracle import Reference
    keys = ("flag", "x", "o1", "e1", "o2", "e2", "step", "max_s", "max_iter", "max_occ", "kmer", "min_len")
    rng = np.random.default_rng(seed)
    ref = Reference()
    bad = []
    t_ref = t_gpu = 0.0
    for i in range(n_pairs):
        t, q = chain_fuzz_pair(rng)
        for kw in CHAIN_OPTS:
            o = make_opt(**kw)
            t0 = time.time()
            es, _, ecig = ref.chain(t, q, o)
            t_ref += time.time() - t0
            t0 = time.time()
            s, _, cig = mw.wfa_chain(t, q, mw.opt_init(**{k: int(getattr(o, k)) for k in keys}))
            t_gpu += time.time() - t0
            if not (s == es and (None if cig is None else list(cig)) == ecig):
                bad.append((f"chain seed {seed} {kw}", i, len(t), len(q), (s, -1), (es, -1)))
        if len(t) + len(q) <= 30000:  # mwf_wfa_auto: the exact branch below 1e8 cells, the chain beyond
            es, eit, ecig = ref.auto(t, q, make_opt(flag=1))
            s, it, cig = mw.wfa_auto(t, q, mw.opt_init(flag=1))
            if not (s == es and it == eit and (None if cig is None else list(cig)) == ecig):
                bad.append((f"auto seed {seed}", i, len(t), len(q), (s, it), (es, eit)))
    if log:
        print(f"   chain seed {seed}: {n_pairs} pairs x {len(CHAIN_OPTS)} option sets, mismatches {len(bad)}; reference {t_ref:.1f} s, this library {t_gpu:.1f} s", flush=True)
    return bad


def report(name, bad, seed):
    for b in bad[:8]:
        print("   BAD", *b, flush=True)
    print(name, "FAILED" if bad else "OK", "seed", seed, flush=True)
    return 1 if bad else 0
