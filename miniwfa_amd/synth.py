"""Deterministic synthetic sequence pairs.

The reference ships no generator (its evaluation data lives on Zenodo, README.md:146), so the
benchmark configs are defined on synthetic pairs: a uniform random ACGT target and a query that is
the target mutated per base with probability P — 60 % substitutions (always to a different
base), 20 % insertions, 20 % deletions, indel lengths geometric with continue-probability 0.3.

Everything is driven by a counter-based splitmix64 hash evaluated with numpy uint64 arithmetic,
so the sequences depend only on (seed, tl, P, ...) and not on the numpy version; the golden
fixtures under tests/golden/ store seeds, not sequences.
"""
from __future__ import annotations

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
_GOLD = np.uint64(0x9E3779B97F4A7C15)
_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def _mix(z: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on a uint64 array."""
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _stream(seed: int, stream: int, n: int, start: int = 0) -> np.ndarray:
    """n 64-bit words of stream `stream` for `seed`, counter = start..start+n-1."""
    with np.errstate(over="ignore"):
        key = _mix(np.array([(seed * 0x632BE59BD9B4E019 + stream * 0xD1B54A32D192ED03 + 0x2545F4914F6CDD1D) & 0xFFFFFFFFFFFFFFFF], dtype=np.uint64))[0]
        ctr = np.arange(start, start + n, dtype=np.uint64)
        return _mix(key + (ctr + np.uint64(1)) * _GOLD)


def _unit(x: np.ndarray) -> np.ndarray:
    """uint64 -> float64 in [0,1) using the top 53 bits."""
    return (x >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def _geom_len(x: np.ndarray, cont: float, cap: int) -> np.ndarray:
    """1 + Geometric(continue-prob cont), capped."""
    u = 1.0 - _unit(x)  # (0,1]
    n = 1 + np.floor(np.log(u) / np.log(cont)).astype(np.int64)
    return np.clip(n, 1, cap)


def random_seq(seed: int, n: int) -> bytes:
    """Uniform random ACGT of length n."""
    return _ACGT[(_stream(seed, 0, n) >> np.uint64(60)).astype(np.int64) & 3].tobytes()


def mutate(target: bytes, seed: int, p: float, cont: float = 0.3, cap: int = 255,
           frac_sub: float = 0.6, frac_ins: float = 0.2) -> bytes:
    """Query = target mutated per base with probability p (see module docstring)."""
    t = np.frombuffer(target, dtype=np.uint8)
    n = t.size
    if n == 0:
        return b""
    code = np.zeros(256, dtype=np.int64)
    for i, c in enumerate(b"ACGT"):
        code[c] = i
    tc = code[t]
    ev = _unit(_stream(seed, 1, n)) < p
    kind = _unit(_stream(seed, 2, n))
    is_sub = ev & (kind < frac_sub)
    is_ins = ev & (kind >= frac_sub) & (kind < frac_sub + frac_ins)
    is_del = ev & (kind >= frac_sub + frac_ins)
    glen = _geom_len(_stream(seed, 3, n), cont, cap)
    # deletions: position i starts a deletion of glen[i] target bases; anything inside a deleted
    # run emits nothing and its own event is ignored.  A deletion start that is itself covered by
    # an earlier deletion still extends the run (union of intervals).
    diff = np.zeros(n + 1, dtype=np.int64)
    ds = np.nonzero(is_del)[0]
    np.add.at(diff, ds, 1)
    np.add.at(diff, np.minimum(ds + glen[ds], n), -1)
    deleted = np.cumsum(diff[:n]) > 0
    subbase = (tc + 1 + ((_stream(seed, 4, n) >> np.uint64(40)) % np.uint64(3)).astype(np.int64)) & 3
    first = np.where(is_sub, subbase, tc)
    count = np.where(deleted, 0, 1 + np.where(is_ins, glen, 0))
    total = int(count.sum())
    if total == 0:
        return b""
    src = np.repeat(np.arange(n, dtype=np.int64), count)
    begin = np.cumsum(count) - count
    within = np.arange(total, dtype=np.int64) - begin[src]
    with np.errstate(over="ignore"):
        insb = (_mix(_stream(seed, 5, 1)[0] + (src.astype(np.uint64) * np.uint64(256) + within.astype(np.uint64)) * _GOLD) >> np.uint64(61)).astype(np.int64) & 3
    out = np.where(within == 0, first[src], insb)
    return _ACGT[out].tobytes()


def long_indels(target: bytes, query: bytes, seed: int, n_events: int, max_len: int) -> bytes:
    """Insert/delete a few long blocks in the query (MHC-like structural differences).

    Applied to the query independently of mutate(): event j picks a query position and a length in
    [max_len/8, max_len]; even j deletes the block, odd j inserts random sequence."""
    q = bytearray(query)
    r = _stream(seed, 6, 3 * max(n_events, 1))
    for j in range(n_events):
        if not q:
            break
        pos = int(r[3 * j] % np.uint64(len(q)))
        ln = int(max_len // 8 + int(r[3 * j + 1] % np.uint64(max(1, max_len - max_len // 8))))
        if j % 2 == 0:
            del q[pos:pos + ln]
        else:
            q[pos:pos] = random_seq(int(r[3 * j + 2] & np.uint64(0x7FFFFFFF)), ln)
    return bytes(q)


def synth_pair(seed: int, tl: int, p: float, n_long: int = 0, long_max: int = 0) -> tuple[bytes, bytes]:
    """(target, query) for one synthetic pair; seed = base seed + pair index by convention."""
    t = random_seq(seed, tl)
    q = mutate(t, seed, p)
    if n_long > 0:
        q = long_indels(t, q, seed, n_long, long_max)
    return t, q


def synth_diverged_block(seed: int, flank: int, block_t: int, block_q: int, p: float) -> tuple[bytes, bytes]:
    """A pair whose middle does not align at all: shared flanks (mutated at rate p) around UNRELATED random blocks of
    block_t / block_q bases.  With both blocks >= 10 kb this is the case chain mode bridges with one deletion + one
    insertion instead of a gap fill (reference miniwfa.c:869, the `mwf_ksim < 0.02` branch)."""
    a, b = random_seq(seed, flank), random_seq(seed + 1, flank)
    t = a + random_seq(seed + 2, block_t) + b
    q = mutate(a, seed + 4, p) + random_seq(seed + 3, block_q) + mutate(b, seed + 5, p)
    return t, q


def synth_batch(base_seed: int, n: int, tl: int, p: float) -> list[tuple[bytes, bytes]]:
    return [synth_pair(base_seed + i, tl, p) for i in range(n)]


def fuzz_pairs(seed: int, n: int, max_len: int = 4000) -> list[tuple[bytes, bytes]]:
    """Pairs that stress the band kernels' corner cases: lengths around the 16-base / 64-lane / 256-column granularities, random,
    homopolymer, tandem-repeat and low-complexity targets (long exact runs), queries mutated at 0 ... 40 %, and — every seventh —
    UNRELATED queries of another length (windows that reach both corners of the matrix, many shrinks, chunks that leave and re-enter
    the window).  Used by profiles/fuzz_band2_oracle.py and tests/test_gpu_parity.py."""
    rng = np.random.default_rng(seed)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)

    def rand_seq(m, kind):
        if kind == 0:
            return acgt[rng.integers(0, 4, m)]
        if kind == 1:
            return np.full(m, acgt[rng.integers(0, 4)], dtype=np.uint8)
        if kind == 2:
            return np.resize(acgt[rng.integers(0, 4, rng.integers(1, 40))], m)
        return acgt[rng.choice(4, m, p=[0.85, 0.05, 0.05, 0.05])]

    def mutate(t, p):
        out = []
        for b, r in zip(t, rng.random(len(t))):
            if r < p / 3:
                continue
            if r < 2 * p / 3:
                out.append(acgt[rng.integers(0, 4)])
            if r < p:
                out.append(acgt[rng.integers(0, 4)])
                continue
            out.append(b)
        return np.array(out, dtype=np.uint8)

    pairs = []
    for i in range(n):
        m = int(rng.choice([0, 1, 15, 16, 17, 63, 64, 65, 255, 256, 257, 511, 512, 1023, 1024, 1025, 2047, 2048])) if i % 3 == 0 else int(rng.integers(0, max_len))
        t = rand_seq(min(m, max_len), i % 4)
        q = mutate(t, float(rng.choice([0.0, 0.01, 0.05, 0.2, 0.4]))) if i % 7 else rand_seq(int(rng.integers(0, max(1, max_len * 3 // 8))), (i + 1) % 4)
        pairs.append((t.tobytes(), q.tobytes()))
    return pairs


def skewed_pairs(seed: int, n: int, lo: int, hi: int) -> list[tuple[bytes, bytes]]:
    """Pairs whose wavefront window MOVES: every third one unrelated (target and query of independent lengths in [lo, hi): the window reaches
    the corners of the matrix, diagonals run out of it and the window's start climbs across chunk boundaries), the others related through
    one long deletion or insertion plus 2 / 8 / 20 % substitutions (the window drifts to one side); target and query swapped at random.
    Used by profiles/fuzz_fold.py and tests/test_gpu_parity.py (the slot mapping of the packed band kernel must follow such windows)."""
    rng = np.random.default_rng(seed)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    out = []
    for i in range(n):
        tl, ql = int(rng.integers(lo, hi)), int(rng.integers(lo, hi))
        t = rng.integers(0, 4, tl).astype(np.uint8)
        kind = i % 3
        if kind == 0:
            q = rng.integers(0, 4, ql).astype(np.uint8)
        else:
            cut = int(rng.integers(20, max(21, tl // 3)))
            at = int(rng.integers(0, max(1, tl - cut)))
            q = np.concatenate([t[:at], t[at + cut:]]) if kind == 1 else np.concatenate([t[:at], rng.integers(0, 4, cut).astype(np.uint8), t[at:]])
            flip = rng.random(len(q)) < rng.choice([0.02, 0.08, 0.2])
            q = q.copy()
            q[flip] = (q[flip] + rng.integers(1, 4, int(flip.sum()))) & 3
        if rng.random() < 0.5:
            t, q = q, t
        out.append((acgt[t].tobytes(), acgt[q].tobytes()))
    return out


class PackedBatch:
    """Pairs packed back to back in one byte buffer (+16 bytes of slack so word-sized device
    reads past the last sequence stay inside the allocation)."""

    def __init__(self, pairs):
        n = len(pairs)
        self.n = n
        self.tl = np.array([len(t) for t, _ in pairs], dtype=np.int32)
        self.ql = np.array([len(q) for _, q in pairs], dtype=np.int32)
        lens = np.empty(2 * n, dtype=np.int64)
        lens[0::2] = self.tl
        lens[1::2] = self.ql
        off = np.zeros(2 * n + 1, dtype=np.int64)
        np.cumsum(lens, out=off[1:])
        self.t_off = off[0:2 * n:2].copy()
        self.q_off = off[1:2 * n:2].copy()
        self.total = int(off[-1])
        buf = bytearray(self.total + 16)
        for i, (t, q) in enumerate(pairs):
            buf[self.t_off[i]:self.t_off[i] + len(t)] = t
            buf[self.q_off[i]:self.q_off[i] + len(q)] = q
        self.seqs = np.frombuffer(bytes(buf), dtype=np.uint8)

    @property
    def bases(self) -> int:
        return int(self.tl.sum() + self.ql.sum())


def spec_pair(spec: dict) -> tuple[bytes, bytes]:
    """(target, query) from a small generator spec — what tests/golden/blind_classes.jsonl stores instead of sequences.  Only
    the counter-based generators above are used, so a spec means the same bytes on every numpy version.
      unrelated: independent uniform sequences of tl and ql bases
      window   : the query is bases [at, at+w) of the target mutated at rate p (length-skewed, related)
      identical: query == target
      fit      : synth_pair(seed, tl, p) with the query cut or padded (random bases) to exactly ql bases
    `swap`: exchange target and query afterwards."""
    kind, seed = spec["kind"], int(spec["seed"])
    if kind == "unrelated":
        t, q = random_seq(seed, int(spec["tl"])), random_seq(seed + 1, int(spec["ql"]))
    elif kind == "window":
        t = random_seq(seed, int(spec["tl"]))
        q = mutate(t[int(spec["at"]):int(spec["at"]) + int(spec["w"])], seed + 1, float(spec["p"]))
    elif kind == "identical":
        t = q = random_seq(seed, int(spec["tl"]))
    elif kind == "fit":
        t = random_seq(seed, int(spec["tl"]))
        q = (mutate(t, seed, float(spec["p"])) + random_seq(seed + 2, int(spec["ql"])))[:int(spec["ql"])]
    else:
        raise ValueError(kind)
    return (q, t) if spec.get("swap") else (t, q)
