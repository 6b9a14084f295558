"""Pins the CPU restatement (oracle/mwf_oracle.c) to the real reference.

Every vector under tests/golden/ was produced by the compiled lh3/miniwfa (see
tests/golden/make_golden.py); the restatement must reproduce s, n_iter and the CIGAR exactly.
When oracle/_ref/libmwf_ref.so is present (build container, or the prebuilt copy that travelled
to the GPU box) a fresh randomised cross-check against it runs as well.
"""
import pytest

from conftest import load_golden, golden_inputs
from oracle.pyoracle import Oracle, Reference, Opt, make_opt, cigar_str, MWF_F_CIGAR


def _opt(v):
    return make_opt(**v["opt"])


def _check(oracle, v):
    t, q = golden_inputs(v)
    o = _opt(v)
    exp = v["expect"]
    if v["entry"] == "exact":
        s, n_iter, cig = oracle.align(t, q, o)
    elif v["entry"] == "auto":
        s, n_iter, cig = oracle.auto_exact_branch(t, q, o)
        assert s >= 0, "fixture is expected to stay on mwf_wfa_auto's exact branch"
    else:
        pytest.skip("chain-mode vectors belong to the next row (SURVEY §8 f1)")
    assert s == exp["s"], v["id"]
    assert n_iter == exp["n_iter"], v["id"]
    got = None if cig is None else cigar_str(cig)
    assert got == exp["cigar"], v["id"]
    if cig is not None:
        sc, tl, ql = oracle.cigar2score(o, cig)
        assert (tl, ql) == (len(t), len(q))
        assert sc == s  # observed for every vector (mwf-dbg.c:30 only warns when it is larger)


SMALL = [v for v in load_golden("exact_small.jsonl") if v["entry"] != "chain"]
BIG = [v for v in load_golden("bench_shaped.jsonl") if v["entry"] != "chain"]
BLIND = load_golden("blind_classes.jsonl")       # unrelated / length-skewed / climbing-window / side-by-side / class-limit pairs (tests/golden/make_golden_blind.py)
BIGPEN = load_golden("big_penalties.jsonl")   # max(x, o1+e1, o2+e2) >= 256: rings deeper than the fast kernels' tables (tests/golden/make_golden_bigpen.py)


def test_struct_layout():
    import ctypes as C
    from oracle.pyoracle import Rst
    assert C.sizeof(Opt) == 56 and C.sizeof(Rst) == 24
    assert Opt.max_iter.offset == 32 and Opt.min_len.offset == 48
    assert Rst.n_iter.offset == 8 and Rst.cigar.offset == 16


def test_opt_init_defaults(oracle):
    o = oracle.opt_init()  # miniwfa.c:11-18
    assert (o.flag, o.x, o.o1, o.e1, o.o2, o.e2, o.step, o.max_s, o.max_iter) == (0, 4, 4, 2, 15, 1, 0, 0, 0)
    assert (o.kmer, o.max_occ, o.min_len) == (13, 2, 30)


def test_t3_known_answer(oracle):
    v = SMALL[0]
    assert v["id"].startswith("t3")
    t, q = golden_inputs(v)
    s, _, cig = oracle.align(t, q, make_opt(flag=MWF_F_CIGAR))
    assert s == 155 and cigar_str(cig) == "1X16=1X14=128I4=1X24="


@pytest.mark.parametrize("chunk", range(8))
def test_small_golden(oracle, chunk):
    for v in SMALL[chunk::8]:
        _check(oracle, v)


@pytest.mark.parametrize("chunk", range(4))
def test_big_penalty_golden(oracle, chunk):
    assert len(BIGPEN) >= 200
    for v in BIGPEN[chunk::4]:
        _check(oracle, v)


@pytest.mark.parametrize("chunk", range(4))
def test_blind_class_golden(oracle, chunk):
    """The input classes round 5's fuzzers found bugs in (187 answers of the compiled reference): the restatement reproduces s, n_iter and
    the CIGAR (as text, or as the SHA-256 of its words where it is long)."""
    import hashlib
    import numpy as np
    from miniwfa_amd.synth import spec_pair
    assert len(BLIND) >= 150
    small = [v for v in BLIND if v["expect"]["n_iter"] <= 5e7]   # what the restatement finishes in about a minute in all; the GPU tests take every vector
    assert len(small) >= 150
    for v in small[chunk::4]:
        t, q = spec_pair(v["spec"])
        assert (len(t), len(q)) == (v["tl"], v["ql"]), v["id"]
        s, n_iter, cig = oracle.align(t, q, make_opt(**v["opt"]))
        exp = v["expect"]
        assert (s, n_iter) == (exp["s"], exp["n_iter"]), v["id"]
        assert (None if cig is None else len(cig)) == exp["n_cigar"], v["id"]
        if cig is not None and exp.get("cigar") is not None:
            assert cigar_str(cig) == exp["cigar"], v["id"]
        elif cig is not None:
            assert hashlib.sha256(np.asarray(cig, dtype="<u4").tobytes()).hexdigest() == exp["cigar_sha256"], v["id"]


@pytest.mark.parametrize("v", BIG, ids=[v["id"] for v in BIG])
def test_bench_shaped_golden(oracle, v):
    _check(oracle, v)


def test_checkpoints_are_on_the_optimal_path(oracle):
    """Low-memory pass 1 (miniwfa.c:551-601): every checkpoint (s,d) must be reproduced by pass 2's band
    resets — i.e. CIGAR and n_iter with step>0 match the goldens — and be monotone in s for step > max_pen."""
    from miniwfa_amd.synth import synth_pair
    t, q = synth_pair(30001, 3000, 0.1)
    seg = oracle.checkpoints(t, q, make_opt(flag=MWF_F_CIGAR, step=100))
    assert len(seg) > 3
    assert all(a[0] < b[0] for a, b in zip(seg, seg[1:]))
    assert all(-len(t) <= d <= len(q) for _, d in seg)


@pytest.mark.skipif(not Reference.available(), reason="oracle/_ref/libmwf_ref.so not built")
def test_fresh_fuzz_against_reference(oracle):
    from miniwfa_amd.synth import synth_pair, _stream
    ref = Reference()
    r = _stream(99, 9, 64)
    for j in range(24):
        tl = int(10 + r[2 * j] % 900)
        p = (0.02, 0.08, 0.2, 0.45)[j % 4]
        t, q = synth_pair(90000 + j, tl, p)
        for o in (make_opt(), make_opt(flag=MWF_F_CIGAR), make_opt(flag=MWF_F_CIGAR, step=1 + j % 37),
                  make_opt(flag=MWF_F_CIGAR, x=2, o1=3, e1=1, o2=11, e2=2, step=(j % 3) * 20)):
            assert oracle.align(t, q, o) == ref.align(t, q, o), (j, tl, p, o.step)


def test_chain_fixtures_are_reproducible():
    """tests/golden/chain_fresh.jsonl (row f1, reference miniwfa.c:850-907): the generators still produce the inputs the stored
    answers belong to; the chain's CIGAR consumes both sequences and its penalty is what the CIGAR costs; and — where the compiled
    reference is present — a sample of the answers is what the reference says today."""
    from miniwfa_amd.synth import synth_pair, synth_diverged_block
    import re
    vecs = load_golden("chain_fresh.jsonl")
    assert len(vecs) >= 60
    ref = Reference() if Reference.available() else None
    for n, v in enumerate(vecs):
        gen = v["gen"]
        t, q = synth_pair(*gen["args"]) if gen["kind"] == "synth" else synth_diverged_block(*gen["args"])
        assert (len(t), len(q)) == (v["tl"], v["ql"]), v["id"]
        cig = v["expect"]["cigar"]
        if cig is not None:
            ops = [(int(a), b) for a, b in re.findall(r"(\d+)([=XID])", cig)]
            assert sum(a for a, b in ops if b in "=XD") == len(t) and sum(a for a, b in ops if b in "=XI") == len(q), v["id"]
        if ref is not None and n % 5 == 0 and v["tl"] <= 30000:
            o = make_opt(**v["opt"])
            s, it, c = (ref.chain if v["entry"] == "chain" else ref.auto)(t, q, o)
            assert s == v["expect"]["s"] and (None if c is None else cigar_str(c)) == cig, v["id"]
