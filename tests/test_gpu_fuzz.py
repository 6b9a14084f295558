"""The randomised cross-checks (tests/fuzzlib.py) with fixed seeds, and the blind-class golden vectors, inside the driver-run `-m gpu` suite.

Round 5's fuzzers — then builder scripts under profiles/ — found two bit-exactness bugs that 977 golden vectors and 80 GPU tests had let
through (a slot mapping that followed a climbing window too early: n_iter off by 82; an identical pair side by side in two-pass mode).  Each
fuzzer now runs here on a budget of roughly half a minute, and tests/golden/blind_classes.jsonl (187 answers of the compiled reference,
tests/golden/make_golden_blind.py) covers the input classes those bugs lived in: unrelated pairs, length-skewed pairs, windows whose start
climbs across a chunk boundary, identical pairs side by side in low-memory mode, pairs exactly on the host's class limits.
Reference semantics at stake: miniwfa.c:139-171 (shrink), :396-426 (driver loop, edge rule :325-326), :413-416 and :551-601 (checkpoints)."""
import hashlib

import numpy as np
import pytest

import miniwfa_amd as mw
from miniwfa_amd.synth import PackedBatch, spec_pair, synth_pair
from conftest import load_golden
from oracle.pyoracle import Reference, make_opt, cigar_str as ocig
import fuzzlib as F

pytestmark = pytest.mark.gpu

OPT_KEYS = ("flag", "x", "o1", "e1", "o2", "e2", "step", "max_s", "max_iter")
BLIND = load_golden("blind_classes.jsonl")


def no_mismatches(bad):
    assert not bad, (len(bad), bad[:5])


# ---- the fuzzers ------------------------------------------------------------------------------------------------------------------

def test_fuzz_fold_vs_unfolded_vs_oracle():
    """Folded == unfolded on forced 512-thread, span and automatic geometries; default routing == oracle (score and CIGAR); shapes whose
    window start climbs.  (profiles/fuzz_fold.py seed 1 found the slot-mapping bug of round 5.)"""
    no_mismatches(F.fuzz_fold(seed=1, n=400, long_sets=False))   # (the seed and size that found the bug, all four penalty sets)
    no_mismatches(F.fuzz_fold(seed=5, n=200, long_sets=True, penalty_sets=(dict(), dict(x=3, o1=3, e1=2, o2=9, e2=2))))
    no_mismatches(F.fuzz_fold(seed=11, n=300, long_sets=False, penalty_sets=(dict(), dict(x=6, o1=6, e1=1, o2=30, e2=1))))


def test_fuzz_default_routing_on_mixed_batches():
    """What a caller of mwf_wfa_batch gets for a mixed batch (reads, medium, skewed, unrelated, a few 5-12 kb pairs): oracle's s, n_iter, CIGAR."""
    no_mismatches(F.fuzz_default_routing(seed=3, scale=2.0, penalty_sets=(dict(), dict(x=6, o1=2, e1=2, o2=20, e2=1))))
    no_mismatches(F.fuzz_default_routing(seed=4, scale=1.0))
    no_mismatches(F.fuzz_default_routing(seed=8, scale=1.5, penalty_sets=(dict(), dict(x=2, o1=2, e1=2, o2=12, e2=1))))


def test_fuzz_all_kernels_side_by_side_including_two_pass():
    """Generic kernel (one / four columns per lane, 16-bit rows), forced band geometries, whole-device kernel; score, CIGAR, low-memory
    two-pass (step = 97: groups of pairs side by side take the provenance pass) — against the oracle."""
    no_mismatches(F.fuzz_all_kernels(seed=2, n_pairs=120, wd_pairs=16))
    no_mismatches(F.fuzz_all_kernels(seed=9, n_pairs=120, wd_pairs=16, modes=(dict(flag=1, step=97), dict(flag=1, step=31), dict(flag=1, x=6, o1=2, e1=2, o2=20, e2=1, step=64))))


def test_fuzz_packed_band_geometries():
    no_mismatches(F.fuzz_band2(seed=6, n_pairs=300, blocks=(0, 64, 128, 256, 512, 768)))
    no_mismatches(F.fuzz_band2(seed=14, n_pairs=300, blocks=(0, 512), modes=(dict(), dict(flag=1), dict(flag=1, step=50), dict(flag=1, max_s=400))))


def test_fuzz_ring16_rows_vs_32_bit_rows():
    no_mismatches(F.fuzz_ring16(seed=2, n_pairs=40))


def test_fuzz_seq2bit_vs_byte_copies():
    no_mismatches(F.fuzz_seq2(seed=3, n_pairs=600))
    no_mismatches(F.fuzz_seq2(seed=4, n_pairs=600, modes=(dict(), dict(flag=1))))


@pytest.mark.skipif(not Reference.available(), reason="oracle/_ref/libmwf_ref.so (the compiled reference) did not travel; chain_fresh.jsonl covers chain mode")
def test_fuzz_chain_and_auto_vs_compiled_reference():
    no_mismatches(F.fuzz_chain(seed=7, n_pairs=30))


# ---- the device-side retry list (no pytest reached it) -------------------------------------------------------------------

def test_device_side_retry_list_with_both_lane_classes(oracle):
    """A read batch in which BOTH lane classes (10: plain A/C/G/T, 12: reads with an N) hold >= 1024 pairs and both hand pairs back (20-25 %
    divergence outgrows the lane kernel's chunks): the follow-up launches take their pairs from device-side lists (mwf_plan.cpp lane_retry,
    BatchArgs::retry_ids).  dev_retry 1 and 0 must both give the oracle's s, n_iter and CIGARs; no pair may be run twice into the CIGAR pool."""
    rng = np.random.default_rng(99)
    pairs = []
    for i in range(2600):
        tl = int(rng.integers(100, 300))
        p = float(rng.choice([0.02, 0.05, 0.2, 0.25], p=[0.45, 0.45, 0.05, 0.05]))
        t, q = synth_pair(880000 + i, tl, p)
        if i % 2 and len(q) > 10:   # half of the reads carry an N: the byte-wise lane class
            k = int(rng.integers(0, len(q)))
            q = q[:k] + b"N" + q[k + 1:]
        if abs(len(t) - len(q)) <= 24:
            pairs.append((t, q))
    assert len(pairs) >= 2300
    pk = PackedBatch(pairs)
    for flag in (0, 1):
        exp = F.oracle_many(oracle, pairs, make_opt(flag=flag))
        seen = {}
        for dev_retry in (1, 0):
            s, it, cig, st = F.run_engine(pk, dict(flag=flag), [("dev_retry", dev_retry)])
            bad = []
            F.compare((s, it, cig, st), exp, f"dev_retry {dev_retry} flag {flag}", pairs, bad, False)
            no_mismatches(bad)
            seen[dev_retry] = st.n_retries
        assert seen[1] > 0 and seen[0] > 0, seen   # the batch did hand pairs back on both paths
        assert seen[1] <= seen[0] + 2 * 256, seen  # (the device path counts what it re-ran, never more than the lists hold on top of the host path)


def test_gap_fill_like_batch_few_mid_size_pairs_among_hundreds_of_tiny_ones(oracle):
    """mwf_wfa_chain's gap fills (reference miniwfa.c:861-891 calls mwf_wfa_exact per gap): hundreds of tiny pairs and a handful of longer ones, some
    of very different lengths — a 54 x 740 fill must open a 686-base gap, so its window reaches that diagonal whatever its divergence.  Until
    round 6 the length limits put such fills into the 64-thread band class (run twice, every call), and a batch of more than 256 pairs kept its
    few mid-size pairs off the mid kernel.  Now: no pair is run twice, with the mid kernel (default) and without it (mid_max_pairs 0), and
    every answer is the oracle's."""
    rng = np.random.default_rng(606)
    pairs = [synth_pair(606000 + i, int(rng.integers(12, 90)), 0.06) for i in range(600)]
    t, q = synth_pair(606900, 900, 0.04)
    pairs += [(t[:54], q[:740]), (t[:555], q[:17]), (t[:899], q[400:499]), (t[:310], q[:330]), synth_pair(606901, 420, 0.10), (t[:120], q[:610])]
    order = rng.permutation(len(pairs))
    pairs = [pairs[i] for i in order]
    pk = PackedBatch(pairs)
    for flag in (1, 0):
        exp = F.oracle_many(oracle, pairs, make_opt(flag=flag))
        for tunables in ((), (("mid_max_pairs", 0),)):
            got = F.run_engine(pk, dict(flag=flag), list(tunables))
            bad = []
            F.compare(got, exp, f"gap fills flag {flag} {tunables}", pairs, bad, False)
            no_mismatches(bad)
            assert got[3].n_retries == 0, (flag, tunables, got[3].n_retries)
    # ... and the same handful in a batch small enough for the mid kernel anyway
    few = [pairs[i] for i in range(len(pairs)) if max(len(pairs[i][0]), len(pairs[i][1])) > 100]
    assert len(few) == 6
    exp = F.oracle_many(oracle, few, make_opt(flag=1))
    got = F.run_engine(PackedBatch(few), dict(flag=1))
    bad = []
    F.compare(got, exp, "the handful alone", few, bad, False)
    no_mismatches(bad)
    assert got[3].n_retries == 0 and got[3].packed == 33, (got[3].n_retries, got[3].packed)


@pytest.mark.parametrize("pen", [dict(x=4, o1=6, e1=3, o2=26, e2=1), dict(x=2, o1=4, e1=4, o2=24, e2=2), dict(x=3, o1=5, e1=3, o2=20, e2=3)])
def test_penalties_the_band_kernel_is_not_built_for_still_get_the_lane_and_mid_kernels(pen, oracle):
    """Gap extensions other than (2,1), (2,2), (1,1) — minimap2's asm5 / asm20 use 3 and 4 — leave the generic kernel for long pairs, but the lane and mid kernels read
    their penalties at run time (the reference has one loop for any penalties, miniwfa.c:261-327): a read batch with a few mid-size pairs among it must come back
    with the oracle's answers, the reads from the lane kernel (stats.packed 32 of the last launch), nothing on the generic kernel twice."""
    rng = np.random.default_rng(77)
    reads = [synth_pair(770000 + i, int(rng.integers(100, 260)), float(rng.choice([0.03, 0.06]))) for i in range(2600)]
    mids = [synth_pair(771000 + i, int(rng.integers(600, 2400)), 0.05) for i in range(24)] + [synth_pair(772000, 6000, 0.04)]
    pairs = mids + reads
    pk = PackedBatch(pairs)
    for flag in (0, 1):
        exp = F.oracle_many(oracle, pairs, make_opt(flag=flag, **pen))
        got = F.run_engine(pk, dict(flag=flag, **pen))
        bad = []
        F.compare(got, exp, f"{pen} flag {flag}", pairs, bad, False)
        no_mismatches(bad)
        assert got[3].packed == 32, got[3].packed   # the last class launched is the reads': on the lane kernel
    # a handful of mid-size pairs alone: the mid kernel
    exp = F.oracle_many(oracle, mids[:8], make_opt(flag=1, **pen))
    got = F.run_engine(PackedBatch(mids[:8]), dict(flag=1, **pen))
    bad = []
    F.compare(got, exp, f"{pen} mid", mids[:8], bad, False)
    no_mismatches(bad)
    assert got[3].packed == 33, got[3].packed


def test_cached_plan_follows_the_round5_tunables(oracle):
    """test_cached_plan_follows_every_tunable for the tunables it left out: dev_retry, band_fold, div_aware."""
    pairs = [synth_pair(97500 + i, (120, 300, 900, 2500, 6000)[i % 5], (0.03, 0.08)[i % 2]) for i in range(60)]
    exp = [oracle.align(t, q, make_opt(flag=1)) for t, q in pairs]
    eng = mw.Engine(0)
    b = eng.upload(PackedBatch(pairs))
    for name, value in [(None, 0), ("dev_retry", 0), ("dev_retry", 1), ("band_fold", 0), ("band_fold", 1), ("div_aware", 0), ("div_aware", 1)]:
        if name:
            eng.set(name, value)
        for flag in (mw.MWF_F_CIGAR, 0):
            b.align(mw.opt_init(flag=flag))
            s, it, nc = b.results()
            for i, (es, eit, ecig) in enumerate(exp):
                assert (int(s[i]), int(it[i])) == (es, eit) and (not flag or b.cigar(i, int(nc[i])).tolist() == ecig), (name, value, flag, i)
    b.free()
    eng.close()


# ---- blind-class golden vectors (answers of the compiled reference) ----------------------------------------------------------------

def check_vector(v, s, n_iter, words):
    exp = v["expect"]
    assert (int(s), int(n_iter)) == (exp["s"], exp["n_iter"]), (v["id"], int(s), int(n_iter), exp["s"], exp["n_iter"])
    if exp["n_cigar"] is None:
        assert words is None or len(words) == 0, v["id"]
        return
    assert len(words) == exp["n_cigar"], (v["id"], len(words), exp["n_cigar"])
    if exp.get("cigar") is not None:
        assert ocig(words) == exp["cigar"], v["id"]
    else:
        assert hashlib.sha256(np.asarray(words, dtype="<u4").tobytes()).hexdigest() == exp["cigar_sha256"], v["id"]


def run_golden_batch(vs, tunables=()):
    """The vectors of one option set as ONE device batch under the given tunables."""
    key = tuple(vs[0]["opt"][k] for k in OPT_KEYS)
    assert all(tuple(v["opt"][k] for k in OPT_KEYS) == key for v in vs)
    pairs = [spec_pair(v["spec"]) for v in vs]
    for v, (t, q) in zip(vs, pairs):
        assert (len(t), len(q)) == (v["tl"], v["ql"]), v["id"]
    eng = mw.Engine(0)
    for k, val in tunables:
        eng.set(k, val)
    b = eng.upload(PackedBatch(pairs))
    b.align(mw.opt_init(**dict(zip(OPT_KEYS, key))))
    s, it, nc = b.results()
    for i, v in enumerate(vs):
        check_vector(v, s[i], it[i], b.cigar(i, int(nc[i])) if key[0] & 1 else None)
    st = eng.stats()
    b.free()
    eng.close()
    return st


def by_opt(vecs):
    groups = {}
    for v in vecs:
        groups.setdefault(tuple(v["opt"][k] for k in OPT_KEYS), []).append(v)
    return list(groups.values())


def test_blind_golden_vectors_default_routing():
    assert len(BLIND) >= 150
    for vs in by_opt(BLIND):
        run_golden_batch(vs)


def test_blind_golden_vectors_in_a_batch_too_large_for_the_mid_kernel():
    """The same vectors three times over in one batch (> 256 pairs, and more than 256 of them in the small band classes: the small-batch
    classes — mid kernel, whole-device kernel — are off, every pair runs in its band / span / generic class)."""
    for vs in by_opt([v for v in BLIND if not v["opt"]["step"]]):
        if len(vs) >= 20:
            big = vs * (256 // len(vs) + 2)
            assert len(big) > 256
            run_golden_batch(big)


@pytest.mark.parametrize("tunables", [
    (("force_kind", 2), ("block", 512), ("band_pack", 1), ("band_fold", 1)),
    (("force_kind", 2), ("block", 512), ("band_pack", 1), ("band_fold", 0)),
    (("band_span", 2),),
    (("wide_slots", 4),),
    (("force_kind", 0),),
    (("force_kind", 0), ("ring16", 2)),
], ids=["512-folded", "512-unfolded", "span", "four-slots", "generic", "generic-16bit"])
def test_blind_golden_vectors_forced_kernels(tunables):
    """Unrelated, skewed and climbing-window vectors of up to 10 kb per sequence through forced geometries of the packed band kernel (folded
    and unfolded), the span geometry, the four-slot geometry and the generic kernel."""
    vecs = [v for v in BLIND if v["group"] in ("unrelated", "skewed", "climb") and max(v["tl"], v["ql"]) <= 10000 and not v["opt"]["step"]
            and (v["opt"]["x"], v["opt"]["e1"], v["opt"]["e2"]) == (4, 2, 1)]
    assert len(vecs) >= 80
    for vs in by_opt(vecs):
        run_golden_batch(vs, tunables)


@pytest.mark.parametrize("group", ["side-by-side-0", "side-by-side-1"])
def test_blind_golden_side_by_side_low_memory_groups(group):
    """Eight pairs — identical ones among them — side by side in low-memory mode: default routing, and forced onto the whole-device kernel
    (six or more side by side take the two-pass provenance form, miniwfa.c:551-601)."""
    vs = [v for v in BLIND if v["group"] == group]
    assert len(vs) == 8 and sum(v["spec"]["kind"] == "identical" for v in vs) >= 2
    run_golden_batch(vs)
    st = run_golden_batch(vs, (("force_kind", 1),))
    assert st.kernel_kind == 1
    run_golden_batch(vs, (("force_kind", 0),))


def test_blind_golden_class_limit_pairs_one_call_each():
    """Pairs exactly on the class limits through the drop-in call, one pair per call (batch of 1: lane / mid / whole-device admission)."""
    for v in BLIND:
        if v["group"] != "class-limit":
            continue
        t, q = spec_pair(v["spec"])
        s, n_iter, cig = mw.wfa_exact(t, q, mw.opt_init(**v["opt"]))
        check_vector(v, s, n_iter, cig)


# ---- divergence-aware classes for device-resident batches -----------------------------------------------------

@pytest.mark.parametrize("div", [0.01, 0.15, 0.30])
def test_wrapped_batch_classes_follow_its_divergence(div, oracle):
    """mwf_gpu_batch_wrap (sequences already in HBM: the library never sees the bytes on the host) sketches a few pairs' 8-mers ON THE DEVICE
    when the batch is wrapped, so that its size classes follow its divergence like a host-built batch's do.  A resident 15 % or 30 % batch
    used to run ~every pair twice; now none — and the answers are the oracle's.  The reference has no classes (miniwfa.c:396-426)."""
    torch = pytest.importorskip("torch")
    pairs = [synth_pair(33000 + i, 2000, div) for i in range(300)]
    pk = PackedBatch(pairs)
    eng = mw.Engine(0)
    b = eng.wrap_packed(pk, torch.device("cuda", 0))
    b.align(mw.opt_init())
    s, it, _ = b.results()
    assert eng.stats().n_retries == 0, (div, eng.stats().n_retries)
    for i in range(0, len(pairs), 7):
        es, eit, _ = oracle.align(pairs[i][0], pairs[i][1], make_opt())
        assert (int(s[i]), int(it[i])) == (es, eit), (div, i)
    # with the sketch switched off the 30 % batch does re-run (the test is looking at the right thing)
    if div == 0.30:
        eng2 = mw.Engine(0)
        eng2.set("div_aware", 0)
        b2 = eng2.wrap_packed(pk, torch.device("cuda", 0))
        b2.align(mw.opt_init())
        s2, it2, _ = b2.results()
        assert eng2.stats().n_retries > 0
        assert (np.array(s2) == np.array(s)).all() and (np.array(it2) == np.array(it)).all()
        b2.free()
        eng2.close()
    b.free()
    eng.close()


# ---- multi-GPU readiness on one device (no 8-GPU node is available to the build) ---------------------------------

def test_batch_multi_eight_engines_on_one_device_config5_share(oracle):
    """mwf_wfa_batch_multi with EIGHT device ordinals (all 0 here — on an 8-GPU node they are 0..7): one rank's share of configs[4]
    (1250 x 50 kb @ 3 % is minutes on one device eight ways; 160 pairs here) dealt longest-first over eight engines on eight host threads,
    merged in the caller's order.  Equal to one engine's answers pair by pair; three pairs equal the oracle."""
    pairs = [synth_pair(60000 + i, 50000 if i % 4 else 20000 + 997 * (i % 13), 0.03) for i in range(160)]
    got = mw.wfa_batch_multi(pairs, mw.opt_init(), devices=[0] * 8)
    one = mw.wfa_batch_multi(pairs, mw.opt_init(), devices=[0])
    assert len(got) == 160 and [r[:2] for r in got] == [r[:2] for r in one]
    for i in (0, 77, 159):
        assert got[i][:2] == oracle.align(pairs[i][0], pairs[i][1], make_opt())[:2], i


# ---- one-pair-per-call callers: submit / wait and the opt-in coalescer -----------------------------------------

def test_submit_wait_gives_the_drop_in_answers(oracle):
    """mwf_wfa_submit / mwf_wfa_wait (include/miniwfa.h part 2): the loop of reference main.c:67-72 with "submit" as its body — 600 pairs of
    mixed lengths under three option sets interleaved, waited for in a different order; every result is what mwf_wfa_exact returns (the
    oracle's s, n_iter, CIGAR), CIGARs come from the caller's kalloc arena, and the dispatcher ran far fewer batches than there were pairs."""
    rng = np.random.default_rng(17)
    pairs = [synth_pair(66000 + i, int(rng.choice([60, 150, 400, 1000, 2500, 6000], p=[0.3, 0.3, 0.2, 0.1, 0.07, 0.03])), float(rng.choice([0.02, 0.08]))) for i in range(600)]
    kws = [dict(flag=1), dict(flag=0), dict(flag=1, x=2, o1=2, e1=2, o2=12, e2=1)]
    exp = [oracle.align(t, q, make_opt(**kws[i % 3])) for i, (t, q) in enumerate(pairs)]
    b0, j0 = mw.async_stats()
    L = mw.lib()
    km = L.km_init()
    jobs = [mw.wfa_submit(t, q, mw.opt_init(**kws[i % 3])) for i, (t, q) in enumerate(pairs)]
    got = [None] * len(jobs)
    for i in list(range(len(jobs) - 1, -1, -2)) + list(range(len(jobs) - 2, -1, -2)):   # odd ones first, backwards
        got[i] = jobs[i].wait(km)
    L.km_destroy(km)
    assert got == exp
    b1, j1 = mw.async_stats()
    assert j1 - j0 == 600 and b1 - b0 <= 60, (b1 - b0, j1 - j0)


def test_coalesced_single_calls_from_many_threads(oracle):
    """MWF_COALESCE_US (read once per process: a child process here): 16 host threads each looping plain mwf_wfa_exact on 1 kb pairs share
    launches through the dispatcher — same answers as the oracle's, and several calls per batch on average."""
    import subprocess, sys, json, os
    code = r'''
import json, sys, threading
sys.path.insert(0, %r)
import torch  # noqa: F401  (its HIP runtime first, like tests/conftest.py)
import miniwfa_amd as mw
from miniwfa_amd.synth import synth_pair
pairs = [synth_pair(67000 + i, 1000, 0.05) for i in range(16)]
out = [[] for _ in pairs]
def loop(k):
    t, q = pairs[k]
    for _ in range(40):
        out[k].append(mw.wfa_exact(t, q, mw.opt_init(flag=1)))
th = [threading.Thread(target=loop, args=(k,)) for k in range(16)]
[x.start() for x in th]; [x.join() for x in th]
print(json.dumps({"stats": mw.async_stats(), "res": [[r[0], r[1], r[2]] for r in (o[-1] for o in out)], "same": all(all(r == o[0] for r in o) for o in out)}))
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MWF_COALESCE_US="150")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["same"]
    for k, (s, it, cig) in enumerate(d["res"]):
        t, q = synth_pair(67000 + k, 1000, 0.05)
        assert (s, it, cig) == tuple(oracle.align(t, q, make_opt(flag=1))), k
    batches, jobs = d["stats"]
    assert jobs == 16 * 40 and batches * 3 <= jobs, d["stats"]   # at least three calls per launch on average
