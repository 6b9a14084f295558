"""Parity of the HIP path (through the C ABI) with the oracle and the reference's golden vectors.

Bar: bit-exact s, n_iter and CIGAR (integer/byte work — no tolerance).  Inputs are the committed
golden fixtures (answers produced by the real lh3/miniwfa) plus seeded synthetic pairs checked
against oracle/mwf_oracle.c at sizes the oracle finishes in seconds."""
import numpy as np
import pytest

import miniwfa_amd as mw
from miniwfa_amd.synth import synth_pair, fuzz_pairs, skewed_pairs, PackedBatch
from conftest import load_golden, golden_inputs
from oracle.pyoracle import make_opt, cigar_str as ocig

pytestmark = pytest.mark.gpu

OPT_KEYS = ("flag", "x", "o1", "e1", "o2", "e2", "step", "max_s", "max_iter")


def gpu_opt(v_opt):
    return mw.opt_init(**{k: v_opt[k] for k in OPT_KEYS})


def explain_band(t, q, o, oracle):
    """On a mismatch: first penalty at which the device band leaves the oracle's."""
    try:
        eng = mw.Engine(0)
        b = eng.upload(PackedBatch([(t, q)]))
        dev = b.debug_band(o, 0) - 1 - len(t)
        ref = np.array(oracle.band_trace(t, q, make_opt(**{k: getattr(o, k) for k in OPT_KEYS})), dtype=np.int32).reshape(-1, 2)
        n = min(len(dev), len(ref))
        bad = np.nonzero((dev[:n] != ref[:n]).any(axis=1))[0]
        if len(bad):
            j = int(bad[0])
            return f"band diverges at penalty {j + 1}: device {dev[j].tolist()} oracle {ref[j].tolist()} (lens {len(dev)}/{len(ref)})"
        return f"bands agree for {n} penalties (lens {len(dev)}/{len(ref)})"
    except Exception as e:  # diagnostics only
        return f"(band trace unavailable: {e})"


@pytest.fixture(scope="module")
def engine():
    e = mw.Engine(0)
    yield e
    e.close()


def run_vectors(engine, oracle, vecs):
    """Group vectors by option set, run each group as one device batch, compare with the stored answers."""
    groups = {}
    for v in vecs:
        groups.setdefault(tuple(v["opt"][k] for k in OPT_KEYS), []).append(v)
    for key, vs in groups.items():
        o = gpu_opt(dict(zip(OPT_KEYS, key)))
        pairs = [golden_inputs(v) for v in vs]
        b = engine.upload(PackedBatch(pairs))
        b.align(o)
        s, it, nc = b.results()
        for i, v in enumerate(vs):
            exp = v["expect"]
            why = ""
            if s[i] != exp["s"] or it[i] != exp["n_iter"]:
                why = explain_band(pairs[i][0], pairs[i][1], o, oracle)
            assert s[i] == exp["s"], (v["id"], int(s[i]), exp["s"], why)
            assert it[i] == exp["n_iter"], (v["id"], int(it[i]), exp["n_iter"], why)
            if exp["cigar"] is None:
                assert nc[i] == 0, v["id"]
            else:
                assert ocig(b.cigar(i, int(nc[i]))) == exp["cigar"], v["id"]
        b.free()


def test_t3_known_answer_through_drop_in_api():
    v = load_golden("exact_small.jsonl")[0]
    t, q = golden_inputs(v)
    s, n_iter, cig = mw.wfa_exact(t, q, mw.opt_init(flag=mw.MWF_F_CIGAR))
    assert s == 155 and mw.cigar_str(cig) == "1X16=1X14=128I4=1X24="
    s2, n_iter2, cig2 = mw.wfa_exact(t, q, mw.opt_init())
    assert (s2, cig2) == (155, None) and n_iter2 == n_iter == 16875


def test_small_golden_vectors(engine, oracle):
    vecs = [v for v in load_golden("exact_small.jsonl") if v["entry"] == "exact"]
    assert len(vecs) > 600
    run_vectors(engine, oracle, vecs)


def test_bench_shaped_golden_vectors(engine, oracle):
    run_vectors(engine, oracle, [v for v in load_golden("bench_shaped.jsonl") if v["entry"] == "exact"])


def test_big_penalty_golden_vectors(engine, oracle):
    """Penalty sets with max(x, o1+e1, o2+e2) >= 256 (gap opens of 254 ... 4000; the reference takes any, miniwfa.c:390-393) against
    the reference's answers: s, n_iter, CIGAR in score, high-memory and low-memory (step 50 / 700) modes, stop rules.  From a ring
    depth of 257 on the pairs run on the generic kernel's big-ring form; the set with o2+e2 = 255 still takes the fast kernels."""
    vecs = load_golden("big_penalties.jsonl")
    assert len(vecs) >= 200
    run_vectors(engine, oracle, vecs)
    # the same through the drop-in call
    v = next(v for v in vecs if v["id"].startswith("big-t3#") and v["opt"]["flag"] and v["opt"]["o2"] == 300 and not v["opt"]["step"])
    t, q = golden_inputs(v)
    s, _, cig = mw.wfa_exact(t, q, gpu_opt(v["opt"]))
    assert s == v["expect"]["s"] and ocig(cig) == v["expect"]["cigar"]
    with pytest.raises(RuntimeError):   # beyond the limit stated in include/miniwfa.h: refused with a message, not computed wrongly
        b = engine.upload(PackedBatch([(t, q)]))
        try:
            b.align(mw.opt_init(o2=5000))
        finally:
            b.free()


def test_auto_exact_branch_matches_reference():
    for v in load_golden("exact_small.jsonl") + load_golden("bench_shaped.jsonl"):
        if v["entry"] != "auto":
            continue
        t, q = golden_inputs(v)
        s, n_iter, cig = mw.wfa_auto(t, q, gpu_opt(v["opt"]))
        assert (s, n_iter) == (v["expect"]["s"], v["expect"]["n_iter"]), v["id"]
        assert (None if cig is None else mw.cigar_str(cig)) == v["expect"]["cigar"], v["id"]


def test_empty_and_degenerate_inputs(engine, oracle):
    # ("","") with MWF_F_CIGAR crashes the reference (miniwfa.c:406-407); here it is defined as s=0, no CIGAR
    assert mw.wfa_exact(b"", b"", mw.opt_init(flag=mw.MWF_F_CIGAR)) == (0, 0, None)
    assert mw.wfa_exact(b"", b"", mw.opt_init()) == (0, 0, None)
    for t, q in ((b"A", b""), (b"", b"ACGTA"), (b"ACGT", b"ACGT"), (b"A" * 5000, b"A" * 5000), (b"A" * 700, b"C" * 650)):
        for o in (make_opt(), make_opt(flag=1), make_opt(flag=1, step=3)):
            exp = oracle.align(t, q, o)
            got = mw.wfa_exact(t, q, mw.opt_init(flag=o.flag, step=o.step))
            assert got == exp, (t[:8], q[:8], o.flag, o.step)


def test_ragged_batch_against_oracle(engine, oracle):
    """One launch, pairs of very different size and divergence (and a few identical / empty ones)."""
    pairs = [synth_pair(81000 + i, (5, 40, 333, 1200, 2500, 4000)[i % 6], (0.0, 0.03, 0.12, 0.3)[i % 4]) for i in range(96)]
    pairs += [(b"", b"ACGT"), (b"ACGT", b""), (b"GATTACA", b"GATTACA")]
    for o in (make_opt(), make_opt(flag=1), make_opt(flag=1, step=64), make_opt(flag=1, x=2, o1=3, e1=1, o2=9, e2=2)):
        go = mw.opt_init(**{k: getattr(o, k) for k in OPT_KEYS})
        b = engine.upload(PackedBatch(pairs))
        b.align(go)
        s, it, nc = b.results()
        for i, (t, q) in enumerate(pairs):
            es, eit, ecig = oracle.align(t, q, o)
            why = explain_band(t, q, go, oracle) if (s[i], it[i]) != (es, eit) else ""
            assert (s[i], it[i]) == (es, eit), (i, len(t), len(q), o.flag, o.step, why)
            if ecig is not None:
                assert b.cigar(i, int(nc[i])).tolist() == ecig, (i, o.step)
        b.free()


def test_generic_kernel_wide_windows(oracle):
    """Generic kernel on windows wider than any register span, with E2/F2 in LDS: the window of the second pair passes
    16384 columns, so the pass moves E2/F2 from LDS to the HBM rows on the way; and the same with LDS switched off."""
    pairs = [synth_pair(89992, 9000, 0.1), synth_pair(89993, 20000, 0.12)]
    expect = {flag: [oracle.align(t, q, make_opt(flag=flag)) for t, q in pairs] for flag in (0, 1)}
    assert max(h - l + 1 for l, h in oracle.band_trace(*pairs[1], make_opt())) > 16384
    for lds, r16 in ((1, 0), (1, 1), (0, 0)):   # 32-bit rows; 16-bit rows (the default: packed recurrence on the codes); everything in HBM
        eng = mw.Engine(0)
        eng.set("force_kind", 0)
        eng.set("lds_e2", lds)
        eng.set("ring16", r16)
        for flag in (0, 1):
            b = eng.upload(PackedBatch(pairs))
            b.align(mw.opt_init(flag=flag))
            st = eng.stats()
            assert (st.kernel_kind, st.block, st.packed) == (0, (512 if r16 and not flag else 768) if lds else 512, 16 if r16 else 0)
            s, it, nc = b.results()
            for i in range(len(pairs)):
                es, eit, ecig = expect[flag][i]
                assert (int(s[i]), int(it[i])) == (es, eit), (lds, r16, flag, i)
                if ecig is not None:
                    assert b.cigar(i, int(nc[i])).tolist() == ecig, (lds, r16, flag, i)
            b.free()
        assert eng.stats().n_retries == 0
        eng.close()


def test_generic_kernel_16_bit_ring_rows(oracle):
    """The generic kernel's 16-bit ring rows (ring16; 2 = 1 since round 3 — round 2 took them only for batches of at least as many pairs as
    CUs take them): same s, n_iter and CIGAR as with 32-bit rows on ragged 13-33 kb pairs (tl = 21000 puts a window edge on a
    chunk boundary: what outlives a penalty in LDS must be collapsed like the coded rows), the oracle's answers on the shortest
    ones, and a pair whose offsets outgrow 16 bits (target + penalty > 65532) comes back through the 32-bit rows."""
    pairs = [synth_pair(7100 + i, 13000 + 4000 * (i % 6), 0.02 + 0.01 * (i % 5)) for i in range(18)]
    out = {}
    for r16 in (0, 2):
        eng = mw.Engine(0)
        eng.set("ring16", r16)
        eng.set("force_kind", 0)
        b = eng.upload(PackedBatch(pairs))
        for flag in (0, 1):
            b.align(mw.opt_init(flag=flag))
            s, it, nc = b.results()
            out[(r16, flag)] = (np.array(s), np.array(it), [b.cigar(i, int(nc[i])).tolist() for i in range(len(pairs))] if flag else None)
            assert eng.stats().kernel_kind == 0 and eng.stats().n_retries == 0
        b.free()
        eng.close()
    for flag in (0, 1):
        a, c = out[(0, flag)], out[(2, flag)]
        assert (a[0] == c[0]).all() and (a[1] == c[1]).all() and a[2] == c[2], flag
    for i in (0, 6, 12):   # the 13 kb pairs against the oracle
        es, eit, ecig = oracle.align(pairs[i][0], pairs[i][1], make_opt(flag=1))
        assert (int(out[(2, 1)][0][i]), int(out[(2, 1)][1][i])) == (es, eit) and out[(2, 1)][2][i] == ecig, i
    big = [synth_pair(7200, 52000, 0.10)]
    res = []
    for r16, aware in ((0, 0), (2, 0), (2, 1)):
        eng = mw.Engine(0)
        eng.set("ring16", r16)
        eng.set("force_kind", 0)
        eng.set("div_aware", aware)   # (0: the round-4 behaviour — the 16-bit rows are tried and outgrown; 1: the batch's k-mer sketch says 10 %, the 32-bit rows are taken at once)
        b = eng.upload(PackedBatch(big))
        b.align(mw.opt_init())
        s, it, nc = b.results()
        res.append((int(s[0]), int(it[0]), eng.stats().n_retries))
        b.free()
        eng.close()
    assert res[0][:2] == res[1][:2] == res[2][:2] and res[0][2] == 0 and res[1][2] == 1 and res[2][2] == 0 and res[0][0] + 52000 > 65532, res


def test_generic_kernel_16_bit_ring_rows_fuzz(oracle):
    """The 16-bit-ring kernel (packed recurrence on the codes, 2-bit sequence copies in device memory) on the fuzz pairs of
    synth.fuzz_pairs — granular lengths, repeats, unrelated pairs that fill the whole matrix — plus one 9 kb unrelated pair that makes
    the batch take the kernel's wide form: s, n_iter and CIGAR equal the oracle's; a pair outside plain ACGT comes back through the
    32-bit rows."""
    pairs = fuzz_pairs(43, 90, 2500)
    rng = np.random.default_rng(1043)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    pairs.append((acgt[rng.integers(0, 4, 5200)].tobytes(), acgt[rng.integers(0, 4, 3900)].tobytes()))
    pairs.append((pairs[5][0].replace(b"A", b"N", 1) if b"A" in pairs[5][0] else b"NNAC", pairs[5][1]))
    for kw in (dict(), dict(flag=1)):
        o = make_opt(**kw)
        eng = mw.Engine(0)
        eng.set("force_kind", 0), eng.set("ring16", 2)
        b = eng.upload(PackedBatch(pairs))
        b.align(mw.opt_init(**kw))
        assert eng.stats().packed == 16
        s, it, nc = b.results()
        assert eng.stats().n_retries >= 1          # the pair with an N
        for i, (t, q) in enumerate(pairs):
            es, eit, ecig = oracle.align(t, q, o)
            assert (int(s[i]), int(it[i])) == (es, eit), (kw, i, len(t), len(q))
            if ecig is not None:
                assert b.cigar(i, int(nc[i])).tolist() == ecig, (kw, i)
        b.free()
        eng.close()


def test_generic_kernel_takes_16_bit_rows_for_big_batches():
    """Default admission of the 16-bit ring rows: long pairs on the generic kernel take them (stats.packed == 16) whatever the batch
    size (round 2: only batches of at least as many pairs as CUs), and give what 32-bit rows give."""
    pairs = [synth_pair(7400 + i, 12600 + 40 * (i % 11), 0.02) for i in range(300)]
    res = {}
    for r16, n in ((0, 300), (1, 300), (1, 40)):
        eng = mw.Engine(0)
        eng.set("ring16", r16)
        eng.set("band_span", 0)        # (by default pairs of this size take the packed band kernel's span geometry: this test is about the generic kernel)
        b = eng.upload(PackedBatch(pairs[:n]))
        b.align(mw.opt_init())
        s, it, nc = b.results()
        res[(r16, n)] = (np.array(s), np.array(it), eng.stats().packed, eng.stats().kernel_kind)
        b.free()
        eng.close()
    assert res[(0, 300)][3] == 0 and res[(0, 300)][2] == 0
    assert res[(1, 300)][2] == 16 and res[(1, 40)][2] == 16
    assert (res[(0, 300)][0] == res[(1, 300)][0]).all() and (res[(0, 300)][1] == res[(1, 300)][1]).all()
    assert (res[(0, 300)][0][:40] == res[(1, 40)][0]).all() and (res[(0, 300)][1][:40] == res[(1, 40)][1]).all()


def test_mixed_batch_runs_in_size_classes(oracle):
    """One batch with three very different pair sizes (what mwf_wfa_chain's gap fills look like): every size class goes
    to its own kernel in its own launch, results identical to the oracle, in the caller's order.  (The mid kernel, which would take
    the mid-size pairs of a batch this small, is switched off: this test is about the band classes; with it on the same batch is
    checked below.)"""
    eng = mw.Engine(0)
    eng.set("mid_max_pairs", 0)
    pairs, must = [], []
    for i in range(120):
        pairs.append(synth_pair(89000 + i, (40, 1500, 3500)[i % 3], (0.02, 0.1)[i % 2]))     # micro / tiny / small band kernel
        if i % 4 == 0:
            pairs.append(synth_pair(89500 + i, 6000, 0.05))                                   # wide band kernel
        if i % 60 == 0:
            pairs.append(synth_pair(89900 + i, 14000, 0.03))                                  # generic kernel (tl+ql > 24 kb)
            must.append(len(pairs))
            pairs.append(synth_pair(89950 + i, 3500, 0.3))   # short, so "small" — but its window outgrows that span: moved up
    for o in (make_opt(), make_opt(flag=1)):
        b = eng.upload(PackedBatch(pairs))
        b.align(mw.opt_init(flag=o.flag))
        assert eng.stats().n_launches == 5   # micro, tiny, small, wide band classes + generic
        s, it, nc = b.results()
        for i, (t, q) in enumerate(pairs):
            es, eit, ecig = oracle.align(t, q, o)
            assert (int(s[i]), int(it[i])) == (es, eit), (i, len(t), o.flag)
            if ecig is not None:
                assert b.cigar(i, int(nc[i])).tolist() == ecig, i
        assert eng.stats().n_retries >= 2       # the two divergent 3500 bp pairs went small -> wide (-> generic)
        b.free()
    eng.close()
    eng = mw.Engine(0)                          # defaults: the 1.5 - 3.5 kb pairs of this small batch take the mid kernel (one more class)
    b = eng.upload(PackedBatch(pairs))
    b.align(mw.opt_init(flag=1))
    s, it, nc = b.results()
    for i, (t, q) in enumerate(pairs):
        es, eit, ecig = oracle.align(t, q, make_opt(flag=1))
        assert (int(s[i]), int(it[i])) == (es, eit) and b.cigar(i, int(nc[i])).tolist() == ecig, (i, len(t))
    b.free()
    eng.close()


@pytest.mark.parametrize("block,scalar", [(64, 0), (128, 0), (256, 0), (512, 0), (1024, 0), (256, 1)])
def test_every_block_size_gives_identical_results(block, scalar, oracle):
    """Generic kernel (ring in HBM), every workgroup size: the four-columns-per-lane pass, and (scalar=1) the
    one-column-per-lane pass that otherwise only serves the low-memory mode."""
    eng = mw.Engine(0)
    eng.set("force_kind", 0)
    eng.set("block", block)
    eng.set("scalar_generic", scalar)
    pairs = [synth_pair(82000 + i, 1500, 0.08) for i in range(12)]
    for o in (make_opt(), make_opt(flag=1), make_opt(flag=1, step=50)):
        b = eng.upload(PackedBatch(pairs))
        b.align(mw.opt_init(flag=o.flag, step=o.step))
        s, it, nc = b.results()
        for i, (t, q) in enumerate(pairs):
            es, eit, ecig = oracle.align(t, q, o)
            assert (s[i], it[i]) == (es, eit), (block, i)
            if ecig is not None:
                assert b.cigar(i, int(nc[i])).tolist() == ecig
        b.free()
    eng.close()


@pytest.mark.parametrize("block,pack", [(64, -1), (128, -1), (256, -1), (512, -1), (768, -1)])
def test_band_kernel_against_oracle(block, pack, oracle):
    """Register-resident (packed) band kernel forced on, every geometry: ragged sizes, the penalty sets it is instantiated for, score
    and CIGAR.  Pairs whose window outgrows the span (block 256 holds < 2800 columns) must come back through the wider kernels with
    identical results."""
    eng = mw.Engine(0)
    eng.set("force_kind", 2)
    eng.set("block", block)
    eng.set("band_pack", pack)
    pairs = [synth_pair(86000 + i, (3, 60, 500, 1800, 3000, 5000)[i % 6], (0.0, 0.02, 0.1, 0.3)[i % 4]) for i in range(48)]
    pairs += [(b"", b"ACGT"), (b"ACGT", b""), (b"GATTACA", b"GATTACA"), (b"A" * 3000, b"A" * 3000), (b"A" * 900, b"C" * 800)]
    for o in (make_opt(), make_opt(flag=1), make_opt(flag=1, o2=4, e2=2), make_opt(flag=0, x=6, o1=2, e1=2, o2=20, e2=1)):
        go = mw.opt_init(**{k: getattr(o, k) for k in OPT_KEYS})
        b = eng.upload(PackedBatch(pairs))
        b.align(go)
        assert eng.stats().kernel_kind == 2
        s, it, nc = b.results()
        for i, (t, q) in enumerate(pairs):
            es, eit, ecig = oracle.align(t, q, o)
            why = explain_band(t, q, go, oracle) if (s[i], it[i]) != (es, eit) else ""
            assert (s[i], it[i]) == (es, eit), (block, i, len(t), len(q), o.flag, o.o2, why)
            if ecig is not None:
                assert b.cigar(i, int(nc[i])).tolist() == ecig, (block, i)
        if block <= 256:
            assert eng.stats().n_retries > 0   # the 3000/5000 bp pairs at 30 % do not fit 512 / 1280 / 2816 columns
        b.free()
    eng.close()


RETRIES = {}


@pytest.mark.parametrize("mode", ["bytes", "2bit"])
def test_sequence_copy_modes_and_alphabet_fallback(mode, oracle):
    """The packed band kernel with its byte-wise sequence copy (seq2bit = 0) and with the 2-bit copy (the default)
    on one batch: plain ACGT pairs of every size class plus pairs
    the 2-bit copy cannot hold (an N, lower case, arbitrary bytes, a lone non-ACGT last base) — those come back as
    ST_ALPHABET and are re-run byte-wise; every result equals the oracle's."""
    eng = mw.Engine(0)
    eng.set("seq2bit", 0 if mode == "bytes" else 1)
    pairs = [synth_pair(91000 + i, (150, 900, 3000, 6500)[i % 4], (0.02, 0.06, 0.12)[i % 3]) for i in range(24)]
    odd = []
    t, q = synth_pair(91100, 6500, 0.05); odd.append((t[:3000] + b"N" + t[3001:], q))
    t, q = synth_pair(91101, 900, 0.05); odd.append((t.lower(), q.lower()))
    t, q = synth_pair(91102, 3000, 0.05); odd.append((t, q[:-1] + b"n"))
    t, q = synth_pair(91103, 150, 0.1); odd.append((bytes(b ^ 0x15 for b in t), bytes(b ^ 0x15 for b in q)))
    odd.append((bytes(range(1, 200)), bytes(range(1, 100)) + bytes(range(101, 200))))
    pairs = pairs + odd
    for o in (make_opt(), make_opt(flag=1), make_opt(flag=1, o2=4, e2=2), make_opt(flag=1, x=1, o1=0, e1=1, o2=0, e2=1)):
        go = mw.opt_init(**{k: getattr(o, k) for k in OPT_KEYS})
        b = eng.upload(PackedBatch(pairs))
        b.align(go)
        s, it, nc = b.results()
        for i, (t, q) in enumerate(pairs):
            es, eit, ecig = oracle.align(t, q, o)
            assert (s[i], it[i]) == (es, eit), (mode, i, len(t), len(q), o.flag, o.o2)
            if ecig is not None:
                assert b.cigar(i, int(nc[i])).tolist() == ecig, (mode, i)
        # batches built from host memory: the host saw every byte while packing, so no pair takes the ST_ALPHABET round trip — what
        # is left are the pairs whose window outgrows their size class, the same in both modes
        RETRIES.setdefault((o.flag, o.o2, o.x), {})[mode] = eng.stats().n_retries
        seen = RETRIES[(o.flag, o.o2, o.x)]
        if len(seen) == 2:   # (round 6: the byte-wise copy runs in ONE geometry, 768 x 2 — its 24 chunks hold windows the small 2-bit classes hand back)
            assert seen["bytes"] <= seen["2bit"], seen
        b.free()
    if mode == "2bit":
        # device-resident inputs (mwf_gpu_batch_wrap): nobody looked at the bytes, the 2-bit copy finds out on the device and
        # every pair that is not plain ACGT goes round once more on the byte-wise copy
        import torch
        pk = PackedBatch(pairs)
        dev = torch.device("cuda", 0)
        bufs = [torch.from_numpy(a.copy()).to(dev) for a in (pk.seqs, pk.t_off, pk.tl, pk.q_off, pk.ql)]
        torch.cuda.synchronize(dev)
        b = eng.wrap(pk.n, bufs[0].data_ptr(), pk.total, bufs[1].data_ptr(), bufs[2].data_ptr(), bufs[3].data_ptr(), bufs[4].data_ptr(), pk.tl, pk.ql, keep=tuple(bufs))
        o = make_opt(flag=1)
        b.align(mw.opt_init(flag=1))
        s, it, nc = b.results()
        assert eng.stats().n_retries >= len(odd)
        for i, (t, q) in enumerate(pairs):
            es, eit, ecig = oracle.align(t, q, o)
            assert (s[i], it[i]) == (es, eit) and b.cigar(i, int(nc[i])).tolist() == ecig, ("wrapped", i)
        b.free()
    eng.close()


def test_band_kernel_stop_rules_and_shrink(oracle):
    """Long enough for several shrinks (every 256 penalties) plus the max_s / max_iter exits, band kernel forced."""
    eng = mw.Engine(0)
    eng.set("force_kind", 2)
    t, q = synth_pair(87000, 4000, 0.12)
    full = oracle.align(t, q, make_opt())
    assert full[0] > 1000
    for kw in (dict(), dict(max_s=full[0] - 1), dict(max_s=full[0]), dict(max_iter=full[1] - 1), dict(max_iter=full[1]), dict(max_s=700)):
        for flag in (0, 1):
            b = eng.upload(PackedBatch([(t, q)]))
            b.align(mw.opt_init(flag=flag, **kw))
            s, it, nc = b.results()
            es, eit, ecig = oracle.align(t, q, make_opt(flag=flag, **kw))
            assert (int(s[0]), int(it[0])) == (es, eit), (kw, flag)
            if ecig is not None:
                assert b.cigar(0, int(nc[0])).tolist() == ecig
            b.free()
    eng.close()


def test_whole_device_kernel_against_oracle(oracle):
    """The whole-device kernel (mwf_sys.hip) forced on (it is chosen automatically only for a few long pairs): score, CIGAR, low-memory mode
    (checkpoints read off the first pass's traceback matrix must give the reference's second-pass n_iter and CIGAR),
    stop rules, degenerate inputs.  Sizes reach several shrinks and tens of 256-column chunks."""
    eng = mw.Engine(0)
    eng.set("force_kind", 1)
    cases = [synth_pair(88000, 300, 0.1), synth_pair(88001, 3000, 0.05), synth_pair(88002, 20000, 0.04),
             synth_pair(88003, 9000, 0.2), synth_pair(88004, 15000, 0.01, 3, 2000),
             (b"", b"ACGT"), (b"ACGT", b""), (b"A" * 7000, b"A" * 7000), (b"A" * 1200, b"C" * 1100)]
    opts = [make_opt(), make_opt(flag=1), make_opt(flag=1, step=100), make_opt(flag=1, step=1000), make_opt(flag=1, step=5000),
            make_opt(flag=1, o2=4, e2=2), make_opt(flag=1, o2=4, e2=2, step=300)]
    for o in opts:
        go = mw.opt_init(**{k: getattr(o, k) for k in OPT_KEYS})
        for lo in range(0, len(cases), 3):
            pairs = cases[lo:lo + 3]
            b = eng.upload(PackedBatch(pairs))
            b.align(go)
            assert eng.stats().kernel_kind == 1
            s, it, nc = b.results()
            for i, (t, q) in enumerate(pairs):
                es, eit, ecig = oracle.align(t, q, o)
                why = explain_band(t, q, go, oracle) if (s[i], it[i]) != (es, eit) and o.step == 0 else ""
                assert (int(s[i]), int(it[i])) == (es, eit), (lo + i, len(t), len(q), o.flag, o.step, o.o2, why)
                if ecig is not None:
                    assert b.cigar(i, int(nc[i])).tolist() == ecig, (lo + i, o.step)
            b.free()
    t, q = cases[2]
    full = oracle.align(t, q, make_opt())
    for kw in (dict(max_s=full[0] - 1), dict(max_s=full[0]), dict(max_iter=full[1] - 1), dict(max_iter=full[1]), dict(max_s=500)):
        for flag, step in ((0, 0), (1, 0), (1, 700)):
            b = eng.upload(PackedBatch([(t, q)]))
            b.align(mw.opt_init(flag=flag, step=step, **kw))
            s, it, nc = b.results()
            assert (int(s[0]), int(it[0])) == oracle.align(t, q, make_opt(flag=flag, step=step, **kw))[:2], (kw, flag, step)
            b.free()
    eng.close()


def test_whole_device_kernel_many_shapes(oracle):
    """The whole-device kernel synchronises through tagged granules and a flag ring: a sweep over pair shapes (window
    from a few to ~150 chunks, runs from 7 to 500 bases, long indels) for the rare orderings a fixed case cannot hit."""
    eng = mw.Engine(0)
    eng.set("force_kind", 1)
    shapes = [(700, 0.15, 0, 0), (2500, 0.002, 0, 0), (5000, 0.08, 2, 300), (9000, 0.03, 0, 0), (12000, 0.12, 0, 0),
              (16000, 0.005, 3, 1500), (22000, 0.06, 0, 0), (30000, 0.02, 1, 4000)]
    pairs = [synth_pair(88200 + i, tl, p, nl, lm) for i, (tl, p, nl, lm) in enumerate(shapes)]
    pairs += [(q, t) for t, q in pairs[:4]]  # and the other way round (insertions <-> deletions)
    for o in (make_opt(), make_opt(flag=1), make_opt(flag=1, step=700)):
        for lo in range(0, len(pairs), 4):
            chunk = pairs[lo:lo + 4]
            b = eng.upload(PackedBatch(chunk))
            b.align(mw.opt_init(flag=o.flag, step=o.step))
            assert eng.stats().kernel_kind == 1
            s, it, nc = b.results()
            for i, (t, q) in enumerate(chunk):
                es, eit, ecig = oracle.align(t, q, o)
                assert (int(s[i]), int(it[i])) == (es, eit), (lo + i, len(t), len(q), o.flag, o.step)
                if ecig is not None:
                    assert b.cigar(i, int(nc[i])).tolist() == ecig, (lo + i, o.step)
            b.free()
    assert eng.stats().n_retries == 0
    eng.close()


def test_long_pairs_run_side_by_side(oracle):
    """A handful of long pairs (chosen automatically): each gets its own group of workgroups of the whole-device kernel and
    they run side by side; a pair whose window outgrows its group's share (the divergent one) is re-run with the device
    to itself.  Results against the oracle, score and CIGAR."""
    eng = mw.Engine(0)
    pairs = [synth_pair(88300 + i, 35000 + 3000 * i, 0.03) for i in range(5)] + [synth_pair(88310, 34000, 0.22)]
    for o in (make_opt(), make_opt(flag=1)):
        b = eng.upload(PackedBatch(pairs))
        b.align(mw.opt_init(flag=o.flag))
        st = eng.stats()
        assert st.kernel_kind == 1 and st.grid > 16 * 4          # several groups in one launch
        s, it, nc = b.results()
        for i, (t, q) in enumerate(pairs):
            es, eit, ecig = oracle.align(t, q, o)
            assert (int(s[i]), int(it[i])) == (es, eit), (i, o.flag)
            if ecig is not None:
                assert b.cigar(i, int(nc[i])).tolist() == ecig, i
        assert eng.stats().n_retries >= 1 and eng.stats().kernel_kind == 1   # the divergent pair went again, alone
        b.free()
    eng.close()


def test_whole_device_kernel_gives_up_gracefully(oracle, capfd):
    """The whole-device kernel synchronises workgroups by polling; every wait is bounded.  With the bound set to zero any
    wait that is not satisfied at once gives up: the pair must come back, bit-exact, through the one-workgroup kernel."""
    eng = mw.Engine(0)
    eng.set("force_kind", 1)
    eng.set("coop_spin_limit", 0)
    pairs = [synth_pair(88100, 6000, 0.05), synth_pair(88101, 2500, 0.1)]
    for o in (make_opt(), make_opt(flag=1), make_opt(flag=1, step=400)):
        b = eng.upload(PackedBatch(pairs))
        b.align(mw.opt_init(flag=o.flag, step=o.step))
        s, it, nc = b.results()
        for i, (t, q) in enumerate(pairs):
            es, eit, ecig = oracle.align(t, q, o)
            assert (int(s[i]), int(it[i])) == (es, eit), (i, o.flag, o.step)
            if ecig is not None:
                assert b.cigar(i, int(nc[i])).tolist() == ecig
        b.free()
    assert eng.stats().n_retries > 0
    assert "gave up waiting" in capfd.readouterr().err
    eng.close()


def test_chain_and_auto_against_reference():
    """mwf_wfa_chain (reference miniwfa.c:850-896): host chaining + one GPU batch of gap fills must give the reference's
    penalty and CIGAR, and mwf_wfa_auto must switch to it exactly when the exact branch stops at 1e8 cells (:898-907).
    Every answer is a stored fixture produced by the compiled reference (tests/golden/make_golden.py and
    make_golden_chain.py -> chain_fresh.jsonl: 40 fresh pairs x options, the auto fall-through, >= 10 kb blocks that do not
    align — the `mwf_ksim < 0.02` bridge of miniwfa.c:869 —, k-mer sizes 9 / 11 / 15); nothing here needs libmwf_ref.so."""
    from miniwfa_amd.synth import synth_diverged_block
    for v in load_golden("exact_small.jsonl") + load_golden("bench_shaped.jsonl"):
        if v["entry"] != "chain":
            continue
        t, q = golden_inputs(v)
        s, _, cig = mw.wfa_chain(t, q, gpu_opt(v["opt"]))
        assert s == v["expect"]["s"], v["id"]
        assert (None if cig is None else mw.cigar_str(cig)) == v["expect"]["cigar"], v["id"]
    n_auto_chain = n_bridge = 0
    for v in load_golden("chain_fresh.jsonl"):
        gen = v["gen"]
        t, q = synth_pair(*gen["args"]) if gen["kind"] == "synth" else synth_diverged_block(*gen["args"])
        assert (len(t), len(q)) == (v["tl"], v["ql"]), v["id"]
        o = mw.opt_init(**v["opt"])
        exp = v["expect"]
        if v["entry"] == "chain":
            s, _, cig = mw.wfa_chain(t, q, o)
        else:
            s, it, cig = mw.wfa_auto(t, q, o)
            if exp["n_iter"] > 100000000: # the exact branch gave up: the chain's answer, n_iter left at the exact branch's count (miniwfa.c:850-896 never writes it)
                n_auto_chain += 1
            assert it == exp["n_iter"], v["id"]
        assert s == exp["s"], v["id"]
        assert (None if cig is None else mw.cigar_str(cig)) == exp["cigar"], v["id"]
        if v["id"].startswith("diverged") and exp["cigar"] and gen["args"][2] >= 10000:
            n_bridge += 1
    assert n_auto_chain >= 1 and n_bridge >= 4


def test_chain_batch_gives_the_reference_chain_answers_pair_by_pair():
    """mwf_wfa_chain_batch: the chain mode of every record of a file (reference main.c:67-72 loops mwf_wfa_chain, miniwfa.c:850-896) with the gap fills of ALL pairs
    in one device batch.  Every chain-mode fixture of chain_fresh.jsonl (answers of the compiled reference), grouped by option set into one call each, must come
    back pair by pair — and the call must agree with mwf_wfa_chain on pairs that need no fill at all (identical sequences) and on an empty query."""
    from miniwfa_amd.synth import synth_diverged_block
    groups = {}
    for v in load_golden("chain_fresh.jsonl"):
        if v["entry"] != "chain":
            continue
        gen = v["gen"]
        t, q = synth_pair(*gen["args"]) if gen["kind"] == "synth" else synth_diverged_block(*gen["args"])
        groups.setdefault(tuple(sorted(v["opt"].items())), []).append((v, t, q))
    assert sum(len(g) for g in groups.values()) >= 40
    for key, vs in groups.items():
        o = mw.opt_init(**dict(key))
        got = mw.wfa_chain_batch([(t, q) for _, t, q in vs], o)
        for (v, t, q), (s, _, cig) in zip(vs, got):
            assert s == v["expect"]["s"], v["id"]
            assert (None if cig is None else mw.cigar_str(cig)) == v["expect"]["cigar"], v["id"]
    t, q = synth_pair(4242, 3000, 0.05, 2, 300)
    odd = [(t, t), (t, q), (t[:40], b""), (b"ACGT" * 10, b"ACGT" * 10), (q, t)]
    for flag in (0, 1):
        o = mw.opt_init(flag=flag)
        assert mw.wfa_chain_batch(odd, o) == [mw.wfa_chain(a, b, o) for a, b in odd]


def test_auto_batch_gives_the_reference_auto_answers_pair_by_pair():
    """mwf_wfa_auto_batch (reference miniwfa.c:898-908 once per record, main.c:67-72): the exact branch of every pair as one batch, chain mode for the pairs it gives
    up on at 1e8 cells — the `auto` fixtures of chain_fresh.jsonl (answers of the compiled reference, one of them beyond 1e8 cells) and short pairs in one call each per
    option set; n_iter is the exact branch's count where the chain answered."""
    from miniwfa_amd.synth import synth_diverged_block
    groups = {}
    for v in load_golden("chain_fresh.jsonl"):
        if v["entry"] != "auto":
            continue
        gen = v["gen"]
        t, q = synth_pair(*gen["args"]) if gen["kind"] == "synth" else synth_diverged_block(*gen["args"])
        groups.setdefault(tuple(sorted(v["opt"].items())), []).append((v, t, q))
    assert groups and any(v["expect"]["n_iter"] > 100000000 for g in groups.values() for v, _, _ in g)
    for key, vs in groups.items():
        o = mw.opt_init(**dict(key))
        got = mw.wfa_auto_batch([(t, q) for _, t, q in vs], o)
        for (v, t, q), (s, it, cig) in zip(vs, got):
            assert (s, it) == (v["expect"]["s"], v["expect"]["n_iter"]), v["id"]
            assert (None if cig is None else mw.cigar_str(cig)) == v["expect"]["cigar"], v["id"]
    small = [synth_pair(61000 + i, 300 + 40 * i, 0.06) for i in range(12)] + [(b"ACGT", b""), (b"", b"")]
    for flag in (0, 1):
        o = mw.opt_init(flag=flag)
        assert mw.wfa_auto_batch(small, o) == [mw.wfa_auto(a, b, o) for a, b in small]


def test_stop_rules(engine, oracle):
    t, q = synth_pair(83000, 2000, 0.1)
    full = oracle.align(t, q, make_opt())
    for kw in (dict(max_s=full[0] - 1), dict(max_s=full[0]), dict(max_iter=full[1] - 1), dict(max_iter=full[1]), dict(max_iter=1000), dict(max_s=300)):
        for flag, step in ((0, 0), (1, 0), (1, 100)):
            o = make_opt(flag=flag, step=step, **kw)
            assert mw.wfa_exact(t, q, mw.opt_init(flag=flag, step=step, **kw)) == oracle.align(t, q, o), (kw, flag, step)


def test_traceback_arena_retry(oracle):
    """A tiny traceback budget forces the overflow -> fewer/larger slots retry path."""
    eng = mw.Engine(0)
    eng.set("tb_budget_mb", 8)
    pairs = [synth_pair(84000 + i, 1500, 0.1) for i in range(40)]
    b = eng.upload(PackedBatch(pairs))
    b.align(mw.opt_init(flag=mw.MWF_F_CIGAR))
    s, it, nc = b.results()
    assert eng.stats().n_retries > 0
    for i, (t, q) in enumerate(pairs):
        es, eit, ecig = oracle.align(t, q, make_opt(flag=1))
        assert (s[i], it[i]) == (es, eit) and b.cigar(i, int(nc[i])).tolist() == ecig
    b.free()
    eng.close()


def test_whole_device_traceback_arena_grows(oracle):
    """Whole-device kernel with a 1 MB traceback arena: the overflow must double the arena and re-run until it fits."""
    eng = mw.Engine(0)
    eng.set("force_kind", 1)
    eng.set("coop_tb_cap_mb", 1)
    t, q = synth_pair(88100, 6000, 0.15)
    retries = []
    for kw in (dict(flag=1), dict(flag=1, step=400)):
        b = eng.upload(PackedBatch([(t, q)]))
        b.align(mw.opt_init(**kw))
        s, it, nc = b.results()
        es, eit, ecig = oracle.align(t, q, make_opt(**kw))
        assert (int(s[0]), int(it[0])) == (es, eit) and b.cigar(0, int(nc[0])).tolist() == ecig, kw
        assert eng.stats().kernel_kind == 1
        retries.append(eng.stats().n_retries)
        b.free()
    # the traceback of this pair is ~s^2 = several MB: the 1 MB arena of the first call must have overflowed and grown
    # (the second call finds the grown arena and may need no retry)
    assert es * es > (4 << 20) and retries[0] > 0, (es, retries)
    eng.close()


def test_drop_in_api_is_reentrant_across_threads(oracle):
    """The reference has no global mutable state (SURVEY §8b threading); here every host thread gets its own engine
    (stream + device pool).  Four threads hammer mwf_wfa_exact / mwf_wfa_auto concurrently, each with its own kalloc arena."""
    import threading
    L = mw.lib()
    pairs = [synth_pair(86500 + i, (200, 900, 2500)[i % 3], 0.07) for i in range(24)]
    expect = [oracle.align(t, q, make_opt(flag=1)) for t, q in pairs]
    errors = []

    def worker(tid):
        try:
            km = L.km_init()
            for rep in range(2):
                for i in range(tid, len(pairs), 4):
                    t, q = pairs[i]
                    fn = mw.wfa_exact if (i + rep) % 2 == 0 else mw.wfa_auto
                    got = fn(t, q, mw.opt_init(flag=mw.MWF_F_CIGAR), km=km)
                    if got != expect[i]:
                        errors.append((tid, i, got[:2], expect[i][:2]))
            L.km_destroy(km)
        except Exception as e:  # noqa: BLE001
            errors.append((tid, repr(e)))

    ts = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
    for t_ in ts:
        t_.start()
    for t_ in ts:
        t_.join()
    assert not errors, errors[:3]


def test_kalloc_arena_owns_the_cigar():
    L = mw.lib()
    km = L.km_init()
    t, q = synth_pair(85000, 800, 0.1)
    s, n_iter, cig = mw.wfa_exact(t, q, mw.opt_init(flag=mw.MWF_F_CIGAR), km=km)   # _take() kfree()s into km
    assert s > 0 and mw.cigar2score(mw.opt_init(), cig) == (s, len(t), len(q))
    L.km_destroy(km)


@pytest.mark.parametrize("block", [0, 512, 768])
def test_packed_band_kernel_fuzz_against_oracle(block, oracle):
    """Seeded fuzz of the packed band kernel (profiles/fuzz_band2_oracle.py runs more seeds): granular lengths, low-complexity and
    repetitive sequences, and unrelated pairs whose window reaches both corners of the matrix — the case that caught a query-index
    bit leaking from the pad column into its neighbour's probe shift.  s, n_iter and CIGAR equal the oracle's."""
    pairs = fuzz_pairs(13, 120, 3000)
    pk = PackedBatch(pairs)
    for kw in (dict(), dict(flag=1), dict(flag=1, o2=4, e2=2)):
        o = make_opt(**kw)
        eng = mw.Engine(0)
        if block:
            eng.set("force_kind", 2), eng.set("block", block), eng.set("band_pack", 1)
        b = eng.upload(pk)
        b.align(mw.opt_init(**kw))
        s, it, nc = b.results()
        for i, (t, q) in enumerate(pairs):
            es, eit, ecig = oracle.align(t, q, o)
            assert (int(s[i]), int(it[i])) == (es, eit), (block, kw, i, len(t), len(q))
            if ecig is not None:
                assert b.cigar(i, int(nc[i])).tolist() == ecig, (block, kw, i)
        b.free()
        eng.close()


def test_window_start_climbing_under_default_routing(oracle):
    """Unrelated pairs of 2-3 kb in one batch take the four-slot 512-thread geometry under the default routing; their windows reach the top
    of the matrix and their START climbs at the shrinks.  The slot mapping used to follow at once: the slot that wrapped ran the chunk 32
    further up while it aged, and that chunk's dead stores fell beyond the end of the row, into the next one (n_iter off by 82 on one pair
    of profiles/fuzz_fold.py seed 1 with x = o1 = 2).  Score and CIGAR, two penalty sets, against the oracle."""
    pairs = skewed_pairs(1, 400, 200, 3000)
    pairs = [p for i, p in enumerate(pairs) if i % 3 == 0 and max(len(p[0]), len(p[1])) >= 2000]
    assert len(pairs) >= 60
    pk = PackedBatch(pairs)
    for kw in (dict(x=2, o1=2, e1=2, o2=12, e2=1), dict()):
        exp = [oracle.align(t, q, make_opt(flag=1, **kw)) for t, q in pairs]
        for flag in (0, 1):
            for fold in (0, 1):
                eng = mw.Engine(0)
                eng.set("band_fold", fold)
                b = eng.upload(pk)
                b.align(mw.opt_init(flag=flag, **kw))
                s, it, nc = b.results()
                for i, (es, eit, ecig) in enumerate(exp):
                    assert (int(s[i]), int(it[i])) == (es, eit), (kw, flag, fold, i)
                    if flag:
                        assert b.cigar(i, int(nc[i])).tolist() == ecig, (kw, flag, fold, i)
                b.free()
                eng.close()


def test_folded_band_kernel_against_unfolded_and_oracle(oracle):
    """Score-only with o1 == x (the default penalties) the packed band kernel keeps max(E1, H[s-x]) in the registers that held E1 and never loads
    the row of lag o1+e1 (mwf_band2.hip: FOLD).  Pairs whose window moves UP (a query much shorter or longer than its target: diagonals run out
    of the matrix and the window's start climbs across chunk boundaries — the slot mapping must follow late), shrinks, several penalty sets
    with o1 == x and one without: folded == unfolded == oracle (s, n_iter), and the folded form's CIGARs == the oracle's."""
    rng = np.random.default_rng(4242)
    pairs = []
    for i in range(24):
        tl, ql = ((2600, 700), (700, 2600), (1800, 1500), (3000, 2900))[i % 4]
        t = rng.integers(0, 4, tl + int(rng.integers(0, 300))).astype(np.uint8)
        if i % 8 < 4:   # unrelated: the window reaches the corners of the matrix
            q = rng.integers(0, 4, ql + int(rng.integers(0, 300))).astype(np.uint8)
        else:           # related with one long gap: the window drifts to one side
            cut = int(rng.integers(50, 400))
            q = np.concatenate([t[:len(t) // 3], t[len(t) // 3 + cut:]])[:max(ql, 400)].copy()
            flip = rng.random(len(q)) < 0.06
            q[flip] = (q[flip] + rng.integers(1, 4, int(flip.sum()))) & 3
        pairs.append((bytes(b"ACGT"[int(c)] for c in t), bytes(b"ACGT"[int(c)] for c in q)))
    pairs += fuzz_pairs(77, 24, 2500)
    pk = PackedBatch(pairs)
    for kw in (dict(), dict(x=2, o1=2, e1=2, o2=12, e2=1), dict(x=6, o1=6, e1=1, o2=30, e2=1), dict(x=3, o1=3, e1=2, o2=9, e2=2), dict(x=4, o1=5, e1=2)):
        o = make_opt(**kw)
        want = [oracle.align(t, q, o)[:2] for t, q in pairs]
        for block in (0, 512, 1024):
            got = {}
            for fold in (1, 0):
                eng = mw.Engine(0)
                eng.set("band_fold", fold)
                if block == 1024:
                    eng.set("band_span", 2)   # every pair the span geometry can hold takes it (1024 threads x 5 slots, biased offsets)
                elif block:
                    eng.set("force_kind", 2), eng.set("block", block), eng.set("band_pack", 1)
                b = eng.upload(pk)
                b.align(mw.opt_init(**kw))
                s, it, _ = b.results()
                if block == 1024:
                    assert eng.stats().block == 1024
                got[fold] = [(int(a), int(c)) for a, c in zip(s, it)]
                b.free()
                eng.close()
            assert got[1] == want, (kw, block, [i for i in range(len(pairs)) if got[1][i] != want[i]][:5])
            assert got[0] == want, (kw, block)
        # with CIGAR the folded form records "this cell's E1 / F1 exceeds the H a gap would open from" and the walk reads that bit from the
        # cell an extension would come from (PairMem::tb_fwd): the CIGARs must still be the reference's
        exp = [oracle.align(t, q, make_opt(flag=1, **kw)) for t, q in pairs]
        for block in (512, 1024):
            eng = mw.Engine(0)
            if block == 1024:
                eng.set("band_span", 2)
            else:
                eng.set("force_kind", 2), eng.set("block", block), eng.set("band_pack", 1)
            b = eng.upload(pk)
            b.align(mw.opt_init(flag=1, **kw))
            s, it, nc = b.results()
            for i, (es, eit, ecig) in enumerate(exp):
                assert (int(s[i]), int(it[i])) == (es, eit) and b.cigar(i, int(nc[i])).tolist() == (ecig or []), (kw, block, i)
            b.free()
            eng.close()


@pytest.mark.parametrize("chunks", [1, 2, 4])
def test_lane_kernel_short_pairs_fuzz_against_oracle(chunks, oracle):
    """The one-diagonal-per-lane kernel (mwf_lane.hip) takes pairs of up to `lane_max_len` bases first; what outgrows its
    64 x `lane_chunks` columns is re-run on the band kernels.  Seeded fuzz (profiles/lane_kernel_probe.py runs more): corner-case
    lengths, empty sequences, bytes outside ACGT, read-like pairs, several penalty sets (the kernel is not specialised on them),
    max_s stops.  s, n_iter and CIGAR equal the oracle's."""
    rng = np.random.default_rng(77 + chunks)
    pairs = fuzz_pairs(31 + chunks, 90, 320)
    pairs += [(b"", b""), (b"A", b""), (b"", b"ACGT"), (b"ACGT", b"ACGT"), (b"ACGTNNRYACGT" * 9, b"ACGTNNRYACGA" * 9), (b"A" * 300, b"A" * 290)]
    pairs += [synth_pair(int(rng.integers(1 << 30)), int(rng.integers(60, 320)), float(rng.choice([0.01, 0.05, 0.1]))) for _ in range(60)]
    pk = PackedBatch(pairs)
    for kw in (dict(), dict(flag=1), dict(flag=1, o2=4, e2=2), dict(x=2, o1=3, e1=1, o2=6, e2=1), dict(flag=1, max_s=20)):
        o = make_opt(**kw)
        eng = mw.Engine(0)
        eng.set("lane_chunks", chunks)
        b = eng.upload(pk)
        b.align(mw.opt_init(**kw))
        s, it, nc = b.results()
        st = eng.stats()
        if "max_s" in kw:  # nothing outgrows the window before penalty 20: one launch, of the lane kernel
            assert (st.n_retries, st.block, st.packed) == (0, 64, 32)
        for i, (t, q) in enumerate(pairs):
            es, eit, ecig = oracle.align(t, q, o)
            assert (int(s[i]), int(it[i])) == (es, eit), (chunks, kw, i, len(t), len(q))
            if ecig is not None and es >= 0:
                assert b.cigar(i, int(nc[i])).tolist() == ecig, (chunks, kw, i)
        b.free()
        eng.close()
    eng = mw.Engine(0)
    eng.set("lane_max_len", 0)  # switched off (and the mid kernel, which takes the pairs of a batch this small, as well): the band kernels take the short pairs
    eng.set("mid_max_pairs", 0)
    b = eng.upload(pk)
    b.align(mw.opt_init(max_s=20))
    b.results()
    assert eng.stats().packed == 1
    b.free()
    eng.close()


def test_mid_kernel_fuzz_against_oracle(oracle):
    """The one-workgroup-per-pair kernel with every ring in LDS (mwf_mid.hip) serves the mid-size pairs of SMALL batches — the single
    pair of a drop-in call first of all.  Seeded fuzz: corner-case lengths, homopolymers and tandem repeats (long exact runs), unrelated
    pairs whose window leaves the span (re-run on the band kernels), pairs past several band shrinks (s >> 256), unequal lengths, bytes
    outside ACGT, five penalty sets (the kernel is not specialised on them; one the band kernels do not take at all), stop rules, both
    workgroup sizes.  s, n_iter and CIGAR equal the oracle's."""
    rng = np.random.default_rng(4141)
    pairs = fuzz_pairs(57, 40, 3000)
    pairs += [synth_pair(int(rng.integers(1 << 30)), int(rng.integers(401, 4000)), float(rng.choice([0.0, 0.01, 0.05, 0.1, 0.2]))) for _ in range(40)]
    pairs += [synth_pair(5150, 2500, 0.04, 2, 300), synth_pair(5151, 1500, 0.05, 3, 700), (b"ACGTNNRYACGT" * 90, b"ACGTNNRYACGA" * 90),
              (b"A" * 1500, b"A" * 1200), synth_pair(5152, 6000, 0.03), synth_pair(5153, 3000, 0.3)]
    pk = PackedBatch(pairs)
    for block, s2, kw in ((0, 1, dict()), (0, 1, dict(flag=1)), (256, 0, dict(flag=1, o2=4, e2=2)), (1024, 1, dict(x=2, o1=3, e1=1, o2=6, e2=1)), (512, 0, dict(flag=1, x=3, o1=5, e1=3, o2=20, e2=1)),
                          (0, 1, dict(flag=1, max_s=300)), (1024, 0, dict(max_iter=150000))):
        o = make_opt(**kw)
        eng = mw.Engine(0)
        eng.set("mid_block", block)
        eng.set("seq2bit", s2)      # 1: 2-bit sequence copies in LDS (pairs outside plain A/C/G/T take the byte-wise band classes); 0: byte copies, any alphabet
        b = eng.upload(pk)
        b.align(mw.opt_init(**kw))
        s, it, nc = b.results()
        for i, (t, q) in enumerate(pairs):
            es, eit, ecig = oracle.align(t, q, o)
            assert (int(s[i]), int(it[i])) == (es, eit), (block, kw, i, len(t), len(q))
            if ecig is not None and es >= 0:
                assert b.cigar(i, int(nc[i])).tolist() == ecig, (block, kw, i)
        b.free()
        eng.close()


def test_mid_kernel_serves_single_calls(oracle):
    """Which kernel a lone mid-size pair takes (stats.packed: 32 lane kernel, 33 mid kernel, 1 packed band kernel): a 2 kb pair runs on the
    mid kernel, score and CIGAR, with no re-run; a 300 bp pair stays on the lane kernel; with "mid_max_pairs" 0, and in a batch
    of more pairs than CUs, the band kernels take the 2 kb pairs.  Low-memory mode with a step the pair cannot reach is served as well."""
    t, q = synth_pair(123, 2000, 0.05)
    for kw in (dict(), dict(flag=1), dict(flag=1, step=5000)):
        eng = mw.Engine(0)
        b = eng.upload(PackedBatch([(t, q)]))
        b.align(mw.opt_init(**kw))
        st = eng.stats()
        assert (st.kernel_kind, st.packed, st.block) == (2, 33, 1024), (kw, st.kernel_kind, st.packed, st.block)   # (4 kb of sequence: sixteen waves)
        s, it, nc = b.results()
        assert eng.stats().n_retries == 0
        es, eit, ecig = oracle.align(t, q, make_opt(**kw))
        assert (int(s[0]), int(it[0])) == (es, eit), kw
        if ecig is not None:
            assert b.cigar(0, int(nc[0])).tolist() == ecig, kw
        b.free()
        eng.set("mid_max_pairs", 0)
        b = eng.upload(PackedBatch([(t, q)]))
        b.align(mw.opt_init(**kw))
        assert eng.stats().packed == 1
        s2, it2, _ = b.results()
        assert (int(s2[0]), int(it2[0])) == (es, eit)
        b.free()
        eng.close()
    eng = mw.Engine(0)
    b = eng.upload(PackedBatch([synth_pair(124, 300, 0.05)]))
    b.align(mw.opt_init())
    assert eng.stats().packed == 32
    b.results()
    b.free()
    many = [synth_pair(7000 + i, 1000, 0.05) for i in range(600)]
    b = eng.upload(PackedBatch(many))
    b.align(mw.opt_init())
    assert eng.stats().packed == 1
    s, it, _ = b.results()
    for i in (0, 299, 599):
        assert (int(s[i]), int(it[i])) == oracle.align(many[i][0], many[i][1], make_opt())[:2]
    b.free()
    eng.close()
    # through the drop-in call
    s, it, cig = mw.wfa_exact(t, q, mw.opt_init(flag=1))
    assert (s, it, cig) == oracle.align(t, q, make_opt(flag=1))


def test_small_batches_share_the_pinned_result_page(oracle):
    """Score-only batches of up to 64 pairs get their result arrays in ONE pinned host page per engine (no copy back); a second
    batch alive on the same engine keeps its device arrays, a CIGAR-mode align of the owner moves it back to them, and uploads
    that fit one pinned half are not waited for.  Whatever the placement, the results equal the oracle's."""
    o = make_opt()
    sets = [[synth_pair(900 + 10 * k + i, 120 + 37 * i, 0.05) for i in range(5)] for k in range(3)]
    want = [[oracle.align(t, q, o)[:2] for t, q in ps] for ps in sets]
    for host_results in (1, 0):
        eng = mw.Engine(0)
        eng.set("host_results", host_results)
        bs = [eng.upload(PackedBatch(ps)) for ps in sets[:2]]       # two uploads back to back, neither waited for
        for b in bs:
            b.align(mw.opt_init())                                   # both enqueued before either is read
        for b, w in zip(reversed(bs), reversed(want[:2])):           # read in the other order
            s, it, _ = b.results()
            assert [(int(x), int(y)) for x, y in zip(s, it)] == w
        bs[0].align(mw.opt_init(flag=mw.MWF_F_CIGAR))                # the page's owner in CIGAR mode: back to its device arrays
        s, it, nc = bs[0].results()
        assert [(int(x), int(y)) for x, y in zip(s, it)] == want[0]
        for i, (t, q) in enumerate(sets[0]):
            assert bs[0].cigar(i, int(nc[i])).tolist() == oracle.align(t, q, make_opt(flag=1))[2]
        bs[0].free()
        b3 = eng.upload(PackedBatch(sets[2]))                        # the page is free again
        for _ in range(2):
            b3.align(mw.opt_init())
            s, it, _ = b3.results()
            assert [(int(x), int(y)) for x, y in zip(s, it)] == want[2]
        bs[1].align(mw.opt_init())
        s, it, _ = bs[1].results()
        assert [(int(x), int(y)) for x, y in zip(s, it)] == want[1]
        eng.close()


def test_full_size_batch_properties(engine, oracle):
    """BASELINE config 3 at full size (1024 x 10 kb, 5 %): spot-check against the oracle, and check the
    size-independent properties on every pair: CIGAR re-scores to s and consumes both sequences."""
    pairs = [synth_pair(50000 + i, 10000, 0.05) for i in range(1024)]
    pk = PackedBatch(pairs)
    b = engine.upload(pk)
    b.align(mw.opt_init())
    s0, it0, _ = b.results()
    b.align(mw.opt_init(flag=mw.MWF_F_CIGAR))
    s1, it1, nc = b.results()
    assert (s0 == s1).all() and (it0 == it1).all() and (s0 > 0).all()
    for i in range(0, 1024, 97):
        assert (int(s0[i]), int(it0[i])) == oracle.align(pairs[i][0], pairs[i][1], make_opt())[:2], i
    o = mw.opt_init()
    for i in range(0, 1024, 8):
        cig = b.cigar(i, int(nc[i])).tolist()
        assert mw.cigar2score(o, cig) == (int(s1[i]), len(pairs[i][0]), len(pairs[i][1])), i
    b.free()


def test_config5_shaped_batch_properties(oracle):
    """BASELINE config 5's pair shape (50 kb, 3 %) on the generic kernel's wide path (768 threads, E2/F2 in LDS): one
    pair against the oracle, and on every pair the size-independent properties — score-only and CIGAR runs agree, the
    CIGAR re-scores to s and consumes both sequences."""
    engine = mw.Engine(0)
    engine.set("force_kind", 0)   # (a dozen such pairs alone would go side by side on the whole-device kernel)
    engine.set("ring16", 0)       # (32-bit ring rows: the default, 16-bit rows, is test_config5_default_kernel_at_full_pair_size)
    pairs = [synth_pair(60000 + i, 50000, 0.03) for i in range(12)]
    b = engine.upload(PackedBatch(pairs))
    b.align(mw.opt_init())
    st = engine.stats()
    assert (st.kernel_kind, st.block) == (0, 768)
    s0, it0, _ = b.results()
    b.align(mw.opt_init(flag=mw.MWF_F_CIGAR))
    s1, it1, nc = b.results()
    assert (s0 == s1).all() and (it0 == it1).all() and (s0 > 0).all()
    assert (int(s0[0]), int(it0[0])) == oracle.align(pairs[0][0], pairs[0][1], make_opt())[:2]
    o = mw.opt_init()
    for i in range(len(pairs)):
        cig = b.cigar(i, int(nc[i])).tolist()
        assert mw.cigar2score(o, cig) == (int(s1[i]), len(pairs[i][0]), len(pairs[i][1])), i
    b.free()
    engine.close()


def test_batch_multi_deals_pairs_over_engines(oracle):
    """mwf_wfa_batch_multi: the pairs of one call dealt longest-first over several engines on host threads (here three
    engines on device 0 — an ordinal may repeat; on a node with 8 GPUs they are 8 devices), results merged back in the
    caller's order, CIGARs allocated from the caller's arena on the calling thread."""
    pairs = [synth_pair(93000 + i, (30, 400, 2500, 9000)[i % 4] if i != 5 else 30000, (0.02, 0.08)[i % 2]) for i in range(41)]
    pairs += [(b"", b"ACGT"), (b"GATTACA", b"GATTACA")]
    for kw in (dict(flag=0), dict(flag=1), dict(flag=1, step=2000)):
        expect = [oracle.align(t, q, make_opt(**kw)) for t, q in pairs]
        L = mw.lib()
        km = L.km_init()
        got = mw.wfa_batch_multi(pairs, mw.opt_init(**kw), devices=[0, 0, 0], km=km)
        L.km_destroy(km)
        assert got == expect, kw
        assert mw.wfa_batch_multi(pairs[:3], mw.opt_init(**kw), n_dev=0) == expect[:3]   # every visible device
        assert mw.wfa_batch(pairs[:7], mw.opt_init(**kw)) == expect[:7]


def test_low_memory_mode_sends_short_pairs_to_the_band_kernel(oracle):
    """opt.step > 0 (mwf_wfa_auto's chain fallback hands step = 5000 to every gap fill): a pair whose penalty cannot reach
    `step` never takes a snapshot, so its low-memory result is its high-memory result — it must run on the band kernel,
    and only the genuinely long pair goes through the two-pass kernel.  Results identical to the reference's low-memory mode."""
    eng = mw.Engine(0)
    short = [synth_pair(93100 + i, (60, 300, 900)[i % 3], 0.06) for i in range(30)]
    o = make_opt(flag=1, step=5000)
    b = eng.upload(PackedBatch(short))
    b.align(mw.opt_init(flag=1, step=5000))
    assert eng.stats().kernel_kind == 2
    s, it, nc = b.results()
    for i, (t, q) in enumerate(short):
        es, eit, ecig = oracle.align(t, q, o)
        assert (int(s[i]), int(it[i])) == (es, eit) and b.cigar(i, int(nc[i])).tolist() == ecig, i
    b.free()
    mixed = short[:10] + [synth_pair(93150, 6000, 0.2)]     # penalty bound of the long one: far beyond 5000
    b = eng.upload(PackedBatch(mixed))
    b.align(mw.opt_init(flag=1, step=5000))
    assert eng.stats().n_launches >= 2
    s, it, nc = b.results()
    b.fetch_cigars()
    for i, (t, q) in enumerate(mixed):
        es, eit, ecig = oracle.align(t, q, o)
        assert (int(s[i]), int(it[i])) == (es, eit) and b.cigar(i, int(nc[i])).tolist() == ecig, i
    assert oracle.align(*mixed[-1], make_opt(flag=1))[1] != int(it[-1])   # the long pair really ran two-pass (its n_iter is the second pass's)
    b.free()
    eng.close()


def test_device_results_carry_a_not_final_sentinel(oracle):
    """Zero-copy consumers of the device result arrays: a pair that still needs a re-run reads s == -2 and status != 0
    until mwf_gpu_batch_results() has run it again (-1 stays the reference's "stopped" answer)."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")

    def peek(ptr, n):
        """n int32 words of device memory, after everything enqueued on the device has finished"""
        assert hip.hipDeviceSynchronize() == 0
        out = (ctypes.c_int32 * n)()
        assert hip.hipMemcpy(out, ctypes.c_void_p(ptr), ctypes.c_size_t(4 * n), 2) == 0   # hipMemcpyDeviceToHost
        return list(out)

    eng = mw.Engine(0)
    eng.set("force_kind", 2)
    eng.set("block", 64)                                     # span of 768 columns
    pairs = [synth_pair(93200, 200, 0.05), synth_pair(93201, 3000, 0.3), synth_pair(93202, 300, 0.05)]
    b = eng.upload(PackedBatch(pairs))
    b.align(mw.opt_init(max_s=0))
    d_s, d_st = peek(b.dev_scores_ptr(), 3), peek(b.dev_status_ptr(), 3)
    exp = [oracle.align(t, q, make_opt())[0] for t, q in pairs]
    assert d_s[0] == exp[0] and d_s[2] == exp[2] and d_st[0] == 0 and d_st[2] == 0
    assert d_s[1] == -2 and d_st[1] != 0                     # its window outgrew the 64-thread kernel's span
    s, it, nc = b.results()
    assert s.tolist() == exp and eng.stats().n_retries >= 1
    assert peek(b.dev_scores_ptr(), 3) == exp
    b.align(mw.opt_init(max_s=40))                           # stopped pairs read -1, as in the reference
    s, _, _ = b.results()
    assert s.tolist() == [oracle.align(t, q, make_opt(max_s=40))[0] for t, q in pairs] and -1 in s.tolist()
    b.free()
    eng.close()


def test_repeated_calls_recycle_device_memory():
    """The drop-in entry point in a loop: after the first call every allocation (batch block, CIGAR pool, workspace,
    pinned staging) is recycled — the engine's held memory does not grow and batches of the same shape reuse one block."""
    eng = mw.Engine(0)
    pk = PackedBatch([synth_pair(93300 + i, 500, 0.05) for i in range(16)])
    sizes = []
    for _ in range(5):
        b = eng.upload(pk)
        b.align(mw.opt_init(flag=1))
        b.results()
        b.free()
        sizes.append(eng.stats().dev_bytes)
    assert sizes[1] == sizes[-1] and sizes[-1] > 0, sizes
    eng.set("trim", 0)
    assert eng.stats().dev_bytes < sizes[-1]
    eng.close()


@pytest.mark.parametrize("kind", [2, 1])
def test_edit_distance_and_single_affine_presets_on_the_fast_kernels(kind, oracle):
    """The reference CLI's -e (x=1, o=0, e=1: every wavefront lag is 1, a ring of two slices) and -a (o2=o1, e2=e1)
    presets (main.c:34-35) are legal mwf_opt_t values: both must run on the band kernel (kind 2) and on the whole-device
    kernel (kind 1), not only on the generic one.  Score, CIGAR and, on the whole-device kernel, the low-memory mode."""
    EDIT = dict(x=1, o1=0, e1=1, o2=0, e2=1)
    AFFINE = dict(x=4, o1=4, e1=2, o2=4, e2=2)
    eng = mw.Engine(0)
    eng.set("force_kind", kind)
    pairs = [synth_pair(94000 + i, (50, 700, 2500, 6000)[i % 4], (0.01, 0.06, 0.2)[i % 3]) for i in range(12)]
    pairs += [(b"", b"ACGT"), (b"GATTACA", b"GATTACA"), (b"A" * 900, b"C" * 800)]
    if kind == 1:
        pairs = pairs[:6] + pairs[-3:] + [synth_pair(94100, 25000, 0.03, 2, 1500)]
    modes = [dict(flag=0), dict(flag=1)] + ([dict(flag=1, step=300)] if kind == 1 else [])
    for pen in (EDIT, AFFINE):
        for kw in modes:
            o = make_opt(**pen, **kw)
            for lo in range(0, len(pairs), 4 if kind == 1 else len(pairs)):
                chunk = pairs[lo:lo + (4 if kind == 1 else len(pairs))]
                b = eng.upload(PackedBatch(chunk))
                b.align(mw.opt_init(**pen, **kw))
                assert eng.stats().kernel_kind == kind, (pen, kw)
                s, it, nc = b.results()
                for i, (t, q) in enumerate(chunk):
                    es, eit, ecig = oracle.align(t, q, o)
                    assert (int(s[i]), int(it[i])) == (es, eit), (kind, pen["x"], kw, lo + i, len(t))
                    if ecig is not None:
                        assert b.cigar(i, int(nc[i])).tolist() == ecig, (kind, pen["x"], kw, lo + i)
                b.free()
    eng.close()


def test_whole_device_kernel_true_low_memory_mode(oracle):
    """The whole-device kernel's two-pass low-memory mode (reference mwf_wfa_seg, miniwfa.c:551-601): the first pass stores no
    traceback — provenance travels through shadow registers, shadow H rows and shadow granules, a snapshot every `step`
    penalties — and must yield the reference's checkpoints, i.e. its second-pass n_iter and CIGAR.  Forced by a 1 MB budget
    for the walk variant; steps from 1 (a snapshot at every penalty) to beyond the penalty."""
    eng = mw.Engine(0)
    eng.set("force_kind", 1)
    eng.set("lowmem_budget_mb", 1)
    cases = [synth_pair(95000, 300, 0.1), synth_pair(95001, 3000, 0.05), synth_pair(95002, 20000, 0.04),
             synth_pair(95003, 9000, 0.2), synth_pair(95004, 15000, 0.01, 3, 2000), synth_pair(95005, 30000, 0.03, 1, 4000),
             (b"", b"ACGT"), (b"ACGT", b""), (b"A" * 7000, b"A" * 7000), (b"A" * 1200, b"C" * 1100)]
    cases += [(q, t) for t, q in cases[1:4]]
    opts = [make_opt(flag=1, step=100), make_opt(flag=1, step=700), make_opt(flag=1, step=5000), make_opt(flag=1, step=1), make_opt(flag=1, step=7),
            make_opt(flag=1, step=256), make_opt(flag=1, o2=4, e2=2, step=300), make_opt(flag=1, x=1, o1=0, e1=1, o2=0, e2=1, step=150)]
    for o in opts:
        go = mw.opt_init(**{k: getattr(o, k) for k in OPT_KEYS})
        for lo in range(0, len(cases), 3):
            pairs = cases[lo:lo + 3]
            if o.step < 10:
                pairs = [p for p in pairs if len(p[0]) <= 9000]   # (a snapshot per penalty: keep it to seconds)
                if not pairs:
                    continue
            b = eng.upload(PackedBatch(pairs))
            b.align(go)
            st = eng.stats()
            assert st.kernel_kind == 1 and st.lowmem_two_pass == 1
            s, it, nc = b.results()
            for i, (t, q) in enumerate(pairs):
                es, eit, ecig = oracle.align(t, q, o)
                assert (int(s[i]), int(it[i])) == (es, eit), (lo + i, len(t), len(q), o.step, o.o2, o.x)
                assert b.cigar(i, int(nc[i])).tolist() == ecig, (lo + i, o.step)
            b.free()
    assert eng.stats().n_retries == 0
    # stop rules apply to the second pass only (miniwfa.c:569-589 has none)
    t, q = cases[2]
    full = oracle.align(t, q, make_opt(flag=1, step=700))
    for kw in (dict(max_s=full[0] - 1), dict(max_s=full[0]), dict(max_iter=full[1] - 1), dict(max_iter=full[1]), dict(max_s=500)):
        b = eng.upload(PackedBatch([(t, q)]))
        b.align(mw.opt_init(flag=1, step=700, **kw))
        s, it, nc = b.results()
        assert (int(s[0]), int(it[0])) == oracle.align(t, q, make_opt(flag=1, step=700, **kw))[:2], kw
        b.free()
    # memory: the 30 kb pair's first pass in both variants
    t, q = cases[5]
    peaks = {}
    for budget in (1, 1 << 20):
        e2 = mw.Engine(0)
        e2.set("force_kind", 1)
        e2.set("lowmem_budget_mb", budget)
        b = e2.upload(PackedBatch([(t, q)]))
        b.align(mw.opt_init(flag=1, step=1000))
        b.results()
        peaks[e2.stats().lowmem_two_pass] = e2.stats().dev_bytes_peak
        b.free()
        e2.close()
    assert set(peaks) == {0, 1} and peaks[1] < peaks[0], peaks
    eng.close()


def test_identical_pair_side_by_side_in_two_pass_low_memory_mode(oracle, capfd):
    """Six or more pairs side by side on the whole-device kernel take the two-pass low-memory form (their walk arenas together exceed the
    budget).  A pair of IDENTICAL sequences ends at penalty 0; its provenance pass used to report the end cell's provenance as 0 instead of
    -1 (the origin, miniwfa.c:119), the checkpoint trace refused the chain and the pair was re-run on the one-workgroup generic kernel with
    a "gave up waiting" warning (found by profiles/fuzz_all_kernels_oracle.py).  No re-run, no warning, the reference's answers."""
    same = synth_pair(96100, 2048, 0.0)[0]
    pairs = [synth_pair(96000 + i, 2000, 0.1) for i in range(8)] + [(same, same)]
    eng = mw.Engine(0)
    eng.set("force_kind", 1)
    o = make_opt(flag=1, step=97)
    b = eng.upload(PackedBatch(pairs))
    b.align(mw.opt_init(flag=1, step=97))
    s, it, nc = b.results()
    st = eng.stats()
    assert st.kernel_kind == 1 and st.lowmem_two_pass == 1 and st.n_retries == 0
    for i, (t, q) in enumerate(pairs):
        es, eit, ecig = oracle.align(t, q, o)
        assert (int(s[i]), int(it[i])) == (es, eit) and b.cigar(i, int(nc[i])).tolist() == ecig, i
    assert int(s[-1]) == 0
    b.free()
    eng.close()
    assert "gave up" not in capfd.readouterr().err


def test_two_threads_on_the_whole_device_kernel(oracle, capfd):
    """Two host threads, each aligning long pairs through the drop-in API at the same time: both land on the whole-device
    kernel, whose workgroups wait for one another and therefore must all be resident — launches are serialised per device,
    so neither starves the other (no wait gives up, nothing falls back to the one-workgroup kernel)."""
    import threading
    pairs = [synth_pair(96000 + i, 70000 + 5000 * i, 0.02) for i in range(4)]
    expect = [oracle.align(t, q, make_opt(flag=1)) for t, q in pairs]
    errors = []

    def worker(tid):
        try:
            for i in range(tid, len(pairs), 2):
                got = mw.wfa_exact(pairs[i][0], pairs[i][1], mw.opt_init(flag=mw.MWF_F_CIGAR))
                if got != expect[i]:
                    errors.append((tid, i, got[:2], expect[i][:2]))
        except Exception as e:  # noqa: BLE001
            errors.append((tid, repr(e)))

    ts = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for t_ in ts:
        t_.start()
    for t_ in ts:
        t_.join()
    assert not errors, errors[:3]
    assert "gave up waiting" not in capfd.readouterr().err


@pytest.mark.parametrize("span", [1, 0])
def test_config5_default_kernel_at_full_pair_size(span, oracle):
    """BASELINE configs[4]'s pair shape on the kernel a rank's share of that batch really selects: >= 256 pairs x 50 kb @ 3 % on
    DEFAULT settings go to the packed band kernel's 1024-thread span geometry (stats.kernel_kind == 2, stats.packed == 1, block 1024:
    16-bit offsets biased by the target length), with "band_span" 0 to the generic kernel with 16-bit ring rows (kernel_kind 0,
    packed 16).  Score-only and CIGAR runs agree, every CIGAR re-scores, four pairs equal the oracle, the stored reference answer of
    cfg5#698 is reproduced inside the batch, and one pair whose window outgrows the span (handed back early, with a forecast) and whose
    target + penalty exceeds 65 532 comes back through the generic kernel's 32-bit rows."""
    gold = [v for v in load_golden("bench_shaped.jsonl") if v["id"].startswith("cfg5") and v["entry"] == "exact" and not v["opt"]["flag"]]
    assert gold, "cfg5 golden vector missing"
    gt, gq = golden_inputs(gold[0])
    pairs = [synth_pair(60000 + i, 50000, 0.03) for i in range(254)]
    pairs.append((gt, gq))
    pairs.append(synth_pair(7200, 52000, 0.10))          # s ~ 15 k: target + penalty > 65 532
    eng = mw.Engine(0)
    eng.set("band_span", span)
    want = (2, 1, 1024) if span else (0, 16, 512)
    b = eng.upload(PackedBatch(pairs))
    b.align(mw.opt_init())
    st = eng.stats()
    assert (st.kernel_kind, st.packed, st.block) == want, (st.kernel_kind, st.packed, st.block)
    s0, it0, _ = b.results()
    assert eng.stats().n_retries == 1                    # the one pair that outgrew the span (its forecast also rules the 16-bit rows out) / the 16-bit rows
    assert (int(s0[254]), int(it0[254])) == (gold[0]["expect"]["s"], gold[0]["expect"]["n_iter"])
    assert int(s0[255]) + 52000 > 65532
    b.align(mw.opt_init(flag=mw.MWF_F_CIGAR))
    st = eng.stats()
    assert (st.kernel_kind, st.packed) == want[:2]
    s1, it1, nc = b.results()
    assert (s0 == s1).all() and (it0 == it1).all() and (s0 > 0).all()
    for i in (0, 85, 170, 253):
        assert (int(s0[i]), int(it0[i])) == oracle.align(pairs[i][0], pairs[i][1], make_opt())[:2], i
    o = mw.opt_init()
    b.fetch_cigars()
    for i in range(len(pairs)):
        cig = b.cigar(i, int(nc[i])).tolist()
        assert mw.cigar2score(o, cig) == (int(s1[i]), len(pairs[i][0]), len(pairs[i][1])), i
    b.free()
    eng.close()


def test_cigar_pool_in_block_mode_for_batches_of_thousands(oracle):
    """A batch of thousands of pairs takes its CIGAR pool in blocks (one atomic on the pool's head per dozen pairs instead of one per pair —
    40 000 of those on one address were half a millisecond; dev::finish_pair, BatchArgs::cig_block): 6000 read-length pairs on the lane kernel
    and on the packed band kernel's short-pair geometries, every CIGAR re-scores to its s and consumes both sequences, a sample equals the
    oracle's word for word — among them pairs whose CIGAR is longer than a quarter of a block (allocated on their own) and empty sequences."""
    rng = np.random.default_rng(99)
    pairs = [synth_pair(52000 + i, int(rng.integers(60, 260)), float(rng.choice([0.0, 0.02, 0.05, 0.1]))) for i in range(5900)]
    pairs += [synth_pair(53000 + i, 700, 0.12) for i in range(60)] + [(b"", b"ACGT"), (b"ACGT", b""), (b"", b"")] + [synth_pair(53100 + i, 400, 0.3) for i in range(37)]
    o = mw.opt_init(flag=mw.MWF_F_CIGAR)
    for lane_max in (400, 0):          # 0: the lane kernel switched off — the same pairs on the band classes
        eng = mw.Engine(0)
        eng.set("lane_max_len", lane_max)
        b = eng.upload(PackedBatch(pairs))
        b.align(o)
        s, it, nc = b.results()
        b.fetch_cigars()
        for i, (t, q) in enumerate(pairs):
            cig = b.cigar(i, int(nc[i])).tolist()
            assert mw.cigar2score(o, cig) == (int(s[i]), len(t), len(q)), (lane_max, i)
        for i in list(range(0, 6000, 41)) + list(range(5900, 6000)):
            es, eit, ecig = oracle.align(pairs[i][0], pairs[i][1], make_opt(flag=1))
            assert (int(s[i]), int(it[i])) == (es, eit) and b.cigar(i, int(nc[i])).tolist() == (ecig or []), (lane_max, i)
        b.free()
        eng.close()


def test_wide_class_chunk_slots_follow_the_batch(oracle):
    """The 512-thread geometry holds 24 chunks with three slots per wave and 32 with four (2 % slower where three suffice) and carries no forecast: with
    three, a pair whose window outgrows them late is re-run alone on the span geometry ("wide_slots" 3: one such pair in this batch).  By default a batch's
    first align under given options runs on four slots and reports whether three would have held every pair; later aligns follow that.  Same results
    whatever the slots; the pair in question equals the oracle."""
    pairs = [synth_pair(60000 + i, 10000, 0.05) for i in range(1024)]
    res = {}
    for slots in (3, 4, 0):
        eng = mw.Engine(0)
        eng.set("wide_slots", slots)
        b = eng.upload(PackedBatch(pairs))
        for rep in range(3):
            b.align(mw.opt_init())
            st = eng.stats()
            assert (st.kernel_kind, st.packed, st.block) == (2, 1, 512)
            s, it, _ = b.results()
            assert eng.stats().n_retries == (1 if slots == 3 else 0), (slots, rep, eng.stats().n_retries)
            res[(slots, rep)] = (np.array(s), np.array(it))
        if slots == 0:
            b.align(mw.opt_init(flag=mw.MWF_F_CIGAR))     # (other options: a new plan)
            s2, it2, nc = b.results()
            assert eng.stats().n_retries == 0 and (s2 == res[(3, 0)][0]).all() and (it2 == res[(3, 0)][1]).all()
            i = int(np.argmax(s2))
            assert mw.cigar2score(mw.opt_init(), b.cigar(i, int(nc[i])).tolist()) == (int(s2[i]), len(pairs[i][0]), len(pairs[i][1]))
        b.free()
        eng.close()
    for k, v in res.items():
        assert (v[0] == res[(3, 0)][0]).all() and (v[1] == res[(3, 0)][1]).all(), k
    i = int(np.argmax(res[(3, 0)][0]))
    assert (int(res[(0, 2)][0][i]), int(res[(0, 2)][1][i])) == oracle.align(pairs[i][0], pairs[i][1], make_opt())[:2]
    # a batch three slots hold: the first align measures (four slots), the others take three — nothing is ever re-run and the results stay
    pairs = [synth_pair(50000 + i, 10000, 0.05) for i in range(256)]
    eng = mw.Engine(0)
    b = eng.upload(PackedBatch(pairs))
    got = []
    for rep in range(3):
        b.align(mw.opt_init())
        s, it, _ = b.results()
        assert eng.stats().n_retries == 0
        got.append((np.array(s), np.array(it)))
    assert all((g[0] == got[0][0]).all() and (g[1] == got[0][1]).all() for g in got)
    assert (int(got[2][0][7]), int(got[2][1][7])) == oracle.align(pairs[7][0], pairs[7][1], make_opt())[:2]
    b.free()
    eng.close()


def test_512_thread_geometry_on_biased_offsets_takes_pairs_of_12_to_20_kb(oracle):
    """Pairs of ~11-21 kb — too long for plain 16-bit offsets by the worst-case rule, short enough for windows of up to ~10 000 / ~12 000 columns — run two per CU on
    the 512-thread geometry's five- and six-slot copies that compute on biased offsets with range checks (class 14; before: the span geometry, one per CU).  Pairs
    whose window outgrows their chunks move on to the span geometry.  Score and CIGAR runs equal the generic kernel's on every pair and the oracle's on a sample."""
    pairs = [synth_pair(71000 + i, 11000 + 97 * i, 0.05) for i in range(30)] + [synth_pair(71100 + i, 13500, 0.13) for i in range(3)] + [synth_pair(71200, 12000, 0.0), synth_pair(71201, 14000, 0.01)]
    pk = PackedBatch(pairs)
    ref = None
    for kw in (dict(), dict(flag=1)):
        eng = mw.Engine(0)
        eng.set("force_kind", 0)
        b = eng.upload(pk); b.align(mw.opt_init(**kw)); ref = [np.array(x) for x in b.results()]
        b.free(); eng.close()
        eng = mw.Engine(0)
        b = eng.upload(pk)
        b.align(mw.opt_init(**kw))
        st = eng.stats()
        assert (st.kernel_kind, st.packed, st.block) == (2, 1, 512), (st.kernel_kind, st.packed, st.block)
        s, it, nc = b.results()
        assert eng.stats().n_retries == 3            # the three pairs at 13 %: the span geometry
        assert (s == ref[0]).all() and (it == ref[1]).all()
        o = make_opt(**kw)
        for i in (0, 17, 29, 31, 33, 34):
            es, eit, ecig = oracle.align(pairs[i][0], pairs[i][1], o)
            assert (int(s[i]), int(it[i])) == (es, eit), (kw, i)
            if ecig is not None:
                assert b.cigar(i, int(nc[i])).tolist() == ecig, (kw, i)
        b.free()
        eng.close()
    # ... and the six-slot copy: 18-20 kb pairs
    pairs = [synth_pair(72000 + i, 18000 + 150 * i, 0.04) for i in range(12)]
    eng = mw.Engine(0)
    eng.set("coop_min_len", 1 << 40)
    b = eng.upload(PackedBatch(pairs))
    b.align(mw.opt_init(flag=1))
    st = eng.stats()
    assert (st.kernel_kind, st.packed, st.block) == (2, 1, 512)
    s, it, nc = b.results()
    assert eng.stats().n_retries == 0
    for i in (0, 11):
        es, eit, ecig = oracle.align(pairs[i][0], pairs[i][1], make_opt(flag=1))
        assert (int(s[i]), int(it[i])) == (es, eit) and b.cigar(i, int(nc[i])).tolist() == ecig, i
    b.free()
    eng.close()


def test_span_geometry_long_pairs_against_oracle(oracle):
    """The packed band kernel's 1024-thread geometry (16 waves x 5 chunk slots, offsets biased by the target length so that targets of up
    to ~60 kb fit 16 bits, mwf_band2.hip wide_bias): pairs of 20-60 kb whose windows stay below 20 000 columns, s, n_iter and CIGAR
    against the oracle — default penalties, the (e1, e2) = (1, 1) and (2, 2) instantiations, a stop rule inside the pass."""
    specs = [(50000, 0.03, 0, 0), (50000, 0.02, 0, 0), (40000, 0.035, 0, 0), (33000, 0.04, 0, 0), (20000, 0.06, 0, 0), (60000, 0.008, 0, 0),
             (50000, 0.034, 0, 0), (25000, 0.05, 0, 0), (45000, 0.02, 3, 2500)]
    pairs = [synth_pair(910 + i, tl, d, nl, lm) for i, (tl, d, nl, lm) in enumerate(specs)]
    pk = PackedBatch(pairs)
    for kw, sel in ((dict(), range(9)), (dict(flag=1), range(9)), (dict(flag=1, x=2, o1=3, e1=1, o2=6, e2=1), (1, 4)), (dict(x=6, o1=5, e1=2, o2=24, e2=2), (1, 5)),
                    (dict(max_s=3000), (0, 5)), (dict(flag=1, max_iter=20000000), (0, 1))):
        eng = mw.Engine(0)
        eng.set("coop_min_len", 1 << 40)   # (a few long pairs alone would share the whole-device kernel)
        eng.set("wide_slots", 3)           # (... and the pairs of up to ~21 kb would take the 512-thread geometry's copies on biased offsets: this test is about the span geometry)
        b = eng.upload(pk)
        b.align(mw.opt_init(**kw))
        st = eng.stats()
        assert (st.kernel_kind, st.packed, st.block) == (2, 1, 1024), (kw, st.kernel_kind, st.packed, st.block)
        s, it, nc = b.results()
        assert eng.stats().n_retries == 0, kw
        o = make_opt(**kw)
        for i in sel:
            es, eit, ecig = oracle.align(pairs[i][0], pairs[i][1], o)
            assert (int(s[i]), int(it[i])) == (es, eit), (kw, i)
            if ecig is not None and es >= 0:
                assert b.cigar(i, int(nc[i])).tolist() == ecig, (kw, i)
        b.free()
        eng.close()


def test_span_geometry_hands_back_what_it_cannot_hold(oracle):
    """What the span geometry cannot finish goes to the generic kernel and still equals the oracle: a window beyond its 80 chunks (handed
    back early with a forecast), a 60 kb target whose dead values would drift into the live range of the biased offsets after ~3400
    penalties (the range check at multiples of 256), two 62 kb sequences whose room arithmetic would leave 16 bits, bytes outside
    A/C/G/T (the geometry exists on 2-bit sequence copies only)."""
    t, q = synth_pair(77, 40000, 0.03)
    pairs = [synth_pair(7301, 40000, 0.11), synth_pair(7302, 60000, 0.025), synth_pair(7303, 62000, 0.022), (t[:20000] + b"N" + t[20001:], q), synth_pair(7304, 30000, 0.03)]
    eng = mw.Engine(0)
    eng.set("coop_min_len", 1 << 40)
    eng.set("div_aware", 0)   # (the classes by length alone, as for device-resident batches: this test is about what the span geometry hands back)
    b = eng.upload(PackedBatch(pairs))
    for kw in (dict(), dict(flag=1)):
        b.align(mw.opt_init(**kw))
        s, it, nc = b.results()
        assert eng.stats().n_retries >= 3, eng.stats().n_retries
        o = make_opt(**kw)
        for i, (t_, q_) in enumerate(pairs):
            es, eit, ecig = oracle.align(t_, q_, o)
            assert (int(s[i]), int(it[i])) == (es, eit), (kw, i)
            if ecig is not None:
                assert b.cigar(i, int(nc[i])).tolist() == ecig, (kw, i)
    b.free()
    eng.close()


def test_span_geometry_fuzz_on_short_pairs(oracle):
    """"band_span" 2 sends every pair the span geometry can take to it, whatever its size: the fuzz set of the other band kernels (corner-case
    lengths, empty sequences, homopolymers, tandem repeats, unrelated pairs, pairs past several shrinks) with a zero bias."""
    pairs = fuzz_pairs(91, 60, 3000) + [synth_pair(8100 + i, 700 * (i + 1), 0.04 * (1 + i % 3)) for i in range(12)] + [(b"", b""), (b"ACGT", b""), (b"", b"AC")]
    pk = PackedBatch(pairs)
    for kw in (dict(), dict(flag=1), dict(flag=1, x=2, o1=3, e1=1, o2=6, e2=1), dict(max_s=200)):
        eng = mw.Engine(0)
        eng.set("band_span", 2)
        b = eng.upload(pk)
        b.align(mw.opt_init(**kw))
        s, it, nc = b.results()
        o = make_opt(**kw)
        for i, (t, q) in enumerate(pairs):
            es, eit, ecig = oracle.align(t, q, o)
            assert (int(s[i]), int(it[i])) == (es, eit), (kw, i, len(t), len(q))
            if ecig is not None and es >= 0:
                assert b.cigar(i, int(nc[i])).tolist() == ecig, (kw, i)
        b.free()
        eng.close()


def test_batch_multi_one_ranks_share_of_config5(oracle):
    """BASELINE configs[4] through the drop-in batch entry point: one GPU's share of the 10 000 x 50 kb batch (1250 pairs @ 3 %,
    score-only) in ONE mwf_wfa_batch_multi call with n_dev = 0 (every visible device; on the test box that is one).  The stored
    reference answer of the cfg5 golden vector is reproduced inside the batch, three pairs equal the oracle, and the call agrees
    pair by pair with the device-resident API (Engine/Batch) on the same inputs."""
    gold = [v for v in load_golden("bench_shaped.jsonl") if v["id"].startswith("cfg5") and v["entry"] == "exact" and not v["opt"]["flag"]]
    assert gold, "cfg5 golden vector missing"
    pairs = [synth_pair(60000 + i, 50000, 0.03) for i in range(1250)]
    pairs[777] = golden_inputs(gold[0])
    got = mw.wfa_batch_multi(pairs, mw.opt_init(), n_dev=0)
    assert len(got) == 1250 and all(r[0] > 0 and r[2] is None for r in got)
    assert got[777][:2] == (gold[0]["expect"]["s"], gold[0]["expect"]["n_iter"])
    for i in (0, 612, 1249):
        assert got[i][:2] == oracle.align(pairs[i][0], pairs[i][1], make_opt())[:2], i
    eng = mw.Engine(0)
    b = eng.upload(PackedBatch(pairs))
    b.align(mw.opt_init())
    s, it, _ = b.results()
    st = eng.stats()
    assert (st.kernel_kind, st.packed, st.block) == (2, 1, 1024) and st.n_retries == 0   # the packed band kernel's span geometry
    assert [r[0] for r in got] == s.tolist() and [r[1] for r in got] == it.tolist()
    b.free()
    eng.close()


def test_two_threads_launch_kernels_with_large_dynamic_lds(oracle):
    """hipFuncAttributeMaxDynamicSharedMemorySize is per device and was once cached per process without a lock: two host
    threads (= two engines, as mwf_wfa_batch_multi has on a multi-GPU node) each launch kernels above 48 KB of dynamic LDS at the
    same time — the mid kernel (~140 KB: every ring of a 2-3 kb pair in LDS), with traceback and without."""
    import threading
    pairs = [[synth_pair(97000 + 10 * k + i, 2000 + 400 * i, 0.04) for i in range(3)] for k in range(2)]
    expect = [[oracle.align(t, q, make_opt(flag=1)) for t, q in ps] for ps in pairs]
    errors = []

    def worker(k):
        try:
            eng = mw.Engine(0)
            for rep in range(4):
                b = eng.upload(PackedBatch(pairs[k]))
                b.align(mw.opt_init(flag=rep & 1))
                s, it, nc = b.results()
                st = eng.stats()
                if st.kernel_kind != 2 or st.packed != 33:
                    errors.append((k, "kernel", st.kernel_kind, st.packed, st.block))
                for i in range(len(pairs[k])):
                    got = (int(s[i]), int(it[i]), b.cigar(i, int(nc[i])).tolist() if rep & 1 else expect[k][i][2])
                    if got != expect[k][i]:
                        errors.append((k, i, int(s[i]), expect[k][i][0]))
                b.free()
            eng.close()
        except Exception as e:  # noqa: BLE001
            errors.append((k, repr(e)))

    ts = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for t_ in ts:
        t_.start()
    for t_ in ts:
        t_.join()
    assert not errors, errors[:3]


def test_long_pair_and_batch_threads_share_the_device(oracle, capfd):
    """One host thread on the whole-device kernel (a long pair through the drop-in API), another launching one-workgroup-per-pair
    batches on its own engine: the whole-device kernel takes the device exclusively (launches wait for one another through a
    per-device gate), so its workgroups are all resident and no wait gives up."""
    import threading
    longp = synth_pair(96100, 90000, 0.02)
    exp_long = oracle.align(longp[0], longp[1], make_opt())
    short = [synth_pair(96200 + i, 3000, 0.05) for i in range(64)]
    exp_short = [oracle.align(t, q, make_opt())[:2] for t, q in short]
    errors = []

    def long_worker():
        try:
            for _ in range(3):
                got = mw.wfa_exact(longp[0], longp[1], mw.opt_init())
                if got[:2] != exp_long[:2]:
                    errors.append(("long", got[:2], exp_long[:2]))
        except Exception as e:  # noqa: BLE001
            errors.append(("long", repr(e)))

    def batch_worker():
        try:
            eng = mw.Engine(0)
            for _ in range(12):
                b = eng.upload(PackedBatch(short))
                b.align(mw.opt_init())
                s, it, _ = b.results()
                if [(int(a), int(c)) for a, c in zip(s, it)] != exp_short:
                    errors.append(("batch",))
                b.free()
            eng.close()
        except Exception as e:  # noqa: BLE001
            errors.append(("batch", repr(e)))

    ts = [threading.Thread(target=long_worker), threading.Thread(target=batch_worker)]
    for t_ in ts:
        t_.start()
    for t_ in ts:
        t_.join()
    assert not errors, errors[:3]
    assert "gave up waiting" not in capfd.readouterr().err


def test_whole_device_block_length_tunable_is_checked(oracle):
    """"sys_p" (penalties per hand-off block of the whole-device kernel) sizes the hand-off boxes and the traceback layout on the host;
    the product build has kernels for 8 only.  A value without a kernel must be refused when it is set — never laid out for and then
    run on the P = 8 kernel — and 8 itself keeps working."""
    eng = mw.Engine(0)
    built_all = True
    for p in (4, 16):
        try:
            eng.set("sys_p", p)
        except ValueError:
            built_all = False
    for bad in (0, 2, 3, 12, 32):
        with pytest.raises(ValueError):
            eng.set("sys_p", bad)
    t, q = synth_pair(99100, 12000, 0.06)
    for p in ((4, 8, 16) if built_all else (8,)):     # (a -DMWF_SYS_ALL_P build: every block length against the oracle)
        eng.set("sys_p", p)
        eng.set("force_kind", 1)
        for kw in (dict(), dict(flag=1), dict(flag=1, step=700)):
            b = eng.upload(PackedBatch([(t, q)]))
            b.align(mw.opt_init(**kw))
            s, it, nc = b.results()
            es, eit, ecig = oracle.align(t, q, make_opt(**kw))
            assert (int(s[0]), int(it[0])) == (es, eit), (p, kw)
            if ecig is not None:
                assert b.cigar(0, int(nc[0])).tolist() == ecig, (p, kw)
            b.free()
    eng.close()


def test_whole_device_kernel_columns_per_lane(oracle, capfd):
    """The systolic whole-device kernel with one and with four columns per lane (sys_c; chosen per pass from the expected window by
    default): same s, n_iter and CIGAR in every mode, two-pass low-memory mode included; and a pair whose window outgrows the
    64-column slots it was optimistically given comes back through the 256-column slots — not through the one-workgroup kernel."""
    cases = [synth_pair(99000, 9000, 0.08), synth_pair(99001, 30000, 0.03, 2, 2500), synth_pair(99002, 700, 0.2), (b"A" * 3000, b"A" * 2990)]
    opts = [make_opt(), make_opt(flag=1), make_opt(flag=1, step=600), make_opt(flag=1, o2=4, e2=2), make_opt(flag=1, x=1, o1=0, e1=1, o2=0, e2=1, step=150)]
    expect = {j: [oracle.align(t, q, o) for t, q in cases] for j, o in enumerate(opts)}
    for c, budget in ((1, 0), (4, 0), (1, 1)):          # (budget 1 MB: the two-pass low-memory mode, its second pass on the systolic kernel)
        eng = mw.Engine(0)
        eng.set("force_kind", 1)
        eng.set("sys_c", c)
        if budget:
            eng.set("lowmem_budget_mb", budget)
        for j, o in enumerate(opts):
            b = eng.upload(PackedBatch(cases))
            b.align(mw.opt_init(**{k: getattr(o, k) for k in OPT_KEYS}))
            assert eng.stats().kernel_kind == 1
            s, it, nc = b.results()
            for i in range(len(cases)):
                es, eit, ecig = expect[j][i]
                assert (int(s[i]), int(it[i])) == (es, eit), (c, budget, j, i)
                if ecig is not None:
                    assert b.cigar(i, int(nc[i])).tolist() == ecig, (c, budget, j, i)
            b.free()
        assert eng.stats().n_retries == 0
        eng.close()
    # the whole-device kernel confined to 16 workgroups (256 slots): 64-column slots hold windows up to 11.5 k columns, 256-column
    # ones up to 59.9 k; the expected window of a 20 kb pair (8192 columns at least) fits the former, the real one of this pair does not
    t, q = synth_pair(99010, 20000, 0.25)
    bands = oracle.band_trace(t, q, make_opt())                  # (one oracle run: penalty = slices, n_iter = sum of their widths)
    es, eit = len(bands), sum(h - l + 1 for l, h in bands)
    widest = max(h - l + 1 for l, h in bands)
    assert 11600 < widest < 59000, widest
    for aware in (0, 1):   # (1, the default: the batch's k-mer sketch sees the 25 % and the first pass takes the 256-column slots at once)
        eng = mw.Engine(0)
        eng.set("force_kind", 1)
        eng.set("coop_grid", 16)
        eng.set("div_aware", aware)
        b = eng.upload(PackedBatch([(t, q)]))
        b.align(mw.opt_init())
        s, it, _ = b.results()
        st = eng.stats()
        assert (int(s[0]), int(it[0])) == (es, eit)
        assert st.kernel_kind == 1 and st.n_retries == 1 - aware, (aware, st.kernel_kind, st.n_retries)
        assert "re-running it on one workgroup" not in capfd.readouterr().err
        b.free()
        eng.close()


def test_cached_plan_follows_every_tunable(oracle):
    """mwf_gpu_batch_align caches the plan of an align (size classes, order, per-class maxima) per batch.  The cache key is a generation
    count bumped by EVERY mwf_gpu_set(), not a hand-kept list of "the tunables that classify": flipping any tunable between two aligns
    of one batch must give the results of a fresh plan — the oracle's."""
    pairs = [synth_pair(97000 + i, (120, 300, 900, 2500, 6000)[i % 5], (0.03, 0.08)[i % 2]) for i in range(60)]
    exp = [oracle.align(t, q, make_opt(flag=1)) for t, q in pairs]
    eng = mw.Engine(0)
    b = eng.upload(PackedBatch(pairs))
    flips = [("lane_max_len", 0), ("lane_max_len", 325), ("mid_max_pairs", 0), ("mid_max_pairs", -1), ("seq2bit", 0), ("seq2bit", 1), ("band_pack", 0), ("band_pack", 1),
             ("force_kind", 0), ("force_kind", -1), ("block", 256), ("block", 0), ("ring16", 0), ("ring16", 1), ("lane_chunks", 2), ("lane_chunks", 0),
             ("mid_block", 512), ("mid_block", 0), ("band_span", 0), ("band_span", 1), ("wide_slots", 3), ("wide_slots", 0), ("host_results", 0), ("host_results", 1)]
    for name, value in [(None, 0)] + flips:
        if name:
            eng.set(name, value)
        b.align(mw.opt_init(flag=mw.MWF_F_CIGAR))
        s, it, nc = b.results()
        for i, (es, eit, ecig) in enumerate(exp):
            assert (int(s[i]), int(it[i])) == (es, eit) and b.cigar(i, int(nc[i])).tolist() == ecig, (name, value, i)
    b.free()
    eng.close()
