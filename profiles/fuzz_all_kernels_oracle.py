"""The fuzz pairs of synth.fuzz_pairs (granular lengths, repeats, unrelated pairs that fill the whole matrix) through the OTHER kernels,
against the oracle: generic kernel (one column per lane, four columns per lane with 32-bit and 16-bit ring rows, low-memory two-pass
mode), unpacked band kernel, whole-device (systolic) kernel with score, CIGAR and low-memory modes.  Usage: python profiles/fuzz_all_kernels_oracle.py [seed] [pairs]"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import miniwfa_amd as mw
from miniwfa_amd.synth import PackedBatch, fuzz_pairs
from oracle.pyoracle import Oracle, make_opt

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n_pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 120
pairs = fuzz_pairs(seed, n_pairs, 3000)
# two longer pairs (one of them unrelated): a batch whose longest pair exceeds 8 kb takes the generic kernel's wide form (512 threads, E2/F2 in LDS),
# which is the one that has 16-bit ring rows
_rng = np.random.default_rng(seed + 1000)
_acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
pairs.append((_acgt[_rng.integers(0, 4, 5200)].tobytes(), _acgt[_rng.integers(0, 4, 3900)].tobytes()))
_t = _acgt[_rng.integers(0, 4, 6000)]
_q = np.delete(_t.copy(), _rng.integers(0, 6000, 250)); _q[_rng.integers(0, len(_q), 200)] = _acgt[_rng.integers(0, 4, 200)]
pairs.append((_t.tobytes(), _q.tobytes()))
pk = PackedBatch(pairs)
orc = Oracle()
CONFIGS = [
    ("generic, one column per lane", dict(force_kind=0, scalar_generic=1)),
    ("generic, four columns per lane", dict(force_kind=0)),
    ("generic, 16-bit ring rows (packed recurrence)", dict(force_kind=0, ring16=2, expect_packed=16)),
    ("generic, 16-bit ring rows, 768 threads", dict(force_kind=0, ring16=2, ring16_block=768, expect_packed=16)),
    ("band, unpacked 256", dict(force_kind=2, block=256, band_pack=0)),
    ("band, unpacked 768", dict(force_kind=2, block=768, band_pack=0)),
    ("whole-device", dict(force_kind=1)),
]
bad = 0
for kw in (dict(), dict(flag=1), dict(flag=1, step=97)):
    o = make_opt(**kw)
    exp = [orc.align(t, q, o) for t, q in pairs]
    for name, sets in CONFIGS:
        if name == "whole-device": sub = [i for i in range(len(pairs)) if len(pairs[i][0]) + len(pairs[i][1]) > 600][:int(os.environ.get("FUZZ_WD_PAIRS", "12"))]   # (one launch per pair: a few by default)
        else: sub = list(range(len(pairs)))
        eng = mw.Engine(0)
        try:
            for k, v in sets.items():
                if k != "expect_packed": eng.set(k, v)
        except Exception as ex:
            print("   (tunable not accepted:", ex, ")")
        b = eng.upload(PackedBatch([pairs[i] for i in sub])); t0 = time.time()
        b.align(mw.opt_init(**kw)); s, it, nc = b.results()
        st = eng.stats(); n_bad = 0
        for j, i in enumerate(sub):
            es, eit, ecig = exp[i]
            ok = (int(s[j]), int(it[j])) == (es, eit) and (ecig is None or b.cigar(j, int(nc[j])).tolist() == ecig)
            if not ok:
                n_bad += 1
                if n_bad <= 3: print("   BAD pair", i, len(pairs[i][0]), len(pairs[i][1]), "got", int(s[j]), int(it[j]), "expected", es, eit, flush=True)
        bad += n_bad
        if "expect_packed" in sets and not kw.get("step") and st.packed != sets["expect_packed"]: print("   (NOT the kernel asked for: packed =", st.packed, ")")
        print(f"seed {seed} {kw} {name}: {len(sub)} pairs, mismatches {n_bad}, kernel kind {st.kernel_kind} packed {st.packed}, retries {st.n_retries}, {time.time() - t0:.1f} s", flush=True)
        b.free(); eng.close()
print("FUZZ", "FAILED" if bad else "OK", "seed", seed)
sys.exit(1 if bad else 0)
