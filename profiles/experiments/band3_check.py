"""The packed band kernel with its 2-bit sequence copy, and the balanced band kernel (mwf_band3.hip), against the packed band kernel
on byte-wise sequences, same batches: results must be identical (s, n_iter, CIGAR); prints kernel times.  Usage: python profiles/band3_check.py [quick]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import miniwfa_amd as mw
from miniwfa_amd.synth import synth_pair, PackedBatch

def run(pk, flag, mode, reps=2, **tun):
    eng = mw.Engine(0)
    eng.set("seq2bit", 0 if mode == "bytes" else 1)
    eng.set("band3", 1 if mode == "band3" else 0)
    for k, v in tun.items(): eng.set(k, v)
    b = eng.upload(pk)
    o = mw.opt_init(flag=flag)
    for _ in range(reps):
        b.align(o); s, it, nc = b.results()
    st = eng.stats()
    cig = [b.cigar(i, int(nc[i])).tolist() for i in range(pk.n)] if flag else None
    out = (np.array(s), np.array(it), cig, st.kernel_ms, st.packed, st.block, st.n_retries)
    b.free(); eng.close()
    return out

def compare(name, pairs, flags=(0, mw.MWF_F_CIGAR)):
    pk = PackedBatch(pairs)
    for flag in flags:
        a = run(pk, flag, "bytes")
        for mode in ("2bit", "band3"):
            c = run(pk, flag, mode)
            ok = (a[0] == c[0]).all() and (a[1] == c[1]).all() and a[2] == c[2]
            print(f"{name} flag={flag}: bytes {a[3]:.3f} ms (packed {a[4]} block {a[5]} retries {a[6]}) | {mode} {c[3]:.3f} ms (packed {c[4]} block {c[5]} retries {c[6]}) | identical {ok}", flush=True)
            if not ok:
                bad = [i for i in range(pk.n) if a[0][i] != c[0][i] or a[1][i] != c[1][i] or (a[2] and a[2][i] != c[2][i])]
                print("   first differing pairs:", bad[:8], [(int(a[0][i]), int(c[0][i]), int(a[1][i]), int(c[1][i])) for i in bad[:4]])

quick = len(sys.argv) > 1
compare("2000 x 150bp", [synth_pair(50 + i, 150, 0.05) for i in range(2000)])
compare("8 x 3kb", [synth_pair(100 + i, 3000, 0.05) for i in range(8)])
compare("64 x 10kb", [synth_pair(1000 + i, 10000, 0.05) for i in range(64)])
compare("ragged 4-9kb, 2-8%", [synth_pair(2000 + i, 4000 + 700 * (i % 8), 0.02 + 0.01 * (i % 7)) for i in range(48)])
mixed = [synth_pair(3000 + i, 6000, 0.04) for i in range(16)]
mixed[3] = (mixed[3][0].replace(b"A", b"N", 3), mixed[3][1])
mixed[7] = (mixed[7][0].lower(), mixed[7][1].lower())
compare("with non-ACGT pairs", mixed)
if not quick:
    compare("1024 x 10kb", [synth_pair(4200 + i, 10000, 0.05) for i in range(1024)])
