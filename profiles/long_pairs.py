#!/usr/bin/env python3
"""Time single long pairs on the whole-device kernel (BASELINE configs 2 and 4 stand-ins) and check the
size-independent properties: the CIGAR re-scores to s and consumes both sequences; s and CIGAR agree across modes."""
import json, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import miniwfa_amd as mw
from miniwfa_amd.synth import synth_pair, PackedBatch

def run(name, t, q, modes, cpu=False, cpu_kw=None):
    eng = mw.Engine(0)
    b = eng.upload(PackedBatch([(t, q)]))
    out = {"name": name, "tl": len(t), "ql": len(q)}
    ref = None
    for label, kw in modes:
        o = mw.opt_init(**kw)
        t0 = time.perf_counter()
        b.align(o); s, it, nc = b.results()
        wall = time.perf_counter() - t0
        st = eng.stats()
        rec = {"s": int(s[0]), "n_iter": int(it[0]), "wall_s": round(wall, 4), "kernel_ms": round(st.kernel_ms, 3), "kind": st.kernel_kind,
               "cells_pass1": int(st.cells_pass1), "gbp_s": round((len(t) + len(q)) / (st.kernel_ms * 1e-3) / 1e9, 6)}
        if kw.get("flag"):
            cig = b.cigar(0, int(nc[0]))
            sc, ctl, cql = mw.cigar2score(mw.opt_init(), cig.tolist())
            rec["n_cigar"] = int(nc[0]); rec["cigar_ok"] = (sc == int(s[0]) and ctl == len(t) and cql == len(q))
            rec_cigar = cig
            if ref is None: ref = cig
            else: rec["cigar_same_as_first_mode"] = bool(len(ref) == len(cig) and (ref == cig).all())
        out[label] = rec
        print(json.dumps({name: {label: rec}}), flush=True)
    if cpu:
        from oracle.pyoracle import Reference, make_opt
        if Reference.available():
            R = Reference(); t0 = time.perf_counter(); rs = R.align(t, q, make_opt()); dt = time.perf_counter() - t0
            out["cpu_reference_score_only"] = {"s": rs[0], "n_iter": rs[1], "wall_s": round(dt, 3)}
            print(json.dumps({name: {"cpu": out["cpu_reference_score_only"]}}), flush=True)
    if cpu_kw is not None:  # the compiled reference (oracle/_ref, one thread) in the same mode as the LAST GPU mode; results compared
        from oracle.pyoracle import Reference, make_opt
        R = Reference(arena=True); t0 = time.perf_counter(); rs = R.align(t, q, make_opt(**cpu_kw)); dt = time.perf_counter() - t0
        rec = {"allocator": "one kalloc arena for the process (km_init), not km = NULL", "mode": cpu_kw, "s": rs[0], "n_iter": rs[1], "wall_s": round(dt, 3), "threads": 1,
               "same_s_n_iter_as_gpu": (rs[0], rs[1]) == (int(s[0]), int(it[0]))}
        if rs[2] is not None:
            rec["same_cigar_as_gpu"] = bool(len(rs[2]) == len(rec_cigar) and (np.asarray(rs[2], dtype=np.uint32) == rec_cigar).all())
        out["cpu_reference"] = rec
        print(json.dumps({name: {"cpu_reference": rec}}), flush=True)
    b.free(); eng.close()
    return out

if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "c4"
    res = []
    if which in ("c4", "all"):
        t, q = synth_pair(2001, 150000, 0.035)
        res.append(run("c4_like_150kb", t, q, [("score", dict()), ("cigar", dict(flag=1)), ("cigar_p5000", dict(flag=1, step=5000))], cpu=True))
    if which in ("mhc", "all"):
        t, q = synth_pair(2002, 5000000, 0.008, 3, 15000)  # s ~ 230 k like GRCh38-vs-CHM13 MHC (README.md:86)
        res.append(run("mhc_like_5Mb", t, q, [("score", dict()), ("cigar_p5000", dict(flag=1, step=5000)), ("cigar", dict(flag=1))]))
    if which == "mhc_cpu":  # GPU low-memory mode, then the CPU reference in the same mode: minutes of one host core
        t, q = synth_pair(2002, 5000000, 0.008, 3, 15000)
        res.append(run("mhc_like_5Mb", t, q, [("score", dict()), ("cigar", dict(flag=1)), ("cigar_p5000", dict(flag=1, step=5000))], cpu_kw=dict(flag=1, step=5000)))
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(res, open(f"gpurun_out/long_pairs_{which}.json", "w"), indent=1)
