"""ctypes bindings for the two CPU checkers (TEST INFRASTRUCTURE ONLY).

* ``Oracle``    — oracle/libmwf_oracle.so, our own restatement (oracle/mwf_oracle.c)
* ``Reference`` — oracle/_ref/libmwf_ref.so, the real lh3/miniwfa compiled by oracle/Makefile
                  (present only if it was built in a container that has /root/reference)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor
import time

HERE = os.path.dirname(os.path.abspath(__file__))

MWF_F_CIGAR = 0x1
MWF_F_NO_KALLOC = 0x2
CIGAR_CHARS = "MIDNSHP=XBid"  # main.c:78


class Opt(C.Structure):  # miniwfa.h:36-44
    _fields_ = [("flag", C.c_int32), ("x", C.c_int32), ("o1", C.c_int32), ("e1", C.c_int32),
                ("o2", C.c_int32), ("e2", C.c_int32), ("step", C.c_int32), ("max_s", C.c_int32),
                ("max_iter", C.c_int64), ("max_occ", C.c_int32), ("kmer", C.c_int32), ("min_len", C.c_int32)]


class Rst(C.Structure):  # miniwfa.h:46-51
    _fields_ = [("s", C.c_int32), ("n_cigar", C.c_int32), ("n_iter", C.c_int64), ("cigar", C.POINTER(C.c_uint32))]


class Stat(C.Structure):
    _fields_ = [("cells_pass1", C.c_int64), ("n_seg", C.c_int32)]


class Chkpt(C.Structure):
    _fields_ = [("s", C.c_int32), ("d", C.c_int32)]


assert C.sizeof(Opt) == 56 and C.sizeof(Rst) == 24


def make_opt(flag=0, x=4, o1=4, e1=2, o2=15, e2=1, step=0, max_s=0, max_iter=0, max_occ=2, kmer=13, min_len=30) -> Opt:
    return Opt(flag, x, o1, e1, o2, e2, step, max_s, max_iter, max_occ, kmer, min_len)


def cigar_str(words) -> str:
    return "".join(f"{w >> 4}{CIGAR_CHARS[w & 0xf]}" for w in words)


def build(force: bool = False) -> None:
    """make -C oracle (restatement always; reference only when /root/reference exists)."""
    so = os.path.join(HERE, "libmwf_oracle.so")
    src = os.path.join(HERE, "mwf_oracle.c")
    need = force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src)
    ref_so = os.path.join(HERE, "_ref", "libmwf_ref.so")
    if os.path.exists("/root/reference/miniwfa.c") and not all(os.path.exists(os.path.join(HERE, "_ref", f)) for f in ("libmwf_ref.so", "libmwf_ref_v3.so", "libmwf_ref_v4.so")):
        need = True
    if need:
        subprocess.run(["make", "-C", HERE], check=True, capture_output=True)


class _Aligner:
    """Shared result handling.  Subclasses set self._exact(opt_ptr, tl, ts, ql, qs, rst_ptr) and self._free."""

    def align(self, t: bytes, q: bytes, opt: Opt):
        r = Rst()
        self._exact(C.byref(opt), len(t), t, len(q), q, C.byref(r))
        cig = None
        if r.cigar:
            cig = [r.cigar[i] for i in range(r.n_cigar)]
            self._free(r.cigar)
        return r.s, r.n_iter, cig

    def align_many(self, pairs, opt: Opt, threads: int = 1):
        """[(s, n_iter, cigar)], wall seconds.  ctypes drops the GIL inside the call."""
        t0 = time.perf_counter()
        if threads <= 1:
            out = [self.align(t, q, opt) for t, q in pairs]
        else:
            with ThreadPoolExecutor(threads) as ex:
                out = list(ex.map(lambda tq: self.align(tq[0], tq[1], opt), pairs))
        return out, time.perf_counter() - t0


class Oracle(_Aligner):
    def __init__(self):
        build()
        L = C.CDLL(os.path.join(HERE, "libmwf_oracle.so"))
        self.lib = L
        L.mwfo_exact.argtypes = [C.POINTER(Opt), C.c_int32, C.c_char_p, C.c_int32, C.c_char_p, C.POINTER(Rst)]
        L.mwfo_exact.restype = None
        L.mwfo_exact_stat.argtypes = L.mwfo_exact.argtypes + [C.POINTER(Stat)]
        L.mwfo_exact_stat.restype = None
        L.mwfo_auto_exact_branch.argtypes = L.mwfo_exact.argtypes
        L.mwfo_auto_exact_branch.restype = None
        L.mwfo_checkpoints.argtypes = [C.POINTER(Opt), C.c_int32, C.c_char_p, C.c_int32, C.c_char_p, C.POINTER(C.POINTER(Chkpt))]
        L.mwfo_checkpoints.restype = C.c_int32
        L.mwfo_cigar2score.argtypes = [C.POINTER(Opt), C.c_int32, C.POINTER(C.c_uint32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.mwfo_cigar2score.restype = C.c_int32
        L.mwfo_band_trace.argtypes = [C.POINTER(Opt), C.c_int32, C.c_char_p, C.c_int32, C.c_char_p, C.POINTER(C.c_int32), C.c_int32]
        L.mwfo_band_trace.restype = C.c_int32
        L.mwfo_free.argtypes = [C.c_void_p]
        L.mwfo_free.restype = None
        L.mwfo_opt_init.argtypes = [C.POINTER(Opt)]
        L.mwfo_batch.argtypes = [C.POINTER(Opt), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_int32, C.c_void_p, C.c_void_p]
        L.mwfo_batch.restype = C.c_double
        L.mwfo_batch_with.argtypes = [C.c_void_p] + L.mwfo_batch.argtypes
        L.mwfo_batch_with.restype = C.c_double
        L.mwfo_batch_arena.argtypes = [C.c_void_p] * 4 + L.mwfo_batch.argtypes
        L.mwfo_batch_arena.restype = C.c_double
        self._exact = L.mwfo_exact
        self._free = lambda p: L.mwfo_free(C.cast(p, C.c_void_p))

    def opt_init(self) -> Opt:
        o = Opt()
        self.lib.mwfo_opt_init(C.byref(o))
        return o

    def align_stat(self, t: bytes, q: bytes, opt: Opt):
        r, st = Rst(), Stat()
        self.lib.mwfo_exact_stat(C.byref(opt), len(t), t, len(q), q, C.byref(r), C.byref(st))
        cig = None
        if r.cigar:
            cig = [r.cigar[i] for i in range(r.n_cigar)]
            self._free(r.cigar)
        return r.s, r.n_iter, cig, st.cells_pass1, st.n_seg

    def auto_exact_branch(self, t: bytes, q: bytes, opt: Opt):
        r = Rst()
        self.lib.mwfo_auto_exact_branch(C.byref(opt), len(t), t, len(q), q, C.byref(r))
        cig = None
        if r.cigar:
            cig = [r.cigar[i] for i in range(r.n_cigar)]
            self._free(r.cigar)
        return r.s, r.n_iter, cig

    def checkpoints(self, t: bytes, q: bytes, opt: Opt):
        seg = C.POINTER(Chkpt)()
        n = self.lib.mwfo_checkpoints(C.byref(opt), len(t), t, len(q), q, C.byref(seg))
        out = [(seg[i].s, seg[i].d) for i in range(n)]
        self.lib.mwfo_free(C.cast(seg, C.c_void_p))
        return out

    def cigar2score(self, opt: Opt, cigar):
        arr = (C.c_uint32 * max(1, len(cigar)))(*cigar)
        tl, ql = C.c_int32(), C.c_int32()
        s = self.lib.mwfo_cigar2score(C.byref(opt), len(cigar), arr, C.byref(tl), C.byref(ql))
        return s, tl.value, ql.value

    def band_trace(self, t: bytes, q: bytes, opt: Opt, cap: int = 1 << 20):
        buf = (C.c_int32 * (2 * cap))()
        n = self.lib.mwfo_band_trace(C.byref(opt), len(t), t, len(q), q, buf, cap)
        return [(buf[2 * i], buf[2 * i + 1]) for i in range(min(n, cap))]

    def batch(self, packed, opt: Opt, threads: int, exact_fn=None, n=None, arena=None):
        """Threaded (pthreads, one pair per thread at a time) batch over the first n pairs of a miniwfa_amd.synth.PackedBatch;
        returns (s[], n_iter[], wall seconds).  exact_fn: address of a function with mwf_wfa_exact's signature to time instead
        of the restatement (e.g. Reference().exact_addr()); arena: (km_init, km_destroy, kfree) addresses of the same library
        — every worker thread then aligns inside a private kalloc arena (Reference().arena_addrs())."""
        import numpy as np
        n = packed.n if n is None else min(n, packed.n)
        s = np.zeros(n, dtype=np.int32)
        it = np.zeros(n, dtype=np.int64)
        a = arena if (arena and exact_fn) else (None, None, None)
        sec = self.lib.mwfo_batch_arena(exact_fn, a[0], a[1], a[2], C.byref(opt), n, packed.seqs.ctypes.data, packed.t_off.ctypes.data,
                                        packed.tl.ctypes.data, packed.q_off.ctypes.data, packed.ql.ctypes.data,
                                        threads, s.ctypes.data, it.ctypes.data)
        return s, it, sec


class Reference(_Aligner):
    """The real lh3/miniwfa (mwf_wfa_exact, miniwfa.c:603) — km=NULL, i.e. libc malloc."""

    path = os.path.join(HERE, "_ref", "libmwf_ref.so")

    @classmethod
    def available(cls) -> bool:
        if not os.path.exists(cls.path):
            try:
                build()
            except Exception:
                return False
        return os.path.exists(cls.path)

    # the builds oracle/Makefile makes of the same three reference sources: flags, what the host must support (/proc/cpuinfo flags)
    VARIANTS = {"sse4.2": ("libmwf_ref.so", "-O3 -msse4.2", ("sse4_2",)),
                "v3": ("libmwf_ref_v3.so", "-O3 -march=x86-64-v3", ("avx2", "bmi2", "fma")),
                "v4": ("libmwf_ref_v4.so", "-O3 -march=x86-64-v4", ("avx512f", "avx512bw", "avx512cd", "avx512dq", "avx512vl"))}

    @classmethod
    def variant_usable(cls, variant: str) -> bool:
        """The build exists AND this host's CPU has every instruction-set extension it was compiled for."""
        so, _, need = cls.VARIANTS[variant]
        if not os.path.exists(os.path.join(HERE, "_ref", so)):
            return False
        try:
            flags = set()
            for line in open("/proc/cpuinfo"):
                if line.startswith("flags"):
                    flags = set(line.split(":", 1)[1].split())
                    break
            return all(f in flags for f in need)
        except OSError:
            return variant == "sse4.2"

    def __init__(self, arena: bool = False, variant: str = "sse4.2"):
        """arena=True: every call allocates inside one kalloc arena owned by this object (km_init) instead of km = NULL —
        what a long-running caller of the library would do; saves the page faults of a fresh arena per call.
        variant: which build of the reference (VARIANTS); the default is the README's recommendation."""
        if not self.available():
            raise FileNotFoundError(self.path)
        self.variant, self.flags = variant, self.VARIANTS[variant][1]
        L = C.CDLL(os.path.join(HERE, "_ref", self.VARIANTS[variant][0]))
        self.lib = L
        sig = [C.c_void_p, C.POINTER(Opt), C.c_int32, C.c_char_p, C.c_int32, C.c_char_p, C.POINTER(Rst)]
        for name in ("mwf_wfa_exact", "mwf_wfa_auto", "mwf_wfa_chain"):
            getattr(L, name).argtypes = sig
            getattr(L, name).restype = None
        L.mwf_opt_init.argtypes = [C.POINTER(Opt)]
        L.mwf_cigar2score.argtypes = [C.POINTER(Opt), C.c_int32, C.POINTER(C.c_uint32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.mwf_cigar2score.restype = C.c_int32
        self._libc = C.CDLL(None)
        self._libc.free.argtypes = [C.c_void_p]
        L.km_init.restype = C.c_void_p
        L.kfree.argtypes = [C.c_void_p, C.c_void_p]
        self._km = C.c_void_p(L.km_init()) if arena else None
        self._exact = lambda o, tl, t, ql, q, r: L.mwf_wfa_exact(self._km, o, tl, t, ql, q, r)
        if arena:
            self._free = lambda p: L.kfree(self._km, C.cast(p, C.c_void_p))
        else:
            self._free = lambda p: self._libc.free(C.cast(p, C.c_void_p))

    def opt_init(self) -> Opt:
        o = Opt()
        self.lib.mwf_opt_init(C.byref(o))
        return o

    def exact_addr(self) -> int:
        """Address of the reference's mwf_wfa_exact, for Oracle.batch(exact_fn=...)."""
        return C.cast(self.lib.mwf_wfa_exact, C.c_void_p).value

    def arena_addrs(self):
        """(km_init, km_destroy, kfree) of the reference's kalloc, for Oracle.batch(arena=...)."""
        return tuple(C.cast(getattr(self.lib, n), C.c_void_p).value for n in ("km_init", "km_destroy", "kfree"))

    def _call(self, fn, t, q, opt):
        r = Rst()
        fn(self._km, C.byref(opt), len(t), t, len(q), q, C.byref(r))
        cig = None
        if r.cigar:
            cig = [r.cigar[i] for i in range(r.n_cigar)]
            self._free(r.cigar)
        return r.s, r.n_iter, cig

    def auto(self, t, q, opt):
        return self._call(self.lib.mwf_wfa_auto, t, q, opt)

    def chain(self, t, q, opt):
        return self._call(self.lib.mwf_wfa_chain, t, q, opt)
