"""Python mirror of the miniwfa interface over the C ABI of libmwf_hip.so (include/miniwfa.h).

Names, argument meaning and error behaviour follow the reference API (miniwfa.h:36-89):
``opt_init()`` -> mwf_opt_init, ``wfa_exact`` -> mwf_wfa_exact, ``wfa_auto`` -> mwf_wfa_auto,
``cigar2score`` -> mwf_cigar2score.  ``Engine``/``Batch`` wrap the device-resident part of the ABI.

This module only marshals arguments.  All alignment work happens in the HIP kernels behind the
C ABI; if the shared library is missing or no GPU can be opened, calls raise — there is no
Python or CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os
import weakref
from typing import Iterable, Sequence

import numpy as np

from . import build as _build

MWF_F_CIGAR = 0x1
MWF_F_NO_KALLOC = 0x2
MWF_F_DEBUG = 0x10000
CIGAR_CHARS = "MIDNSHP=XBid"  # reference main.c:78


class MwfOpt(C.Structure):
    """mwf_opt_t (reference miniwfa.h:36-44), 56 bytes."""
    _fields_ = [("flag", C.c_int32), ("x", C.c_int32), ("o1", C.c_int32), ("e1", C.c_int32),
                ("o2", C.c_int32), ("e2", C.c_int32), ("step", C.c_int32), ("max_s", C.c_int32),
                ("max_iter", C.c_int64), ("max_occ", C.c_int32), ("kmer", C.c_int32), ("min_len", C.c_int32)]


class MwfRst(C.Structure):
    """mwf_rst_t (reference miniwfa.h:46-51), 24 bytes."""
    _fields_ = [("s", C.c_int32), ("n_cigar", C.c_int32), ("n_iter", C.c_int64), ("cigar", C.POINTER(C.c_uint32))]


class GpuStats(C.Structure):
    _fields_ = [("kernel_ms", C.c_double), ("cells", C.c_int64), ("cells_pass1", C.c_int64), ("n_launches", C.c_int32),
                ("n_retries", C.c_int32), ("grid", C.c_int32), ("block", C.c_int32), ("kernel_kind", C.c_int32),
                ("dev_bytes", C.c_int64), ("dev_bytes_peak", C.c_int64), ("packed", C.c_int32), ("lowmem_two_pass", C.c_int32)]


class KmStat(C.Structure):
    _fields_ = [("capacity", C.c_size_t), ("available", C.c_size_t), ("n_blocks", C.c_size_t),
                ("n_cores", C.c_size_t), ("largest", C.c_size_t)]


# every symbol include/miniwfa.h and include/kalloc.h declare
ABI_SYMBOLS = (
    "mwf_opt_init", "mwf_wfa_exact", "mwf_wfa_auto", "mwf_wfa_chain", "mwf_cigar2score", "mwf_assert_cigar",
    "mwf_wfa_batch", "mwf_wfa_batch_multi", "mwf_wfa_chain_batch", "mwf_wfa_auto_batch", "mwf_wfa_submit", "mwf_wfa_wait", "mwf_wfa_async_stats", "mwf_gpu_batch_dev_status", "mwf_gpu_batch_fetch_cigars", "mwf_gpu_device_count", "mwf_gpu_create", "mwf_gpu_destroy", "mwf_gpu_last_error",
    "mwf_gpu_batch_upload", "mwf_gpu_batch_wrap", "mwf_gpu_batch_free", "mwf_gpu_batch_align", "mwf_gpu_batch_results",
    "mwf_gpu_batch_dev_scores", "mwf_gpu_batch_dev_iters", "mwf_gpu_batch_cigar", "mwf_gpu_get_stats", "mwf_gpu_set",
    "mwf_gpu_debug_band",
    "kmalloc", "kcalloc", "krealloc", "krelocate", "kfree", "km_init", "km_init2", "km_destroy", "km_stat", "km_stat_print",
)

_lib = None


def lib() -> C.CDLL:
    """The C-ABI library (built on demand by hipcc; raises if that is impossible)."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("MWF_HIP_LIB")   # experiments only (e.g. an instrumented build); default: the in-tree library
    if not path:
        path = _build.LIB
        if not os.path.exists(path) or _build.stale():
            path = _build.build()
    L = C.CDLL(path)
    P = C.POINTER
    sig = [C.c_void_p, P(MwfOpt), C.c_int32, C.c_char_p, C.c_int32, C.c_char_p, P(MwfRst)]
    for name in ("mwf_wfa_exact", "mwf_wfa_auto", "mwf_wfa_chain"):
        getattr(L, name).argtypes = sig
        getattr(L, name).restype = None
    L.mwf_opt_init.argtypes = [P(MwfOpt)]
    L.mwf_opt_init.restype = None
    L.mwf_cigar2score.argtypes = [P(MwfOpt), C.c_int32, P(C.c_uint32), P(C.c_int32), P(C.c_int32)]
    L.mwf_cigar2score.restype = C.c_int32
    L.mwf_assert_cigar.argtypes = [P(MwfOpt), C.c_int32, P(C.c_uint32), C.c_int32, C.c_int32, C.c_int32]
    L.mwf_assert_cigar.restype = None
    L.mwf_wfa_batch.argtypes = [C.c_void_p, P(MwfOpt), C.c_int32, P(C.c_int32), P(C.c_char_p), P(C.c_int32), P(C.c_char_p), P(MwfRst)]
    L.mwf_wfa_batch.restype = None
    L.mwf_wfa_batch_multi.argtypes = L.mwf_wfa_batch.argtypes + [C.c_int32, P(C.c_int32)]
    L.mwf_wfa_batch_multi.restype = None
    L.mwf_wfa_chain_batch.argtypes = L.mwf_wfa_batch.argtypes
    L.mwf_wfa_chain_batch.restype = None
    L.mwf_wfa_auto_batch.argtypes = L.mwf_wfa_batch.argtypes
    L.mwf_wfa_auto_batch.restype = None
    L.mwf_wfa_submit.argtypes = [P(MwfOpt), C.c_int32, C.c_char_p, C.c_int32, C.c_char_p]
    L.mwf_wfa_submit.restype = C.c_void_p
    L.mwf_wfa_wait.argtypes = [C.c_void_p, C.c_void_p, P(MwfRst)]
    L.mwf_wfa_wait.restype = None
    L.mwf_wfa_async_stats.argtypes = [P(C.c_int64), P(C.c_int64)]
    L.mwf_wfa_async_stats.restype = None
    L.mwf_gpu_batch_dev_status.argtypes = [C.c_void_p]
    L.mwf_gpu_batch_dev_status.restype = C.c_void_p
    L.mwf_gpu_batch_fetch_cigars.argtypes = [C.c_void_p, C.c_void_p]
    L.mwf_gpu_batch_fetch_cigars.restype = C.c_int
    L.mwf_gpu_device_count.restype = C.c_int
    L.mwf_gpu_create.argtypes = [C.c_int, C.c_void_p]
    L.mwf_gpu_create.restype = C.c_void_p
    L.mwf_gpu_destroy.argtypes = [C.c_void_p]
    L.mwf_gpu_destroy.restype = None
    L.mwf_gpu_last_error.argtypes = [C.c_void_p]
    L.mwf_gpu_last_error.restype = C.c_char_p
    L.mwf_gpu_batch_upload.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.mwf_gpu_batch_upload.restype = C.c_void_p
    L.mwf_gpu_batch_wrap.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.mwf_gpu_batch_wrap.restype = C.c_void_p
    L.mwf_gpu_batch_free.argtypes = [C.c_void_p]
    L.mwf_gpu_batch_free.restype = None
    L.mwf_gpu_batch_align.argtypes = [C.c_void_p, C.c_void_p, P(MwfOpt)]
    L.mwf_gpu_batch_align.restype = C.c_int
    L.mwf_gpu_batch_results.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.mwf_gpu_batch_results.restype = C.c_int
    L.mwf_gpu_batch_dev_scores.argtypes = [C.c_void_p]
    L.mwf_gpu_batch_dev_scores.restype = C.c_void_p
    L.mwf_gpu_batch_dev_iters.argtypes = [C.c_void_p]
    L.mwf_gpu_batch_dev_iters.restype = C.c_void_p
    L.mwf_gpu_batch_cigar.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32]
    L.mwf_gpu_batch_cigar.restype = C.c_int32
    L.mwf_gpu_get_stats.argtypes = [C.c_void_p, P(GpuStats)]
    L.mwf_gpu_get_stats.restype = None
    L.mwf_gpu_set.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
    L.mwf_gpu_set.restype = C.c_int
    L.mwf_gpu_test_hook.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
    L.mwf_gpu_test_hook.restype = C.c_int
    L.mwf_gpu_debug_band.argtypes = [C.c_void_p, C.c_void_p, P(MwfOpt), C.c_int32, C.c_void_p, C.c_int32]
    L.mwf_gpu_debug_band.restype = C.c_int32
    for name, res, args in (("kmalloc", C.c_void_p, [C.c_void_p, C.c_size_t]), ("kcalloc", C.c_void_p, [C.c_void_p, C.c_size_t, C.c_size_t]),
                            ("krealloc", C.c_void_p, [C.c_void_p, C.c_void_p, C.c_size_t]), ("krelocate", C.c_void_p, [C.c_void_p, C.c_void_p, C.c_size_t]),
                            ("kfree", None, [C.c_void_p, C.c_void_p]), ("km_init", C.c_void_p, []), ("km_init2", C.c_void_p, [C.c_void_p, C.c_size_t]),
                            ("km_destroy", None, [C.c_void_p]), ("km_stat", None, [C.c_void_p, P(KmStat)]), ("km_stat_print", None, [C.c_void_p])):
        getattr(L, name).restype = res
        getattr(L, name).argtypes = args
    _lib = L
    return L


def opt_init(**overrides) -> MwfOpt:
    """mwf_opt_init() (reference miniwfa.c:11-18) plus keyword overrides of individual fields."""
    o = MwfOpt()
    lib().mwf_opt_init(C.byref(o))
    for k, v in overrides.items():
        setattr(o, k, v)
    return o


def cigar_str(words: Iterable[int]) -> str:
    """CIGAR words (len<<4|op) as text, the way reference main.c:76-79 prints them."""
    return "".join(f"{int(w) >> 4}{CIGAR_CHARS[int(w) & 0xf]}" for w in words)


def _take(r: MwfRst, km):
    cig = None
    if r.cigar:
        cig = [r.cigar[i] for i in range(r.n_cigar)]
        lib().kfree(km, C.cast(r.cigar, C.c_void_p))
    return r.s, r.n_iter, cig


def wfa_exact(t: bytes, q: bytes, opt: MwfOpt, km=None):
    """mwf_wfa_exact (reference miniwfa.c:603-615) -> (s, n_iter, cigar words or None)."""
    r = MwfRst()
    lib().mwf_wfa_exact(km, C.byref(opt), len(t), t, len(q), q, C.byref(r))
    return _take(r, km)


def wfa_auto(t: bytes, q: bytes, opt: MwfOpt, km=None):
    """mwf_wfa_auto (reference miniwfa.c:898-908)."""
    r = MwfRst()
    lib().mwf_wfa_auto(km, C.byref(opt), len(t), t, len(q), q, C.byref(r))
    return _take(r, km)


def wfa_chain(t: bytes, q: bytes, opt: MwfOpt, km=None):
    """mwf_wfa_chain (reference miniwfa.c:850-896)."""
    r = MwfRst()
    lib().mwf_wfa_chain(km, C.byref(opt), len(t), t, len(q), q, C.byref(r))
    return _take(r, km)


def wfa_chain_batch(pairs: Sequence[tuple[bytes, bytes]], opt: MwfOpt, km=None):
    """mwf_wfa_chain_batch: mwf_wfa_chain of every pair, their gap fills in one device batch -> list of (s, n_iter, cigar)."""
    n = len(pairs)
    if n == 0:
        return []
    tl = (C.c_int32 * n)(*[len(t) for t, _ in pairs])
    ql = (C.c_int32 * n)(*[len(q) for _, q in pairs])
    ts = (C.c_char_p * n)(*[t for t, _ in pairs])
    qs = (C.c_char_p * n)(*[q for _, q in pairs])
    r = (MwfRst * n)()
    lib().mwf_wfa_chain_batch(km, C.byref(opt), n, tl, ts, ql, qs, r)
    return [_take(r[i], km) for i in range(n)]


def wfa_auto_batch(pairs: Sequence[tuple[bytes, bytes]], opt: MwfOpt, km=None):
    """mwf_wfa_auto_batch: mwf_wfa_auto of every pair (exact branch as one batch, chain mode for what it gives up on) -> list of (s, n_iter, cigar)."""
    n = len(pairs)
    if n == 0:
        return []
    tl = (C.c_int32 * n)(*[len(t) for t, _ in pairs])
    ql = (C.c_int32 * n)(*[len(q) for _, q in pairs])
    ts = (C.c_char_p * n)(*[t for t, _ in pairs])
    qs = (C.c_char_p * n)(*[q for _, q in pairs])
    r = (MwfRst * n)()
    lib().mwf_wfa_auto_batch(km, C.byref(opt), n, tl, ts, ql, qs, r)
    return [_take(r[i], km) for i in range(n)]


class Job:
    """A pair submitted with mwf_wfa_submit; keeps the buffers it borrows alive until wait()."""

    def __init__(self, t: bytes, q: bytes, opt: MwfOpt):
        self._keep = (t, q, opt)
        self.h = lib().mwf_wfa_submit(C.byref(opt), len(t), t, len(q), q)
        if not self.h:
            raise ValueError("mwf_wfa_submit refused the pair")

    def wait(self, km=None):
        """mwf_wfa_wait -> (s, n_iter, cigar words or None), exactly what wfa_exact returns for the pair."""
        r = MwfRst()
        lib().mwf_wfa_wait(km, self.h, C.byref(r))
        self.h = None
        return _take(r, km)


def wfa_submit(t: bytes, q: bytes, opt: MwfOpt) -> Job:
    """mwf_wfa_submit: queue one pair for the dispatcher (include/miniwfa.h part 2); job.wait() collects it."""
    return Job(t, q, opt)


def async_stats():
    """(batches the dispatcher has run, pairs in them)."""
    a, b = C.c_int64(), C.c_int64()
    lib().mwf_wfa_async_stats(C.byref(a), C.byref(b))
    return a.value, b.value


def wfa_batch(pairs: Sequence[tuple[bytes, bytes]], opt: MwfOpt, km=None):
    """mwf_wfa_batch over host buffers -> list of (s, n_iter, cigar)."""
    n = len(pairs)
    if n == 0:
        return []
    tl = (C.c_int32 * n)(*[len(t) for t, _ in pairs])
    ql = (C.c_int32 * n)(*[len(q) for _, q in pairs])
    ts = (C.c_char_p * n)(*[t for t, _ in pairs])
    qs = (C.c_char_p * n)(*[q for _, q in pairs])
    r = (MwfRst * n)()
    lib().mwf_wfa_batch(km, C.byref(opt), n, tl, ts, ql, qs, r)
    return [_take(r[i], km) for i in range(n)]


def wfa_batch_multi(pairs: Sequence[tuple[bytes, bytes]], opt: MwfOpt, devices: Sequence[int] | None = None, n_dev: int = 0, km=None):
    """mwf_wfa_batch_multi: the batch dealt over several devices (an ordinal may repeat) -> list of (s, n_iter, cigar)."""
    n = len(pairs)
    if n == 0:
        return []
    tl = (C.c_int32 * n)(*[len(t) for t, _ in pairs])
    ql = (C.c_int32 * n)(*[len(q) for _, q in pairs])
    ts = (C.c_char_p * n)(*[t for t, _ in pairs])
    qs = (C.c_char_p * n)(*[q for _, q in pairs])
    r = (MwfRst * n)()
    dv = (C.c_int32 * len(devices))(*devices) if devices else None
    lib().mwf_wfa_batch_multi(km, C.byref(opt), n, tl, ts, ql, qs, r, len(devices) if devices else n_dev, dv)
    return [_take(r[i], km) for i in range(n)]


def cigar2score(opt: MwfOpt, cigar: Sequence[int]):
    """mwf_cigar2score (reference mwf-dbg.c:6-22) -> (score, target length, query length)."""
    arr = (C.c_uint32 * max(1, len(cigar)))(*cigar)
    tl, ql = C.c_int32(), C.c_int32()
    s = lib().mwf_cigar2score(C.byref(opt), len(cigar), arr, C.byref(tl), C.byref(ql))
    return s, tl.value, ql.value


class Engine:
    """mwf_gpu_t: one device, one stream, one workspace pool."""

    def __init__(self, device: int = 0, stream: int | None = None):
        L = lib()
        if L.mwf_gpu_device_count() <= 0:
            raise RuntimeError("libmwf_hip: no HIP device visible; the library has no CPU path")
        self.h = L.mwf_gpu_create(device, stream)
        if not self.h:
            raise RuntimeError(f"libmwf_hip: cannot open device {device}")
        self.device = device
        self._batches = weakref.WeakSet()

    def close(self):
        if self.h:
            for b in list(self._batches):  # a batch must not outlive its engine
                b.free()
            lib().mwf_gpu_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def error(self) -> str:
        return lib().mwf_gpu_last_error(self.h).decode()

    def set(self, name: str, value: int):
        """A user tunable (mwf_gpu_set, include/miniwfa.h) or — for tests and profiling scripts — one of the library's test hooks
        (mwf_gpu_test_hook: forced kernels / geometries / failure paths; exported, not part of the public header)."""
        if lib().mwf_gpu_set(self.h, name.encode(), int(value)) != 0 and lib().mwf_gpu_test_hook(self.h, name.encode(), int(value)) != 0:
            raise ValueError(f"bad tunable {name}={value}")

    def stats(self) -> GpuStats:
        st = GpuStats()
        lib().mwf_gpu_get_stats(self.h, C.byref(st))
        return st

    def upload(self, packed) -> "Batch":
        """Batch from a miniwfa_amd.synth.PackedBatch living in host memory."""
        h = lib().mwf_gpu_batch_upload(self.h, packed.n, packed.seqs.ctypes.data, packed.total, packed.t_off.ctypes.data,
                                       packed.tl.ctypes.data, packed.q_off.ctypes.data, packed.ql.ctypes.data)
        if not h:
            raise RuntimeError("upload failed: " + self.error())
        return Batch(self, h, packed.n, keep=(packed,))

    def wrap(self, n, d_seqs_ptr, seq_bytes, d_t_off_ptr, d_tl_ptr, d_q_off_ptr, d_ql_ptr, h_tl: np.ndarray, h_ql: np.ndarray, keep=()) -> "Batch":
        """Batch over buffers that already live in this device's HBM (e.g. torch tensors' data_ptr())."""
        h_tl = np.ascontiguousarray(h_tl, dtype=np.int32)
        h_ql = np.ascontiguousarray(h_ql, dtype=np.int32)
        h = lib().mwf_gpu_batch_wrap(self.h, n, d_seqs_ptr, seq_bytes, d_t_off_ptr, d_tl_ptr, d_q_off_ptr, d_ql_ptr,
                                     h_tl.ctypes.data, h_ql.ctypes.data)
        if not h:
            raise RuntimeError("wrap failed: " + self.error())
        return Batch(self, h, n, keep=tuple(keep))

    def wrap_packed(self, packed, torch_device) -> "Batch":
        """A miniwfa_amd.synth.PackedBatch copied into torch tensors on `torch_device` and wrapped zero-copy: the device-resident path a
        torch program uses — the library never sees the bytes on the host."""
        import torch
        t = [torch.from_numpy(np.array(a, copy=True)).to(torch_device) for a in (packed.seqs, packed.t_off, packed.tl, packed.q_off, packed.ql)]
        torch.cuda.synchronize(torch_device)
        return self.wrap(packed.n, t[0].data_ptr(), packed.total, t[1].data_ptr(), t[2].data_ptr(), t[3].data_ptr(), t[4].data_ptr(),
                         packed.tl, packed.ql, keep=tuple(t))


class Batch:
    """mwf_gpu_batch_t."""

    def __init__(self, eng: Engine, handle, n: int, keep=()):
        self.eng, self.h, self.n, self._keep = eng, handle, n, keep
        eng._batches.add(self)

    def free(self):
        if self.h and self.eng.h:
            lib().mwf_gpu_batch_free(self.h)
        self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def align(self, opt: MwfOpt):
        """Enqueue the alignment kernels on the engine's stream."""
        rc = lib().mwf_gpu_batch_align(self.eng.h, self.h, C.byref(opt))
        if rc != 0:
            raise RuntimeError(f"align failed ({rc}): " + self.eng.error())

    def results(self):
        """(s[int32], n_iter[int64], n_cigar[int32]) as numpy arrays; waits for the stream."""
        s = np.zeros(self.n, dtype=np.int32)
        it = np.zeros(self.n, dtype=np.int64)
        nc = np.zeros(self.n, dtype=np.int32)
        rc = lib().mwf_gpu_batch_results(self.eng.h, self.h, s.ctypes.data, it.ctypes.data, nc.ctypes.data)
        if rc != 0:
            raise RuntimeError(f"results failed ({rc}): " + self.eng.error())
        return s, it, nc

    def cigar(self, i: int, n_cigar: int) -> np.ndarray:
        out = np.zeros(max(1, n_cigar), dtype=np.uint32)
        rc = lib().mwf_gpu_batch_cigar(self.eng.h, self.h, i, out.ctypes.data, n_cigar)
        if rc < 0:
            raise RuntimeError(f"cigar download failed ({rc}): " + self.eng.error())
        return out[:rc]

    def dev_scores_ptr(self) -> int:
        return lib().mwf_gpu_batch_dev_scores(self.h)

    def dev_iters_ptr(self) -> int:
        return lib().mwf_gpu_batch_dev_iters(self.h)

    def dev_status_ptr(self) -> int:
        return lib().mwf_gpu_batch_dev_status(self.h)

    def fetch_cigars(self):
        """Bring every CIGAR of the batch to the host in one copy (cigar() then serves from it)."""
        rc = lib().mwf_gpu_batch_fetch_cigars(self.eng.h, self.h)
        if rc != 0:
            raise RuntimeError(f"CIGAR download failed ({rc}): " + self.eng.error())

    def debug_band(self, opt: MwfOpt, pair: int, cap: int = 1 << 20):
        """[(lo, hi)] in DIAGONAL coordinates of every slice the core pass opened for `pair` (diagnostics)."""
        buf = np.zeros(2 * cap, dtype=np.int32)
        n = lib().mwf_gpu_debug_band(self.eng.h, self.h, C.byref(opt), pair, buf.ctypes.data, cap)
        if n < 0:
            raise RuntimeError(f"debug_band failed ({n}): " + self.eng.error())
        return buf[:2 * n].reshape(-1, 2)
