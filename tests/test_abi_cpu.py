"""CPU-side checks of the C-ABI library: it builds, loads, exports every declared symbol, and its
host-only pieces (mwf_opt_init, mwf_cigar2score, the kalloc-compatible allocator) behave like the
reference's.  No alignment is run here — that needs a GPU (tests/test_gpu_parity.py)."""
import ctypes as C
import re
import os

import pytest

import miniwfa_amd as mw
from miniwfa_amd import api
from conftest import load_golden, golden_inputs, ROOT


def test_library_builds_and_exports_every_declared_symbol():
    L = mw.lib()
    for name in api.ABI_SYMBOLS:
        assert hasattr(L, name), name
    # every function prototype in the two headers is in ABI_SYMBOLS (and therefore exported)
    declared = set()
    for hdr in ("miniwfa.h", "kalloc.h"):
        text = open(os.path.join(ROOT, "include", hdr)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        declared |= set(re.findall(r"\b((?:mwf|km?)_?[a-z0-9_]*)\s*\(", text))
    declared = {d for d in declared if d.startswith(("mwf_", "km_", "kmalloc", "kcalloc", "krealloc", "krelocate", "kfree"))}
    assert declared <= set(api.ABI_SYMBOLS), declared - set(api.ABI_SYMBOLS)


def test_struct_layout_matches_reference():
    assert C.sizeof(api.MwfOpt) == 56 and C.sizeof(api.MwfRst) == 24
    assert api.MwfOpt.step.offset == 24 and api.MwfOpt.max_iter.offset == 32 and api.MwfOpt.min_len.offset == 48
    assert api.MwfRst.n_iter.offset == 8 and api.MwfRst.cigar.offset == 16


def test_opt_init_defaults():
    o = mw.opt_init()   # reference miniwfa.c:11-18
    assert (o.flag, o.x, o.o1, o.e1, o.o2, o.e2, o.step, o.max_s, o.max_iter) == (0, 4, 4, 2, 15, 1, 0, 0, 0)
    assert (o.kmer, o.max_occ, o.min_len) == (13, 2, 30)


def _parse(cig):
    return [int(n) << 4 | api.CIGAR_CHARS.index(c) for n, c in re.findall(r"(\d+)([MIDNSHP=XBid])", cig)]


def test_cigar2score_on_reference_cigars(oracle):
    from oracle.pyoracle import make_opt
    n = 0
    for v in load_golden("exact_small.jsonl"):
        if v["entry"] != "exact" or v["expect"]["cigar"] is None:
            continue
        words = _parse(v["expect"]["cigar"])
        o = mw.opt_init(**{k: v["opt"][k] for k in ("x", "o1", "e1", "o2", "e2")})
        t, q = golden_inputs(v)
        got = mw.cigar2score(o, words)
        assert got == oracle.cigar2score(make_opt(**v["opt"]), words)
        assert got == (v["expect"]["s"], len(t), len(q))
        n += 1
    assert n > 300


def test_kalloc_contract():
    L = mw.lib()
    # km == NULL is libc
    p = L.kmalloc(None, 100)
    assert p
    p = L.krealloc(None, p, 1000)
    L.kfree(None, p)
    assert L.kmalloc(None, 0) is None
    # arenas: nested, reuse after free, calloc zeroes, krelocate compacts, destroy releases
    km = L.km_init()
    child = L.km_init2(km, 0)
    blocks = [L.kmalloc(child, 10 + 37 * i) for i in range(200)]
    assert len(set(blocks)) == 200 and all(blocks)
    for i, b in enumerate(blocks):
        C.memset(b, i & 0xff, 10 + 37 * i)
    for i, b in enumerate(blocks):
        assert C.string_at(b, 10 + 37 * i) == bytes([i & 0xff]) * (10 + 37 * i)
    for b in blocks[::2]:
        L.kfree(child, b)
    z = L.kcalloc(child, 50, 8)
    assert C.string_at(z, 400) == b"\0" * 400
    big = L.krealloc(child, blocks[1], 1 << 20)
    assert C.string_at(big, 47) == bytes([1]) * 47
    moved = L.krelocate(child, big, 47)
    assert C.string_at(moved, 47) == bytes([1]) * 47
    st = api.KmStat()
    L.km_stat(child, C.byref(st))
    assert st.capacity >= st.available > 0 and st.n_cores >= 1
    L.km_destroy(child)
    st2 = api.KmStat()
    L.km_stat(km, C.byref(st2))
    # everything the child took is back and coalesced: one free block per core, only core headers missing
    assert st2.n_blocks == st2.n_cores and st2.capacity - st2.available <= 64 * st2.n_cores
    L.km_destroy(km)
    assert L.krelocate(None, 1234, 8) == 1234  # km == NULL: pointer returned unchanged (reference kalloc.c:179-180)


def test_align_without_gpu_fails_loudly():
    if mw.lib().mwf_gpu_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError):
        mw.Engine(0)


def test_kalloc_macros_compile_and_work(tmp_path):
    """A C program written against the reference's kalloc.h convenience macros (kalloc.h:33-80: KMALLOC, KCALLOC,
    KREALLOC, KEXPAND, KALLOC_POOL_INIT) compiles against include/kalloc.h and runs on the library's allocator."""
    import subprocess
    from miniwfa_amd import build as b
    b.build()
    src = tmp_path / "km.c"
    src.write_text(r'''
#include <stdio.h>
#include <string.h>
#include "kalloc.h"
typedef struct { int a, b; } item_t;
KALLOC_POOL_INIT(item, item_t)
int main(void)
{
	void *km = km_init();
	int *v = 0, i;
	size_t m = 0, n = 0;
	long sum = 0;
	double *w;
	char *z;
	kmp_item_t *mp;
	item_t *x, *y;
	for (i = 0; i < 1000; ++i) {
		if (n == m) KEXPAND(km, v, m);
		v[n++] = i;
	}
	for (i = 0; i < 1000; ++i) sum += v[i];
	KMALLOC(km, w, 10); w[9] = 1.5;
	KCALLOC(km, z, 64);
	for (i = 0; i < 64; ++i) if (z[i]) return 2;
	KREALLOC(km, w, 100); if (w[9] != 1.5) return 3;
	mp = kmp_init_item(km);
	x = kmp_alloc_item(mp); if (x->a || x->b || mp->cnt != 1) return 4;
	x->a = 7; kmp_free_item(mp, x); if (mp->cnt != 0 || mp->n != 1) return 5;
	y = kmp_alloc_item(mp); if (y != x || y->a != 7) return 6;
	kmp_free_item(mp, y);
	kmp_destroy_item(mp);
	kfree(km, v); kfree(km, w); kfree(km, z);
	km_destroy(km);
	printf("%ld %zu\n", sum, m);
	return 0;
}
''')
    exe = tmp_path / "km"
    csrc = os.path.dirname(b.LIB)
    subprocess.run(["gcc", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                    "-L", csrc, "-lmwf_hip", "-Wl,-rpath," + csrc], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    assert out[0] == str(sum(range(1000))) and int(out[1]) >= 1000
