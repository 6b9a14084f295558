// mwf_device.h — device-side helpers shared by every alignment kernel (mwf_kernels.hip generic, mwf_band2.hip packed band,
// mwf_lane.hip short pairs, mwf_mid.hip mid-size pairs of small batches, mwf_sys.hip whole device): the recurrence and its
// traceback byte, kernel-argument access, the shared traceback, per-pair memory views and outputs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mwf_internal.h"

namespace mwf {
namespace dev {

template <int NR>
struct SharedT {
	int32_t flags[3][4];   // per penalty (mod 3): new lo edge live, new hi edge live, end cell reached, payload
	int32_t red[2];        // shrink: first / last good column
	int32_t item;          // work item broadcast
	int32_t word[4];       // scratch broadcast
	int32_t rng_lo[NR], rng_hi[NR]; // column window of the slice held by each H slot
};
typedef SharedT<kMaxRing> Shared;
// penalty sets with max(x, o1+e1, o2+e2) >= 256 (the reference takes any, miniwfa.c:390-393): the one-column-per-lane generic kernel with
// a window table of kBigRing entries (32 KB of LDS), mwf_kernels.hip wfa_bigring_kernel
typedef SharedT<kBigRing> SharedBig;

struct PassResult {
	int32_t status;
	int32_t s;          // final penalty
	int32_t info;       // TB: last_state; SEG: provenance of the end cell
	int32_t n_snap;     // SEG: snapshots taken
	int64_t cells;      // cells computed (n_iter of the reference for the core pass)
};

__device__ __forceinline__ int32_t uni(int32_t v) { return __builtin_amdgcn_readfirstlane(v); }

// A kernel's BatchArgs read from the kernarg segment WHERE they are needed: the compiler otherwise loads every field it will ever
// use at the top of the kernel and keeps it in an SGPR for the kernel's lifetime (the band kernels spilled 130-150 SGPRs that way).
// `fresh` hides the pointer's origin, so loads through its result can neither be hoisted above it nor merged with earlier ones.
typedef const BatchArgs __attribute__((address_space(4))) KArgs;
__device__ __forceinline__ KArgs &kernel_args()
{
	KArgs *p = (KArgs*)__builtin_amdgcn_kernarg_segment_ptr(); // BatchArgs is the kernel's only parameter
	asm volatile("" : "+s"(p));
	return *p;
}
template <typename T>
__device__ __forceinline__ const T &fresh(const T &a)
{
	// (the halves go through v_readfirstlane first: the compiler must see a scalar going into the asm)
	const uint64_t v = (uint64_t)(uintptr_t)(const void*)&a;
	uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int32_t)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int32_t)(uint32_t)(v >> 32));
	asm volatile("" : "+s"(lo), "+s"(hi));
	return *(const T*)(uintptr_t)((uint64_t)hi << 32 | lo);
}

__device__ __forceinline__ uint64_t ld8(const uint8_t *p)
{
	uint64_t x;
	__builtin_memcpy(&x, p, 8); // gfx950 global loads may be unaligned: one global_load_dwordx2
	return x;
}

// k -> k + LCP(ts[k+1..], qs[d+k+1..]); the caller guarantees (k,d) is inside the DP matrix.
__device__ __forceinline__ int32_t extend_run(const uint8_t *ts, const uint8_t *qs, int32_t tl, int32_t ql, int32_t k, int32_t d)
{
	const int32_t j = k + 1, i = d + j;
	const int32_t room = min(tl - j, ql - i);
	const uint8_t *pt = ts + j, *pq = qs + i;
	int32_t n = 0;
	while (n < room) {
		const uint64_t x = ld8(pt + n) ^ ld8(pq + n);
		if (x) { n += (int32_t)(__builtin_ctzll(x) >> 3); break; }
		n += 8;
	}
	return k + min(n, room);
}

// ---- sequences held in LDS by the one-diagonal-per-lane kernels (mwf_lane.hip, mwf_mid.hip) ----------------------------------------
// eight bytes at an arbitrary byte offset of an LDS array (three aligned dwords, two v_alignbyte)
__device__ __forceinline__ uint64_t lds_ld8(const uint8_t *base, int32_t off)
{
	const uint32_t *p = (const uint32_t*)(base + (off & ~3));
	const uint32_t a = p[0], b = p[1], c = p[2];
	const uint32_t lo = __builtin_amdgcn_alignbyte(b, a, (uint32_t)off), hi = __builtin_amdgcn_alignbyte(c, b, (uint32_t)off);
	return (uint64_t)hi << 32 | lo;
}
// Length of the exact-match run t[j..] == q[i..] on byte copies, at most `room` (<= 0: none; j and i must then still be readable
// offsets).  The wave walks together, eight bytes per lane and trip, while any lane's run is open: straight-line trips under one
// uniform branch (a divergent while loop costs ~25 mask instructions per trip).
__device__ __forceinline__ int32_t lds_extend8(const uint8_t *lt, const uint8_t *lq, int32_t j, int32_t i, int32_t room)
{
	int32_t n = 0;
	bool open = room > 0;
	if (__ballot(open)) {
		do { // (bottom-tested: one uniform branch per trip)
			const uint64_t x = lds_ld8(lt, j + n) ^ lds_ld8(lq, i + n);
			const int32_t adv = x ? (int32_t)(__builtin_ctzll(x) >> 3) : 8;
			n += open ? adv : 0;
			open = open && x == 0 && n < room;
		} while (__ballot(open));
	}
	return max(min(n, room), 0);
}
// 2-bit copies (pairs of plain A/C/G/T): sixteen bases per dword, base j at bits 2*(j & 15) of dword j >> 4 — one ds_read2_b32 and one
// v_alignbit per sequence give SIXTEEN bases from any position: a trip is two LDS instructions instead of six, and a run must be
// twice as long before a second trip is needed (as in mwf_band2.hip).
__device__ __forceinline__ uint32_t lds_seq16(const uint8_t *b2, int32_t j)
{
	const uint32_t *p = (const uint32_t*)(b2 + ((j >> 4) << 2));
	return __builtin_amdgcn_alignbit(p[1], p[0], (uint32_t)j << 1);
}
__device__ __forceinline__ int32_t lds_extend16(const uint8_t *lt, const uint8_t *lq, int32_t j, int32_t i, int32_t room)
{
	int32_t n = 0;
	bool open = room > 0;
	if (__ballot(open)) {
		do {
			const uint32_t x = lds_seq16(lt, j + n) ^ lds_seq16(lq, i + n);
			const int32_t adv = x ? (int32_t)(__builtin_ctz(x) >> 1) : 16;
			n += open ? adv : 0;
			open = open && x == 0 && n < room;
		} while (__ballot(open));
	}
	return max(min(n, room), 0);
}
// Bytes -> 2 bits per base into LDS at `dst` (two dwords of slack behind the last base), by T threads.  Returns nonzero when a byte
// is not one of A, C, G, T.  code = (byte >> 1) & 3: A 0, C 1, T 2, G 3.
template <int T>
__device__ __forceinline__ uint32_t lds_pack2bit(const uint8_t *src, int32_t len, uint8_t *dst)
{
	uint32_t bad = 0;
	const int32_t n_dw = (len >> 4) + 2;
	for (int32_t w = threadIdx.x; w < n_dw; w += T) {
		uint32_t out = 0;
		const int32_t b0 = w << 4;
		if (b0 + 16 <= len) {
			uint32_t q[4];
			__builtin_memcpy(q, src + b0, 16);
#pragma unroll
			for (int k = 0; k < 4; ++k) {
				const uint32_t x = q[k], code = (x >> 1) & 0x03030303u;
				const uint32_t lo1 = code & 0x01010101u, hi1 = (code >> 1) & 0x01010101u;
				const uint32_t expect = 0x41414141u + (lo1 & ~hi1) * 0x02u + (hi1 & ~lo1) * 0x13u + (hi1 & lo1) * 0x06u; // 0 'A', 1 'C', 2 'T', 3 'G'
				bad |= x ^ expect;
				out |= ((code | code >> 6 | code >> 12 | code >> 18) & 0xffu) << (8 * k);
			}
		} else {
#pragma unroll 1
			for (int32_t k = 0; k < 16 && b0 + k < len; ++k) {
				const uint32_t x = src[b0 + k], code = (x >> 1) & 3u;
				bad |= x ^ ((0x47544341u >> (8 * code)) & 0xffu);
				out |= code << (2 * k);
			}
		}
		*(uint32_t*)(dst + 4 * w) = out;
	}
	return bad;
}

// largest value of v over the lanes of the wave (rare paths only: six cross-lane steps)
__device__ __forceinline__ int32_t wave_max(int32_t v)
{
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
	return v;
}
// Early hand-back of a pair that will outgrow the span of the kernel it was given.  A window is about twice the penalty wide and the
// penalty grows in proportion to the progress along the target, so once the furthest offset `kmax` reached at penalty s is known the
// final window is about 2 s tl / (kmax + 1): beyond `cap` columns (with a safety factor that tightens as s grows and the estimate's noise falls) the pair goes back
// to the host NOW — after s penalties instead of after cap / 2 — with the estimate, so that the re-run starts on a kernel that fits.
// Returns 0 (carry on) or the estimated window.  A wrong guess costs a re-run on a wider kernel, never a wrong result.
__device__ __forceinline__ int32_t window_forecast(int32_t s, int32_t kmax, int32_t tl, int32_t ql, int32_t cap, bool last_resort = false)
{
	if (kmax < 8 || tl < 64) return 0;
	const int64_t need = min(2 * (int64_t)s * tl / min(kmax + 1, tl) + 16, (int64_t)tl + ql + 1); // (no window is wider than the matrix has diagonals)
	// the progress after s penalties is that of ~s / 4.5 independent differences: relative noise ~ sqrt(4.5 / s), i.e. 0.27, 0.13 and 0.07 at
	// penalties 64, 256 and 1024 — the factors leave five of those: a batch has thousands of pairs and a pair handed back by mistake is
	// re-run on a slower kernel (factors of three sigma sent three pairs of the 1024 x 10 kb headline batch to the generic kernel per
	// align: 28.9 instead of 17.5 ms).  last_resort: the widest kernel of its family — what it hands back goes to a much slower one, so it
	// only does at penalty 1024 and with a margin of 1.5.
	if (last_resort && s < 1024) return 0;
	const int32_t slack = last_resort ? 15 : s < 128 ? 25 : s < 512 ? 17 : 13; // tenths (callers forecast at penalties 64, 256, 1024)
	return need * 10 > (int64_t)cap * slack ? (int32_t)min(need, (int64_t)0x3fffffff) : 0;
}

// offset k on diagonal d is a cell of the DP matrix (reference good_diag, miniwfa.c:139-142)
__device__ __forceinline__ bool in_matrix(int32_t d, int32_t k, int32_t tl, int32_t ql)
{
	return (uint32_t)(k + 1) < (uint32_t)(tl + 1) && (uint32_t)(d + k + 1) < (uint32_t)(ql + 1);
}

struct Cell { int32_t h, e1, f1, e2, f2; uint32_t tb; };

// The recurrence and its tie-breaking (miniwfa.c:267-278 values, :289-306 traceback byte):
// gap states prefer "open" on ties; H prefers mismatch, then insertion piece 1, piece 2, deletion piece 1, piece 2.
// Written with selects and arithmetic only, so that it compiles to straight-line code; WANT_TB=false drops the byte.
template <bool WANT_TB = true>
__device__ __forceinline__ Cell wf_cell(int32_t hx, int32_t o1m, int32_t g1m, int32_t o2m, int32_t g2m,
                                        int32_t o1p, int32_t g1p, int32_t o2p, int32_t g2p)
{
	Cell c;
	c.e1 = max(o1m, g1m);
	c.e2 = max(o2m, g2m);
	c.f1 = max(o1p, g1p) + 1;
	c.f2 = max(o2p, g2p) + 1;
	const int32_t e = max(c.e1, c.e2), f = max(c.f1, c.f2), g = max(e, f), m = hx + 1;
	c.h = max(m, g);
	c.tb = 0;
	if (WANT_TB) {
		const uint32_t xe1 = (uint32_t)(g1m > o1m), xe2 = (uint32_t)(g2m > o2m), xf1 = (uint32_t)(g1p > o1p), xf2 = (uint32_t)(g2p > o2p);
		const uint32_t ze = 1u + ((uint32_t)(c.e1 < c.e2) << 1);  // 1: E1, 3: E2 (ties -> E1)
		const uint32_t zf = 2u + ((uint32_t)(c.f1 < c.f2) << 1);  // 2: F1, 4: F2 (ties -> F1)
		uint32_t z = e >= f ? ze : zf;                            // ties -> insertion
		z = m >= g ? 0u : z;                                      // ties -> mismatch
		c.tb = z | (xe1 << 3) | (xf1 << 4) | (xe2 << 5) | (xf2 << 6);
	}
	return c;
}

// Provenance follows the choices recorded in the traceback byte (second loop of wf_next_seg, miniwfa.c:504-523).
__device__ __forceinline__ Cell shadow_cell(uint32_t tb, int32_t hx, int32_t o1m, int32_t g1m, int32_t o2m, int32_t g2m,
                                            int32_t o1p, int32_t g1p, int32_t o2p, int32_t g2p)
{
	Cell c;
	c.e1 = (tb & 0x08u) ? g1m : o1m;
	c.f1 = (tb & 0x10u) ? g1p : o1p;
	c.e2 = (tb & 0x20u) ? g2m : o2m;
	c.f2 = (tb & 0x40u) ? g2p : o2p;
	const uint32_t z = tb & 7u;
	c.h = z == 1 ? c.e1 : z == 2 ? c.f1 : z == 3 ? c.e2 : z == 4 ? c.f2 : hx;
	c.tb = 0;
	return c;
}

// A source slice: row pointer (column-indexed) and its window.
struct Src {
	const int32_t *p;
	int32_t lo, hi;
	__device__ __forceinline__ int32_t at(int32_t c) const { return p[c]; } // unchecked
	__device__ __forceinline__ int32_t rd(int32_t c) const                  // window-checked
	{
		const int32_t v = p[c];
		return (c >= lo && c <= hi) ? v : kNegInf;
	}
};

// bits of the 64-column word starting at column w0 that fall inside [lo,hi]
__device__ __forceinline__ unsigned long long window_mask(int32_t w0, int32_t lo, int32_t hi)
{
	if (hi < w0 || lo > w0 + 63 || lo > hi) return 0ull;
	unsigned long long m = ~0ull;
	if (lo > w0) m &= ~0ull << (lo - w0);
	if (hi < w0 + 63) m &= ~0ull >> (w0 + 63 - hi);
	return m;
}

// Everything a pass needs to know about the pair and this slot's memory.
struct PairMem {
	const uint8_t *ts, *qs;
	int32_t tl, ql;
	int32_t *H, *E1, *F1, *E2, *F2;       // ring rows (row r of array X at X + r*W), column-indexed
	int32_t *sH, *sE1, *sF1, *sE2, *sF2;  // shadow ring (low-memory first pass)
	unsigned long long *good;
	uint8_t *tb;
	int64_t *row_off;
	int32_t *row_lo;
	int32_t *snap, *snap_meta, *seg;
	int32_t *dbg;
	// traceback bytes laid out per epoch of 256 penalties and chunk slot (mwf_sys.hip); null: rows back to back (row_off / row_lo)
	const int64_t *ep;
	int32_t ep_ow, ep_p, ep_kw;
	// rows of a fixed width that all start at the same column (mwf_lane.hip): byte of (row, col) at row * tb_stride + col - tb_left; 0: not this layout
	int32_t tb_stride, tb_left;
	// Packed band kernel, folded form with traceback (mwf_band2.hip): bits 3 and 4 of a byte look FORWARD — E1 (F1) of this cell exceeds the H it
	// would be opened from, i.e. the E1 (F1) of penalty + e1 in the neighbouring column is an extension of this one — instead of saying whether
	// this cell's own E1 (F1) was extended: the walk reads them from the cell an extension would come from.
	int32_t tb_fwd;
	// 2-bit copies of the two sequences in LDS (kernels that hold them: the traceback's back-match then stays on chip); null: compare
	// the bytes at ts / qs (which may themselves point into LDS)
	const uint8_t *t2, *q2;
};

// the traceback byte of (penalty row + 1, column col)
__device__ __forceinline__ uint32_t tb_byte(const PairMem &M, int32_t row, int32_t col)
{
	if (M.ep) {
		const int64_t base = M.ep[2 * (row >> 8)], gn = M.ep[2 * (row >> 8) + 1];
		const int32_t g = col / M.ep_ow;
		return M.tb[base + ((int64_t)(row & 255) * (int32_t)(gn >> 32) + (g - (int32_t)(gn & 0xffffffff))) * M.ep_kw + (col - (g * M.ep_ow - M.ep_p))];
	}
	if (M.tb_stride) return M.tb[row * M.tb_stride + (col - M.tb_left)]; // (no row table to read first: one memory round trip per traceback step)
	return M.tb[M.row_off[row] + (col - M.row_lo[row])];
}

// Traceback on one wave (reference wf_traceback, miniwfa.c:329-377).  Ops are emitted from the end of
// the alignment to its start, so writing them backwards from the end of the scratch buffer leaves the
// CIGAR in input order.  Returns n_cigar (>= 0) or -1 when the scratch buffer is too small.
// (ArgsT: BatchArgs, or BatchArgs in the constant address space when a kernel reads its arguments where it needs them)
template <typename ArgsT>
static __device__ int32_t traceback_wave(const ArgsT &A, const PairMem &M, uint32_t *scratch, int64_t cap,
                                  int32_t s_final, int32_t last, int32_t *end_state)
{
	const auto &P = A.pen;
	const int32_t lane = threadIdx.x & 63;
	int32_t i = M.ql - 1, k = M.tl - 1, row = s_final - 1;
	int64_t pos = cap;
	int32_t run_op = -1, run_len = 0;
	bool overflow = false;
	auto push = [&](int32_t op, int32_t len) { // reference wf_cigar_push1, miniwfa.c:51-62
		if (op == run_op) { run_len += len; return; }
		if (run_op >= 0) {
			if (pos == 0) { overflow = true; return; }
			--pos;
			if (lane == 0) scratch[pos] = (uint32_t)run_len << 4 | (uint32_t)run_op;
		}
		run_op = op, run_len = len;
	};
	// The byte of the cell the walk stands on is requested as soon as the cell is known — before the back-match along its diagonal
	// (which changes neither the row nor the column): the two round trips of a step travel together instead of one after the other.
	uint32_t x_next = 0;
	if (row >= 0 && i >= 0 && k >= 0) x_next = tb_byte(M, row, i - k + M.tl + 1);
	// A step is one DEPENDENT byte from a matrix written long ago: a round trip to HBM (~1 us) per CIGAR operation, which is what the traceback of
	// a pair costs.  The path moves back by at most nH rows and one column per step, so every 40 rows the 64 lanes fetch the byte in the walk's
	// current column of the next 64 rows — one round trip, together — and the steps through those rows find their cache lines in L2.
#ifndef MWF_TB_NO_PREFETCH
	const bool prefetch = A.tb_slot_bytes > 0;
#else
	const bool prefetch = false;
#endif
	int32_t pf_at = row;
	while (i >= 0 && k >= 0 && !overflow) {
		uint32_t pf = 0;
		const bool pf_now = prefetch && row <= pf_at && row > 0;
		if (pf_now) {
			const int32_t r = row - 1 - lane, col = i - k + M.tl + 1;
			if (r >= 0) {
				int64_t at;
				if (M.ep) { // (the whole-device kernel's layout: per epoch of 256 penalties and chunk slot)
					const int64_t base = M.ep[2 * (r >> 8)], gn = M.ep[2 * (r >> 8) + 1];
					const int32_t g = col / M.ep_ow;
					at = base + ((int64_t)(r & 255) * (int32_t)(gn >> 32) + (g - (int32_t)(gn & 0xffffffff))) * M.ep_kw + (col - (g * M.ep_ow - M.ep_p));
				} else at = M.tb_stride ? (int64_t)r * M.tb_stride + (col - M.tb_left) : M.row_off[r] + (col - M.row_lo[r]);
				at = min(max(at, (int64_t)0), (int64_t)A.tb_slot_bytes - 1); // (an older row may not reach this column: any byte of the arena will do)
				pf = M.tb[at];
			}
			pf_at = row - 40;
		}
		if (last == 0) { // greedy back-match, 64 bases per trip (miniwfa.c:335-341)
			int32_t run = 0;
			if (M.t2) { // 2-bit copies in LDS: one base per lane and trip
				for (;;) {
					const int32_t ii = i - run - lane, kk = k - run - lane;
					bool eq = ii >= 0 && kk >= 0;
					if (eq) eq = ((((const uint32_t*)M.q2)[ii >> 4] >> ((ii & 15) << 1)) & 3u) == ((((const uint32_t*)M.t2)[kk >> 4] >> ((kk & 15) << 1)) & 3u);
					const unsigned long long m = __ballot(eq);
					const int32_t n = m == ~0ull ? 64 : (int32_t)__builtin_ctzll(~m);
					run += n;
					if (n < 64) break;
				}
				goto matched;
			}
			// far from the start of both sequences: eight bases per lane, 512 per trip (a long pair is mostly exact matches, and a
			// trip is a round trip to the sequences)
			while (i - run >= 511 && k - run >= 511) {
				const int32_t ii = i - run - 8 * lane, kk = k - run - 8 * lane;
				const uint64_t x = ld8(M.qs + ii - 7) ^ ld8(M.ts + kk - 7);
				const int32_t m8 = x ? (int32_t)(__builtin_clzll(x) >> 3) : 8; // equal bases from the top byte down
				const unsigned long long stop = __ballot(m8 < 8);
				if (stop == 0) { run += 512; continue; }
				const int32_t first = (int32_t)__builtin_ctzll(stop);
				run += 8 * first + __builtin_amdgcn_readlane(m8, first);
				goto matched;
			}
			for (;;) {
				const int32_t ii = i - run - lane, kk = k - run - lane;
				const bool eq = ii >= 0 && kk >= 0 && M.qs[ii] == M.ts[kk];
				const unsigned long long m = __ballot(eq);
				const int32_t n = m == ~0ull ? 64 : (int32_t)__builtin_ctzll(~m);
				run += n;
				if (n < 64) break;
			}
		matched:
			if (run > 0) push(7, run);
			i -= run, k -= run;
			if (i < 0 || k < 0) break;
		}
		if (row < 0) { overflow = true; break; }
		if (pf_now) asm volatile("" :: "v"(pf)); // (the prefetched bytes have arrived — and stay in L2)
		const uint32_t x = x_next;
		const int32_t state = last == 0 ? (int32_t)(x & 7u) : last;           // :346
		int32_t ext = state > 0 ? (int32_t)(x >> (state + 2)) & 1 : 0;        // :347
		uint32_t x_from = 0;
		const bool fwd = M.tb_fwd && (state == 1 || state == 2);
		if (fwd) { // the first gap piece's bit sits in the cell the extension would come from; a cell its row does not hold was not computed: dead, no extension
			const int32_t r2 = row - P.e1, c2 = i - k + M.tl + 1 + (state == 1 ? -1 : 1);
			ext = 0;
			if (r2 >= 0) {
				const int64_t at = M.row_off[r2] + (c2 - M.row_lo[r2]);
				if (c2 >= M.row_lo[r2] && at < M.row_off[r2 + 1]) x_from = M.tb[at], ext = (int32_t)(x_from >> (state + 2)) & 1;
			}
		}
		if (state == 0) { push(8, 1); --i, --k; row -= P.x; }
		else if (state == 1) { push(1, 1); --i; row -= ext ? P.e1 : P.oe1; }
		else if (state == 3) { push(1, 1); --i; row -= ext ? P.e2 : P.oe2; }
		else if (state == 2) { push(2, 1); --k; row -= ext ? P.e1 : P.oe1; }
		else { push(2, 1); --k; row -= ext ? P.e2 : P.oe2; }
		last = (state > 0 && ext) ? state : 0;                                // :365
		if (fwd && ext) x_next = x_from; // (the cell just looked at)
		else if (row >= 0 && i >= 0 && k >= 0) x_next = tb_byte(M, row, i - k + M.tl + 1); // (the cell the path came from: its row holds it)
	}
	end_state[0] = row, end_state[1] = i, end_state[2] = k;
	if (i >= 0) push(1, i + 1);          // :368-369
	else if (k >= 0) push(2, k + 1);
	push(-2, 0);                         // flush the pending run
	if (overflow) return -1;
	return (int32_t)(cap - pos);
}


// Set up the per-slot memory views of one pair.
template <typename ArgsT>
__device__ __forceinline__ void pair_mem(const ArgsT &A, int32_t slot, int32_t pair, PairMem &M)
{
	const auto &P = A.pen;
	M.tl = A.tl[pair], M.ql = A.ql[pair];
	M.ts = A.seqs + A.t_off[pair], M.qs = A.seqs + A.q_off[pair];
	const int64_t W = A.W;
	int32_t *ring = A.ring + (int64_t)slot * A.ring_slot_ints;
	M.H = ring, M.E1 = M.H + P.nH * W, M.F1 = M.E1 + P.n1 * W, M.E2 = M.F1 + P.n1 * W, M.F2 = M.E2 + P.n2 * W;
	M.sH = M.sE1 = M.sF1 = M.sE2 = M.sF2 = 0;
	if (A.sring) {
		int32_t *sr = A.sring + (int64_t)slot * A.ring_slot_ints;
		M.sH = sr, M.sE1 = M.sH + P.nH * W, M.sF1 = M.sE1 + P.n1 * W, M.sE2 = M.sF1 + P.n1 * W, M.sF2 = M.sE2 + P.n2 * W;
	}
	M.good = A.good + (int64_t)slot * P.nH * A.GW;
	M.tb = A.tb ? A.tb + (int64_t)slot * A.tb_slot_bytes : 0;
	M.row_off = A.row_off ? A.row_off + (int64_t)slot * A.rows_slot : 0;
	M.row_lo = A.row_lo ? A.row_lo + (int64_t)slot * A.rows_slot : 0;
	M.snap = A.snap ? A.snap + (int64_t)slot * A.snap_slot_ints : 0;
	M.snap_meta = A.snap_meta ? A.snap_meta + (int64_t)slot * A.snap_meta_slot : 0;
	M.seg = A.seg ? A.seg + (int64_t)slot * 2 * A.seg_slot : 0;
	M.dbg = A.dbg;
	M.ep = 0, M.ep_ow = 0, M.ep_p = 0, M.ep_kw = 0;
	M.tb_stride = 0, M.tb_left = 0, M.tb_fwd = 0;
	M.t2 = M.q2 = 0;
}

// After the forward pass(es): traceback on the first wave, CIGAR into the pool, per-pair outputs.
// a workgroup's share of the CIGAR pool in block mode (BatchArgs::cig_block): where its next CIGAR goes and how many words are left there
struct CigLocal { unsigned long long base; int32_t left; };

template <typename ArgsT>
__device__ __forceinline__ void finish_pair(const ArgsT &A, const PairMem &M, int32_t slot, int32_t pair,
                                            const PassResult &R, int32_t status, int64_t cells1, CigLocal *loc = nullptr)
{
	int32_t n_cigar = 0;
	int64_t cig_off = 0;
	if (A.want_cigar && status == ST_OK) {
		__syncthreads();
		if (threadIdx.x < 64) {
			uint32_t *scratch = A.cig_scratch + (int64_t)slot * A.cig_scratch_slot;
			int32_t end_state[3];
			n_cigar = traceback_wave(A, M, scratch, A.cig_scratch_slot, R.s, R.info, end_state);
			if (n_cigar < 0) status = ST_INTERNAL, n_cigar = 0;
			else {
				unsigned long long off = 0;
				const int32_t blk = loc ? A.cig_block : 0;
				if (blk > 0 && n_cigar <= blk / 4) {
					// block mode: from the workgroup's current block; one that cannot hold this CIGAR is abandoned (less than a quarter of it is lost: the host
					// sized the pool for that) and the next one taken with ONE atomic for the dozen pairs it will hold
					if (n_cigar > loc->left) {
						unsigned long long nb = 0;
						if (threadIdx.x == 0) nb = atomicAdd(A.cig_head, (unsigned long long)blk);
						loc->base = ((unsigned long long)(uint32_t)uni((int32_t)(nb >> 32)) << 32) | (uint32_t)uni((int32_t)(nb & 0xffffffffu));
						loc->left = blk;
					}
					off = loc->base, loc->base += (unsigned long long)n_cigar, loc->left -= n_cigar;
				} else {
					if (threadIdx.x == 0) off = atomicAdd(A.cig_head, (unsigned long long)n_cigar);
					off = ((unsigned long long)(uint32_t)uni((int32_t)(off >> 32)) << 32) | (uint32_t)uni((int32_t)(off & 0xffffffffu));
				}
				if ((int64_t)off + n_cigar > A.cig_pool_words) status = ST_CIGAR_OVERFLOW, n_cigar = 0;
				else {
					cig_off = (int64_t)off;
					const uint32_t *src = scratch + (A.cig_scratch_slot - n_cigar);
					for (int32_t j = threadIdx.x; j < n_cigar; j += 64) A.cig_pool[off + j] = src[j];
				}
			}
			if (threadIdx.x == 0 && A.out_dbg) {
				A.out_dbg[4 * pair] = end_state[0], A.out_dbg[4 * pair + 1] = end_state[1];
				A.out_dbg[4 * pair + 2] = end_state[2], A.out_dbg[4 * pair + 3] = R.info;
			}
		}
	}
	if (threadIdx.x == 0) {
		// -1 is the reference's "stopped" answer (miniwfa.c:427); a pair the host still has to re-run holds -2
		A.out_s[pair] = status == ST_OK ? R.s : status == ST_STOPPED ? -1 : -2;
		A.out_iter[pair] = R.cells;
		A.out_ncig[pair] = n_cigar;
		A.out_cigoff[pair] = cig_off;
		A.out_cells1[pair] = cells1;
		A.out_status[pair] = status;
		if (status == ST_BAND_OVERFLOW && A.retry_count) { // handed back: onto the list of the follow-up launch (BatchArgs::retry_ids)
			const unsigned int k = atomicAdd(A.retry_count, 1u);
			if (k < (unsigned int)A.retry_cap) A.retry_ids[k] = pair;
		}
	}
	__syncthreads();
}

} // namespace dev
} // namespace mwf
