// mwf_kernels.hip — gfx950 kernels for the exact WFA score-and-CIGAR path.
//
// What the reference does on one CPU thread per pair (miniwfa.c:380-435 core loop, :551-601
// low-memory first pass, :329-377 traceback) is restated here for a wave64 machine:
//
//   * one workgroup owns one sequence pair; its threads own diagonals.  A penalty step is
//     "advance + extend" fused: every thread computes the five wavefront values of its diagonals
//     (reference wf_next_score/wf_next_tb, miniwfa.c:261-308), immediately walks the new H offset
//     along exact matches with unaligned 8-byte loads (reference wf_extend1_padded, :212-226) and
//     stores the extended value.  One s_barrier per penalty.
//   * the ring keeps only what the recurrence can read again: H for max_pen+1 penalties, E1/F1
//     for e1+1, E2/F2 for e2+1 (27 array-slices with default penalties instead of the reference's
//     85, miniwfa.c:90).  Columns are absolute (column = diagonal + tl + 1), so a wave's 64 lanes
//     always touch one aligned 256-byte segment per array — coalesced int32 loads/stores.
//   * a slice is a window [lo,hi] of columns; reads outside a source window yield NEG_INF (what
//     the reference's pads supply, miniwfa.c:96-99).  Waves whose 64 columns lie inside every
//     source window take a branch-free path without any window test.
//   * band bookkeeping (edge rule :325-326, growth :417-418, shrink every 256 penalties :144-171,
//     checkpoint resets :413-416, n_iter :421, stop rules :422-425) is replicated exactly; it is
//     evaluated redundantly by every thread from a few LDS flags, so no second barrier is needed.
//     "Does any array hold an in-matrix offset on this diagonal" — all the shrink needs from the
//     E/F arrays the ring no longer keeps — is recorded as one ballot bit per cell, only for the
//     slices a shrink can still see.
//   * traceback bytes (7-bit packing of :289-306) go to a per-workgroup arena, rows back to back
//     (row offset = cells computed so far); the traceback itself runs on the first wave right
//     after the forward pass, matching runs compared 64 bases per step with a ballot.
//   * low-memory mode: the first pass carries a shadow ring of provenance indices and flattens it
//     every `step` penalties (:451-474, :495-526) but never materialises traceback bytes; the
//     checkpoints it yields drive the band resets of the second pass.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "mwf_internal.h"

#include "mwf_device.h"

namespace mwf {

using namespace dev;

namespace {

// Flatten the shadow ring into the next snapshot and renumber it (reference wf_snapshot1, miniwfa.c:451-474).
// Snapshot record in snap_meta, stride 4+4*NS ints: [arena offset lo, hi, penalty, n] then per array-slice
// [penalty of the slice, first column, width, first index].  Array-slices are listed H slots, then E1, F1, E2, F2.
template <typename ArgsT>
__device__ __forceinline__ Penalty load_penalty(const ArgsT &A)
{
	Penalty P;
	P.x = A.pen.x, P.oe1 = A.pen.oe1, P.e1 = A.pen.e1, P.oe2 = A.pen.oe2, P.e2 = A.pen.e2, P.o1 = A.pen.o1, P.o2 = A.pen.o2;
	P.nH = A.pen.nH, P.n1 = A.pen.n1, P.n2 = A.pen.n2;
	return P;
}

template <int T, typename ArgsT, typename ShT>
__device__ bool take_snapshot(const ArgsT &A, const PairMem &M, ShT &sh, int32_t n_snap, int64_t &snap_used,
                              int32_t s, int32_t curH, int32_t cur1, int32_t cur2)
{
	const Penalty P = load_penalty(A);
	const int32_t NS = P.nH + 2 * P.n1 + 2 * P.n2, MS = 4 + 4 * NS;
	int32_t *meta = M.snap_meta + (int64_t)n_snap * MS;
	if (threadIdx.x == 0) {
		int32_t t = 0, q = 4;
		bool ok = (int64_t)(n_snap + 1) * MS <= A.snap_meta_slot;
		if (ok) {
			for (int32_t j = 0; j < P.nH; ++j, q += 4) {
				const int32_t age = curH - j < 0 ? curH - j + P.nH : curH - j;
				const int32_t lo = sh.rng_lo[j], hi = sh.rng_hi[j], n = hi >= lo ? hi - lo + 1 : 0;
				meta[q] = s - age, meta[q + 1] = lo, meta[q + 2] = n, meta[q + 3] = t;
				t += n;
			}
			for (int32_t arr = 0; arr < 4; ++arr) {
				const int32_t nr = arr < 2 ? P.n1 : P.n2, cur = arr < 2 ? cur1 : cur2;
				for (int32_t j = 0; j < nr; ++j, q += 4) {
					const int32_t age = cur - j < 0 ? cur - j + nr : cur - j;
					int32_t lo = 1, hi = 0;
					if (s - age >= 0) {
						const int32_t hj = curH - age < 0 ? curH - age + P.nH : curH - age;
						lo = sh.rng_lo[hj], hi = sh.rng_hi[hj];
					}
					const int32_t n = hi >= lo ? hi - lo + 1 : 0;
					meta[q] = s - age, meta[q + 1] = lo, meta[q + 2] = n, meta[q + 3] = t;
					t += n;
				}
			}
			ok = snap_used + t <= A.snap_slot_ints;
			meta[0] = (int32_t)(snap_used & 0xffffffff), meta[1] = (int32_t)(snap_used >> 32), meta[2] = s, meta[3] = t;
		}
		sh.word[0] = ok ? t : -1;
	}
	__syncthreads();
	const int32_t total = uni(sh.word[0]);
	if (total < 0) return false;
	int32_t *x = M.snap + snap_used;
	for (int32_t k = 0; k < NS; ++k) {
		const int32_t lo = uni(meta[4 + 4 * k + 1]), n = uni(meta[4 + 4 * k + 2]), t0 = uni(meta[4 + 4 * k + 3]);
		if (n == 0) continue;
		int32_t *row;
		if (k < P.nH) row = M.sH + (int64_t)k * A.W;
		else if (k < P.nH + P.n1) row = M.sE1 + (int64_t)(k - P.nH) * A.W;
		else if (k < P.nH + 2 * P.n1) row = M.sF1 + (int64_t)(k - P.nH - P.n1) * A.W;
		else if (k < P.nH + 2 * P.n1 + P.n2) row = M.sE2 + (int64_t)(k - P.nH - 2 * P.n1) * A.W;
		else row = M.sF2 + (int64_t)(k - P.nH - 2 * P.n1 - P.n2) * A.W;
		for (int32_t i = threadIdx.x; i < n; i += T) {
			x[t0 + i] = row[lo + i];
			row[lo + i] = t0 + i;
		}
	}
	snap_used += total;
	__syncthreads();
	return true;
}

// Walk the provenance chain back through the snapshots (reference wf_traceback_seg, miniwfa.c:528-549). One thread.
template <typename ArgsT>
__device__ int32_t trace_checkpoints(const ArgsT &A, const PairMem &M, int32_t n_snap, int32_t last)
{
	const Penalty P = load_penalty(A);
	const int32_t NS = P.nH + 2 * P.n1 + 2 * P.n2, MS = 4 + 4 * NS;
	if (n_snap > A.seg_slot) return ST_SNAP_OVERFLOW;
	for (int32_t j = n_snap - 1; j >= 0; --j) {
		const int32_t *meta = M.snap_meta + (int64_t)j * MS;
		const int64_t base = (int64_t)(uint32_t)meta[0] | (int64_t)meta[1] << 32;
		int32_t k;
		for (k = 0; k < NS; ++k) {
			const int32_t n = meta[4 + 4 * k + 2], t0 = meta[4 + 4 * k + 3];
			if (n > 0 && last >= t0 && last < t0 + n) break;
		}
		if (k == NS) return ST_INTERNAL;
		M.seg[2 * j] = meta[4 + 4 * k];
		M.seg[2 * j + 1] = meta[4 + 4 * k + 1] + (last - meta[4 + 4 * k + 3]);
		last = M.snap[base + last];
	}
	return last == -1 ? ST_OK : ST_INTERNAL;
}

// One forward pass over a pair.
//   TB : store traceback bytes, honour checkpoints (core pass with MWF_F_CIGAR)
//   SEG: low-memory first pass (shadow ring + snapshots, no traceback bytes, no stop rules, miniwfa.c:569-589)
template <int T, bool TB, bool SEG, typename ArgsT, typename ShT>
__device__ PassResult forward_pass(const ArgsT &A, const PairMem &M, ShT &sh, int32_t n_seg, bool trace_band)
{
	const Penalty P = load_penalty(A);
	const int32_t tl = M.tl, ql = M.ql, cmax = tl + ql + 1; // columns 1..cmax hold diagonals -tl..ql
	const int32_t tid = threadIdx.x, lane = tid & 63, wave0 = tid & ~63;
	const int64_t W = A.W;
	PassResult R;
	R.status = ST_OK, R.s = 0, R.info = 0, R.n_snap = 0, R.cells = 0;

	// ---- penalty 0: the origin (reference wf_stripe_init, miniwfa.c:103-121) and its extension
	if (tid == 0) {
		for (int32_t j = 0; j < P.nH; ++j) sh.rng_lo[j] = 1, sh.rng_hi[j] = 0;
		for (int32_t j = 0; j < 12; ++j) (&sh.flags[0][0])[j] = 0;
		const int32_t c0 = tl + 1;
		const int32_t k0 = extend_run(M.ts, M.qs, tl, ql, -1, 0);
		M.H[c0] = k0;
		M.E1[c0] = M.F1[c0] = M.E2[c0] = M.F2[c0] = kNegInf;
		// a ring deeper than 256 slices still holds the origin when the first shrink looks at it (miniwfa.c:144-171): its good bit
		M.good[c0 >> 6] = in_matrix(0, k0, tl, ql) ? 1ull << (c0 & 63) : 0ull;
		if (SEG) {
			M.sH[c0] = -1;
			M.sE1[c0] = M.sF1[c0] = M.sE2[c0] = M.sF2[c0] = kNegInf;
		}
		sh.rng_lo[0] = sh.rng_hi[0] = c0;
		sh.word[1] = k0;
	}
	__syncthreads();
	{
		const int32_t k0 = uni(sh.word[1]);
		if (k0 == tl - 1 && k0 == ql - 1) { // identical (or both empty) sequences
			R.info = SEG ? -1 : 0;
			return R;
		}
	}

	int32_t s = 0, wf_lo = tl + 1, wf_hi = tl + 1;  // live band, in columns
	int32_t curH = 0, cur1 = 0, cur2 = 0, par = 0, sid = 0;
	int32_t snap_ctr = A.step == 1 ? 0 : 1;         // (s+1) % step
	int64_t cells = 0, tb_used = 0, snap_used = 0;
	int32_t n_snap = 0;

	for (;;) {
		// ---- checkpoint reset of the second pass (miniwfa.c:413-416)
		if (TB && sid < n_seg) {
			if (uni(M.seg[2 * sid]) == s) {
				const int32_t c = uni(M.seg[2 * sid + 1]);
				if (c < wf_lo || c > wf_hi) { R.status = ST_INTERNAL; break; }
				wf_lo = wf_hi = c;
				++sid;
			}
		}
		// ---- window of the new slice (miniwfa.c:417-418)
		const int32_t lo = wf_lo > 1 ? wf_lo - 1 : 1;
		const int32_t hi = wf_hi < cmax ? wf_hi + 1 : cmax;
		const int32_t w = hi - lo + 1;
		if (SEG) {
			if (snap_ctr == 0) {
				if (!take_snapshot<T>(A, M, sh, n_snap, snap_used, s, curH, cur1, cur2)) { R.status = ST_SNAP_OVERFLOW; break; }
				++n_snap;
			}
			snap_ctr = snap_ctr + 1 == A.step ? 0 : snap_ctr + 1;
		}
		const int32_t s_new = s + 1;
		const int32_t newH = curH + 1 == P.nH ? 0 : curH + 1;
		const int32_t new1 = cur1 + 1 == P.n1 ? 0 : cur1 + 1;
		const int32_t new2 = cur2 + 1 == P.n2 ? 0 : cur2 + 1;
		const int32_t npar = par + 1 == 3 ? 0 : par + 1;
		if (TB) {
			if (s_new - 1 >= A.rows_slot) { R.status = ST_ROWS_OVERFLOW; break; }
			if (tb_used + w > A.tb_slot_bytes) { R.status = ST_TB_OVERFLOW; break; }
		}
		// ---- source slices (reference wf_next_prep, miniwfa.c:243-259)
		int32_t jx = newH - P.x;    if (jx < 0) jx += P.nH;
		int32_t j1 = newH - P.oe1;  if (j1 < 0) j1 += P.nH;
		int32_t j2 = newH - P.oe2;  if (j2 < 0) j2 += P.nH;
		int32_t jg1 = newH - P.e1;  if (jg1 < 0) jg1 += P.nH;
		int32_t jg2 = newH - P.e2;  if (jg2 < 0) jg2 += P.nH;
		const int32_t r1 = new1 + 1 == P.n1 ? 0 : new1 + 1; // row of penalty s_new - e1 in the E1/F1 ring
		const int32_t r2 = new2 + 1 == P.n2 ? 0 : new2 + 1;
		Src sx, so1, so2, se1, sf1, se2, sf2;
		sx.p  = M.H + jx * W,   sx.lo  = uni(sh.rng_lo[jx]),  sx.hi  = uni(sh.rng_hi[jx]);
		so1.p = M.H + j1 * W,   so1.lo = uni(sh.rng_lo[j1]),  so1.hi = uni(sh.rng_hi[j1]);
		so2.p = M.H + j2 * W,   so2.lo = uni(sh.rng_lo[j2]),  so2.hi = uni(sh.rng_hi[j2]);
		se1.p = M.E1 + r1 * W,  se1.lo = uni(sh.rng_lo[jg1]), se1.hi = uni(sh.rng_hi[jg1]);
		sf1.p = M.F1 + r1 * W,  sf1.lo = se1.lo,              sf1.hi = se1.hi;
		se2.p = M.E2 + r2 * W,  se2.lo = uni(sh.rng_lo[jg2]), se2.hi = uni(sh.rng_hi[jg2]);
		sf2.p = M.F2 + r2 * W,  sf2.lo = se2.lo,              sf2.hi = se2.hi;
		Src tx, to1, to2, te1, tf1, te2, tf2; // shadow sources, same windows
		if (SEG) {
			tx = sx, to1 = so1, to2 = so2, te1 = se1, tf1 = sf1, te2 = se2, tf2 = sf2;
			tx.p = M.sH + jx * W, to1.p = M.sH + j1 * W, to2.p = M.sH + j2 * W;
			te1.p = M.sE1 + r1 * W, tf1.p = M.sF1 + r1 * W, te2.p = M.sE2 + r2 * W, tf2.p = M.sF2 + r2 * W;
		}
		// columns for which no read can leave a source window
		int32_t ilo = max(max(lo, sx.lo), max(max(so1.lo, so2.lo), max(se1.lo, se2.lo)) + 1);
		int32_t ihi = min(min(hi, sx.hi), min(min(so1.hi, so2.hi), min(se1.hi, se2.hi)) - 1);
		int32_t *dH = M.H + newH * W, *dE1 = M.E1 + new1 * W, *dF1 = M.F1 + new1 * W, *dE2 = M.E2 + new2 * W, *dF2 = M.F2 + new2 * W;
		int32_t *uH = 0, *uE1 = 0, *uF1 = 0, *uE2 = 0, *uF2 = 0;
		if (SEG) uH = M.sH + newH * W, uE1 = M.sE1 + new1 * W, uF1 = M.sF1 + new1 * W, uE2 = M.sE2 + new2 * W, uF2 = M.sF2 + new2 * W;
		uint8_t *tbrow = TB ? M.tb + tb_used - lo : 0; // tbrow[c] is the byte of column c
		// a shrink at the next multiple of 256 can still see this slice (miniwfa.c:148-154)
		const bool track_good = (((256 - (s_new & 255)) & 255) < P.nH);
		unsigned long long *gword = M.good + (int64_t)newH * A.GW;

		if (tid == 0) {
			sh.rng_lo[newH] = lo, sh.rng_hi[newH] = hi;
			// clear the flag set of the NEXT penalty: its last readers passed the previous barrier, its next
			// writers start after this penalty's barrier (clearing the current set here would race with them)
			const int32_t nn = npar + 1 == 3 ? 0 : npar + 1;
			sh.flags[nn][0] = sh.flags[nn][1] = sh.flags[nn][2] = sh.flags[nn][3] = 0;
			if (TB) M.row_off[s_new - 1] = tb_used, M.row_lo[s_new - 1] = lo;
			if (trace_band && s_new - 1 < A.dbg_cap) M.dbg[2 * (s_new - 1)] = lo, M.dbg[2 * (s_new - 1) + 1] = hi;
		}

		// ---- advance + extend over the window, 64 columns per wave per trip
		for (int32_t c0 = (lo & ~63) + wave0; c0 <= hi; c0 += T) {
			const int32_t c = c0 + lane;
			Cell v, u;
			bool active = true;
			if (c0 >= ilo && c0 + 63 <= ihi) { // interior wave: plain coalesced loads
				v = wf_cell(sx.at(c), so1.at(c - 1), se1.at(c - 1), so2.at(c - 1), se2.at(c - 1),
				            so1.at(c + 1), sf1.at(c + 1), so2.at(c + 1), sf2.at(c + 1));
				if (SEG) u = shadow_cell(v.tb, tx.at(c), to1.at(c - 1), te1.at(c - 1), to2.at(c - 1), te2.at(c - 1),
				                         to1.at(c + 1), tf1.at(c + 1), to2.at(c + 1), tf2.at(c + 1));
			} else {
				active = c >= lo && c <= hi;
				const int32_t cc = active ? c : lo; // keep idle lanes on a valid address
				v = wf_cell(sx.rd(cc), so1.rd(cc - 1), se1.rd(cc - 1), so2.rd(cc - 1), se2.rd(cc - 1),
				            so1.rd(cc + 1), sf1.rd(cc + 1), so2.rd(cc + 1), sf2.rd(cc + 1));
				if (SEG) u = shadow_cell(v.tb, tx.rd(cc), to1.rd(cc - 1), te1.rd(cc - 1), to2.rd(cc - 1), te2.rd(cc - 1),
				                         to1.rd(cc + 1), tf1.rd(cc + 1), to2.rd(cc + 1), tf2.rd(cc + 1));
			}
			const int32_t d = c - 1 - tl;
			if (track_good) {
				const bool g = active && (in_matrix(d, v.h, tl, ql) || in_matrix(d, v.e1, tl, ql) || in_matrix(d, v.f1, tl, ql) ||
				                          in_matrix(d, v.e2, tl, ql) || in_matrix(d, v.f2, tl, ql));
				const unsigned long long m = __ballot(g);
				if (lane == 0) gword[c0 >> 6] = m;
			}
			if (active) {
				// edge rule (miniwfa.c:325-326): H is the max of the five, so "any of them live" == "H live"
				if (c == lo && v.h >= -1) sh.flags[npar][0] = 1;
				if (c == hi && v.h >= -1) sh.flags[npar][1] = 1;
				int32_t k = v.h;
				if (in_matrix(d, k, tl, ql)) { // extension sweep of the next iteration (miniwfa.c:400-411)
					k = extend_run(M.ts, M.qs, tl, ql, k, d);
					if (k == tl - 1 && d + k == ql - 1) {
						sh.flags[npar][2] = 1;
						sh.flags[npar][3] = SEG ? u.h : (k == v.h ? (int32_t)(v.tb & 7u) : 0);
					}
				}
				dH[c] = k, dE1[c] = v.e1, dF1[c] = v.f1, dE2[c] = v.e2, dF2[c] = v.f2;
				if (TB) tbrow[c] = (uint8_t)v.tb;
				if (SEG) uH[c] = u.h, uE1[c] = u.e1, uF1[c] = u.f1, uE2[c] = u.e2, uF2[c] = u.f2;
			}
		}
		__syncthreads();

		// ---- bookkeeping, identical on every thread
		if (uni(sh.flags[npar][0])) wf_lo = lo;
		if (uni(sh.flags[npar][1])) wf_hi = hi;
		const int32_t done = uni(sh.flags[npar][2]), payload = uni(sh.flags[npar][3]);
		s = s_new, curH = newH, cur1 = new1, cur2 = new2, par = npar;
		if (TB) tb_used += w;
		if ((s & 0xff) == 0) { // shrink (reference wf_stripe_shrink, miniwfa.c:144-171)
			if (tid == 0) sh.red[0] = 0x7fffffff, sh.red[1] = -1;
			__syncthreads();
			const int32_t wfirst = wf_lo >> 6, wlast = wf_hi >> 6;
			for (int32_t wi = wfirst + tid; wi <= wlast; wi += T) {
				unsigned long long m = 0;
				for (int32_t j = 0; j < P.nH; ++j)
					m |= M.good[(int64_t)j * A.GW + wi] & window_mask(wi << 6, sh.rng_lo[j], sh.rng_hi[j]);
				m &= window_mask(wi << 6, wf_lo, wf_hi);
				if (m) {
					atomicMin(&sh.red[0], (wi << 6) + (int32_t)__builtin_ctzll(m));
					atomicMax(&sh.red[1], (wi << 6) + 63 - (int32_t)__builtin_clzll(m));
				}
			}
			__syncthreads();
			const int32_t glo = uni(sh.red[0]), ghi = uni(sh.red[1]);
			if (ghi < 0) { R.status = ST_INTERNAL; break; } // reference asserts this cannot happen (:157,169)
			wf_lo = glo, wf_hi = ghi;
		}
		cells += w;
		if (!SEG && ((A.max_iter > 0 && cells > A.max_iter) || (A.max_s > 0 && s > A.max_s))) { // miniwfa.c:422-425
			R.status = ST_STOPPED;
			break;
		}
		if (done) {
			R.info = payload;
			break;
		}
	}
	R.s = s, R.cells = cells, R.n_snap = n_snap;
	return R;
}


// ---------------------------------------------------------------------------------------------------------------------
// The same pass with four columns per lane (the band kernel's inner code, mwf_band.hip) for every mode except the
// low-memory first pass.  forward_pass above moves 9 dwords in and 5 out per lane per cell; here a lane owns four
// consecutive columns of a 256-column chunk (chunk g belongs to wave g mod waves), so each array-slice is ONE 16-byte
// load or store per lane, the d-1 / d+1 neighbours come from the adjacent lane through a DPP wave shift and only
// lanes 0 and 63 load a neighbouring chunk's outer column.  Same 48 algorithmic bytes per cell, a quarter of the
// memory instructions, four times the bytes in flight per wave.
__device__ __forceinline__ int32_t from_left(int32_t v, int32_t fill) { return __builtin_amdgcn_update_dpp(fill, v, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ int32_t from_right(int32_t v, int32_t fill) { return __builtin_amdgcn_update_dpp(fill, v, 0x130, 0xf, 0xf, false); }

__device__ __forceinline__ uint32_t probe4g(const PairMem &M, int32_t j, int32_t i)
{
	uint32_t a, b;
	__builtin_memcpy(&a, M.ts + j, 4);
	__builtin_memcpy(&b, M.qs + i, 4);
	return a ^ b;
}

#ifndef MWF_H16_RELAX
#define MWF_H16_RELAX 1 // 16-bit-ring kernel: the last chunk's stores may cross the per-penalty barrier (0: drain everything)
#endif
// ---- packed 16-bit arithmetic on the codes of the 16-bit ring rows (two columns per register, one VOP3P instruction each)
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
#define MWF_BC(T, v) __builtin_bit_cast(T, v)
__device__ __forceinline__ int32_t pk_maxu(int32_t a, int32_t b) { return MWF_BC(int32_t, __builtin_elementwise_max(MWF_BC(u16x2, a), MWF_BC(u16x2, b))); }
__device__ __forceinline__ int32_t pk_minu(int32_t a, int32_t b) { return MWF_BC(int32_t, __builtin_elementwise_min(MWF_BC(u16x2, a), MWF_BC(u16x2, b))); }
__device__ __forceinline__ int32_t pk_add(int32_t a, int32_t b) { return MWF_BC(int32_t, (u16x2)(MWF_BC(u16x2, a) + MWF_BC(u16x2, b))); }
__device__ __forceinline__ int32_t pk_sub(int32_t a, int32_t b) { return MWF_BC(int32_t, (u16x2)(MWF_BC(u16x2, a) - MWF_BC(u16x2, b))); }
__device__ __forceinline__ int32_t pk_subsat(int32_t a, int32_t b) { return MWF_BC(int32_t, __builtin_elementwise_sub_sat(MWF_BC(u16x2, a), MWF_BC(u16x2, b))); } // max(a - b, 0)
__device__ __forceinline__ int32_t pk_mad(int32_t a, int32_t b, int32_t c)
{
	int32_t m;
	asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(m) : "v"(a), "v"(b), "v"(c));
	return m;
}
__device__ __forceinline__ int32_t pk_ne1(int32_t a, int32_t b) // 1 in every half where a != b
{
	int32_t m;
	asm("v_xor_b32 %0, %1, %2\n\tv_pk_min_u16 %0, %0, 1 op_sel_hi:[1,0]" : "=&v"(m) : "v"(a), "v"(b));
	return m;
}
__device__ __forceinline__ int32_t pk_nonzero_mask(int32_t x) // 0xffff in every half of x that is not zero
{
	int32_t m;
	asm("v_pk_min_u16 %0, %1, 1 op_sel_hi:[1,0]\n\tv_pk_sub_i16 %0, 0, %0 op_sel_hi:[0,1]" : "=&v"(m) : "v"(x));
	return m;
}
__device__ __forceinline__ int32_t both16(int32_t v) { return (int32_t)(((uint32_t)v << 16) | ((uint32_t)v & 0xffffu)); }
// code + 1 for a live code (>= 2), 0 for a dead one (0, or the 1 a dead code picked up): the collapse of what outlives a penalty
__device__ __forceinline__ int32_t pk_live_inc(int32_t x)
{
	const int32_t t = pk_subsat(x, 0x00010001);
	return pk_mad(pk_minu(t, 0x00010001), 0x00020002, t);
}

// ---- 2-bit sequence copies in GLOBAL memory (the kernel with 16-bit ring rows, pairs of plain A/C/G/T): sixteen bases per dword, base j
// at bits 2*(j & 15) of dword j >> 4.  The byte sequences of the pairs resident on an XCD (64 x 100 KB for the 50 kb configuration) do
// not fit its 4 MB of L2 and every first probe of the match extension — two unaligned loads per cell — went out to the fabric (a third
// of the kernel's fetch traffic); a quarter of the bytes does fit, and a probe then looks at sixteen bases
// instead of four, so the per-lane walk is entered for the cells near the alignment path only.
__device__ __forceinline__ uint32_t seq16g(const uint32_t *p2, int32_t j)
{
	uint32_t w[2];
	__builtin_memcpy(w, p2 + (j >> 4), 8); // one global_load_dwordx2 (4-byte aligned)
	return __builtin_amdgcn_alignbit(w[1], w[0], (uint32_t)j << 1);
}
// Bytes -> 2 bits per base (two dwords of slack behind the last base); nonzero when a byte is not one of A, C, G, T.
// code = (byte >> 1) & 3: A 0, C 1, T 2, G 3.
template <int T>
__device__ __forceinline__ uint32_t pack2bit_global(const uint8_t *src, int32_t len, uint32_t *dst)
{
	uint32_t bad = 0;
	const int32_t n_dw = (len >> 4) + 2;
	for (int32_t w = threadIdx.x; w < n_dw; w += T) {
		uint32_t out = 0;
#pragma unroll 1
		for (int32_t k = 0, b0 = w << 4; k < 16 && b0 + k < len; ++k) {
			const uint32_t x = src[b0 + k], code = (x >> 1) & 3u;
			bad |= x ^ ((0x47544341u >> (8 * code)) & 0xffu);
			out |= code << (2 * k);
		}
		dst[w] = out;
	}
	return bad;
}

__device__ __forceinline__ uint32_t inm_bit(int32_t d, int32_t k, int32_t tl, int32_t ql)
{
	return (uint32_t)((uint32_t)(k + 1) < (uint32_t)(tl + 1)) & (uint32_t)((uint32_t)(d + k + 1) < (uint32_t)(ql + 1));
}

__device__ __forceinline__ int32_t pick4(int32_t i, int32_t a0, int32_t a1, int32_t a2, int32_t a3)
{
	return i == 0 ? a0 : i == 1 ? a1 : i == 2 ? a2 : a3;
}

// lanes l of interleaved good word k (column = base + 4*l + k) whose column lies in [a,b]
__device__ __forceinline__ unsigned long long lane_mask4(int32_t base, int32_t k, int32_t a, int32_t b)
{
	int32_t lmin = a - base - k, lmax = b - base - k;
	if (lmax < 0) return 0ull;
	lmin = lmin <= 0 ? 0 : (lmin + 3) >> 2;
	lmax = min(lmax >> 2, 63);
	if (lmin > lmax) return 0ull;
	return (~0ull >> (63 - lmax)) & (~0ull << lmin);
}

// exact-match run t[j..] == q[i..] walked by the whole wave, 256 bytes per trip (arguments wave-uniform)
__device__ __forceinline__ int32_t run_wave_g(const PairMem &M, int32_t j, int32_t i, int32_t room, int32_t n0)
{
	const int32_t lane = threadIdx.x & 63;
	int32_t n = n0;
	while (n < room) {
		const int32_t off = n + 4 * lane;
		int32_t m = 0;
		if (off < room) {
			const uint32_t x = probe4g(M, j + off, i + off);
			m = min(x ? (int32_t)(__builtin_ctz(x) >> 3) : 4, room - off);
		}
		const unsigned long long stop = __ballot(m < 4);
		if (stop == 0) { n += 256; continue; }
		const int32_t first = (int32_t)__builtin_ctzll(stop);
		n += 4 * first + __builtin_amdgcn_readlane(m, first);
		break;
	}
	return min(n, room);
}

// LDS2 (gap-extension penalty e2 == 1 only): E2/F2 are written at one penalty and read at the next and never again, so
// while the window fits A.lds_e2_cols columns they live in LDS, updated in place (column c -> slot c mod cols; a chunk's
// slots are only ever touched by the wave that owns the chunk), and never see HBM: 32 instead of 48 bytes per cell.
// What a neighbouring wave needs of them — the chunk's first F2 and last E2 — goes through a small table with one copy
// per penalty parity.  When the window outgrows the LDS span the pass carries on in the HBM rows (and comes back).
extern __shared__ __attribute__((aligned(16))) int32_t lds_e2f2[];

// SEG: the low-memory first pass (shadow ring + snapshots, no traceback bytes, no stop rules, miniwfa.c:569-589) in the same
// four-columns-per-lane form: every shadow array-slice is one 16-byte load or store per lane as well.
// H16: the ring rows hold 16-bit codes instead of 32-bit offsets (half the HBM traffic of this HBM-bound kernel): a live offset
// k >= -1 is stored as k + 3, every dead one (k < -1: NEG_INF plus whatever drift) as 0, which reads back as -3 — dead, and still
// dead after the one or two increments a penalty can add before the value is stored (and collapsed) again.  Dead values only ever
// meet comparisons whose outcome does not depend on how dead they are (and traceback bytes of dead cells are never visited), so
// s, n_iter and the CIGAR are unchanged.  Offsets up to 65532 fit: the pass gives up (ST_BAND_OVERFLOW, re-run with 32-bit rows)
// when target length + penalty could exceed that.
__device__ __forceinline__ int32_t dec16(uint32_t u) { return (int32_t)u - 3; }
__device__ __forceinline__ uint32_t enc16(int32_t k) { return k < -1 ? 0u : (uint32_t)(k + 3); }

template <int T, bool TB, bool LDS2, bool SEG = false, bool H16 = false, typename ArgsT = BatchArgs>
__device__ PassResult stream_pass(const ArgsT &A, const PairMem &M, Shared &sh, int32_t n_seg, bool trace_band)
{
	static_assert(!(SEG && (TB || LDS2)), "the low-memory first pass stores no traceback and keeps every array in HBM");
	static_assert(!(SEG && H16), "16-bit ring rows: not in the low-memory first pass");
	using Raw4 = std::conditional_t<H16, uint2, int4>; // four columns of a ring row as they lie in memory
	constexpr int ESH = H16 ? 1 : 2;                    // log2(bytes per element)
	const int64_t Wrow = A.W;
	auto rowp = [&](const int32_t *base, int32_t r) -> char* { return (char*)base + (((int64_t)r * Wrow) << ESH); };
	auto ld4 = [&](const char *row, int32_t c) -> Raw4 { return *(const Raw4*)(row + ((int64_t)c << ESH)); };
	auto ld1 = [&](const char *row, int32_t c) -> int32_t {
		if constexpr (H16) return (int32_t)*(const uint16_t*)(row + ((int64_t)c << 1)); // (the code itself: the packed column code works on codes)
		else return *(const int32_t*)(row + ((int64_t)c << 2));
	};
	auto unpack4 = [&](const Raw4 &v, int32_t *o) {
		if constexpr (H16) o[0] = dec16(v.x & 0xffffu), o[1] = dec16(v.x >> 16), o[2] = dec16(v.y & 0xffffu), o[3] = dec16(v.y >> 16);
		else o[0] = v.x, o[1] = v.y, o[2] = v.z, o[3] = v.w;
	};
	auto st4 = [&](char *row, int32_t c, const int32_t *v) {
		if constexpr (H16) *(uint2*)(row + ((int64_t)c << 1)) = make_uint2(enc16(v[0]) | enc16(v[1]) << 16, enc16(v[2]) | enc16(v[3]) << 16);
		else *(int4*)(row + ((int64_t)c << 2)) = make_int4(v[0], v[1], v[2], v[3]);
	};
	auto st1 = [&](char *row, int32_t c, int32_t v) {
		if constexpr (H16) *(uint16_t*)(row + ((int64_t)c << 1)) = (uint16_t)enc16(v);
		else *(int32_t*)(row + ((int64_t)c << 2)) = v;
	};
	const int32_t cap = LDS2 ? A.lds_e2_cols : 0, cap_mask = cap - 1;
	char *const lE2 = (char*)lds_e2f2, *const lF2 = (char*)lds_e2f2 + ((int64_t)cap << ESH);

	constexpr bool WTB = TB || SEG;
	constexpr int NW = T / 64;
	constexpr int32_t kChunk = 256;
	__shared__ int32_t e2_edge[2][64][2]; // [penalty parity][chunk mod 64]{first column's F2, last column's E2}
	bool prev_in_lds = false; // where the previous penalty left its E2/F2
	const Penalty P = load_penalty(A);
	const int32_t tl = M.tl, ql = M.ql, cmax = tl + ql + 1;
	const int32_t tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6);
	const int64_t W = A.W;
	PassResult R;
	R.status = ST_OK, R.s = 0, R.info = 0, R.n_snap = 0, R.cells = 0;

	// H16: 2-bit copies of the two sequences in the unused upper half of this slot's H rows (16-bit rows fill the lower half);
	// a pair with any other byte goes back to the host like one whose offsets outgrow 16 bits and is re-run with 32-bit rows
	constexpr int FULLG = H16 ? 16 : 4; // bases the first probe of the match extension looks at
	uint32_t *const t2 = H16 ? (uint32_t*)((char*)M.H + (((int64_t)P.nH * Wrow) << 1)) : nullptr;
	uint32_t *const q2 = H16 ? t2 + ((tl >> 4) + 2) : nullptr;
	if constexpr (H16) {
		if (ql > 65532) { R.status = ST_BAND_OVERFLOW; return R; } // (query indices are computed mod 2^16 as well)
		uint32_t bad = pack2bit_global<T>(M.ts, tl, t2);
		bad |= pack2bit_global<T>(M.qs, ql, q2);
		if (__syncthreads_or(bad != 0)) { R.status = ST_BAND_OVERFLOW; return R; }
	}

	// ---- penalty 0 (reference wf_stripe_init, miniwfa.c:103-121) and its extension
	if (tid == 0) {
		for (int32_t j = 0; j < P.nH; ++j) sh.rng_lo[j] = 1, sh.rng_hi[j] = 0;
		for (int32_t j = 0; j < 12; ++j) (&sh.flags[0][0])[j] = 0;
		const int32_t c0 = tl + 1;
		const int32_t k0 = extend_run(M.ts, M.qs, tl, ql, -1, 0);
		st1(rowp(M.H, 0), c0, k0);
		st1(rowp(M.E1, 0), c0, kNegInf), st1(rowp(M.F1, 0), c0, kNegInf), st1(rowp(M.E2, 0), c0, kNegInf), st1(rowp(M.F2, 0), c0, kNegInf);
		if (SEG) {
			M.sH[c0] = -1;
			M.sE1[c0] = M.sF1[c0] = M.sE2[c0] = M.sF2[c0] = kNegInf;
		}
		sh.rng_lo[0] = sh.rng_hi[0] = c0;
		sh.word[1] = k0;
	}
	__syncthreads();
	{
		const int32_t k0 = uni(sh.word[1]);
		if (k0 == tl - 1 && k0 == ql - 1) { R.info = SEG ? -1 : 0; return R; }
	}

	int32_t s = 0, wf_lo = tl + 1, wf_hi = tl + 1;
	int32_t curH = 0, cur1 = 0, cur2 = 0, par = 0, sid = 0;
	int64_t cells = 0, tb_used = 0, snap_used = 0;
	const int32_t step_ = SEG ? A.step : 0;
	const int64_t rows_slot = TB ? A.rows_slot : 0, tb_slot_bytes = TB ? A.tb_slot_bytes : 0;
	const int64_t iter_limit = A.max_iter > 0 ? A.max_iter : INT64_MAX;
	const int32_t s_limit = A.max_s > 0 ? A.max_s : INT32_MAX;
	int32_t snap_ctr = step_ == 1 ? 0 : 1, n_snap = 0; // SEG: (s+1) % step, snapshots taken
	const int32_t cfin = ql + 1; // the end cell (tl-1, ql-1) lies on diagonal ql-tl, i.e. in this column

	for (;;) {
		if (TB && sid < n_seg) { // checkpoint reset of the second pass (miniwfa.c:413-416)
			if (uni(M.seg[2 * sid]) == s) {
				const int32_t c = uni(M.seg[2 * sid + 1]);
				if (c < wf_lo || c > wf_hi) { R.status = ST_INTERNAL; break; }
				wf_lo = wf_hi = c;
				++sid;
			}
		}
		const int32_t lo = wf_lo > 1 ? wf_lo - 1 : 1;       // miniwfa.c:417-418
		const int32_t hi = wf_hi < cmax ? wf_hi + 1 : cmax;
		if (SEG) {
			if (snap_ctr == 0) {
				if (!take_snapshot<T>(A, M, sh, n_snap, snap_used, s, curH, cur1, cur2)) { R.status = ST_SNAP_OVERFLOW; break; }
				++n_snap;
			}
			snap_ctr = snap_ctr + 1 == step_ ? 0 : snap_ctr + 1;
		}
		const int32_t s_new = s + 1;
		const int32_t newH = curH + 1 == P.nH ? 0 : curH + 1;
		const int32_t new1 = cur1 + 1 == P.n1 ? 0 : cur1 + 1;
		const int32_t new2 = cur2 + 1 == P.n2 ? 0 : cur2 + 1;
		const int32_t npar = par + 1 == 3 ? 0 : par + 1;
		const int32_t origin = lo & ~3;                     // traceback rows start on a dword: one 4-byte store per lane
		const int32_t row_bytes = (hi | 3) - origin + 1;
		if (TB) {
			if (s_new - 1 >= rows_slot) { R.status = ST_ROWS_OVERFLOW; break; }
			if (tb_used + row_bytes > tb_slot_bytes) { R.status = ST_TB_OVERFLOW; break; }
		}
		if (H16 && tl + s_new + 3 > 65532) { R.status = ST_BAND_OVERFLOW; break; } // an offset (a target index, or past the matrix by at most one per penalty) may no longer fit
		// source slices (reference wf_next_prep, miniwfa.c:243-259) and their windows
		int32_t jx = newH - P.x;    if (jx < 0) jx += P.nH;
		int32_t j1 = newH - P.oe1;  if (j1 < 0) j1 += P.nH;
		int32_t j2 = newH - P.oe2;  if (j2 < 0) j2 += P.nH;
		int32_t jg1 = newH - P.e1;  if (jg1 < 0) jg1 += P.nH;
		int32_t jg2 = newH - P.e2;  if (jg2 < 0) jg2 += P.nH;
		const int32_t r1 = new1 + 1 == P.n1 ? 0 : new1 + 1; // row of penalty s_new - e1 in the E1/F1 ring
		const int32_t r2 = new2 + 1 == P.n2 ? 0 : new2 + 1;
		const int32_t xlo = uni(sh.rng_lo[jx]), xhi = uni(sh.rng_hi[jx]);
		const int32_t alo = uni(sh.rng_lo[j1]), ahi = uni(sh.rng_hi[j1]);
		const int32_t blo = uni(sh.rng_lo[j2]), bhi = uni(sh.rng_hi[j2]);
		const int32_t p1lo = uni(sh.rng_lo[jg1]), p1hi = uni(sh.rng_hi[jg1]);
		const int32_t p2lo = uni(sh.rng_lo[jg2]), p2hi = uni(sh.rng_hi[jg2]);
		const int32_t ilo = max(max(lo, xlo), max(max(alo, blo), max(p1lo, p2lo)) + 1);
		const int32_t ihi = min(min(hi, xhi), min(min(ahi, bhi), min(p1hi, p2hi)) - 1);
		const char *sHx = rowp(M.H, jx), *sHa = rowp(M.H, j1), *sHb = rowp(M.H, j2);
		const char *sE1 = rowp(M.E1, r1), *sF1 = rowp(M.F1, r1), *sE2 = rowp(M.E2, r2), *sF2 = rowp(M.F2, r2);
		char *dH = rowp(M.H, newH), *dE1 = rowp(M.E1, new1), *dF1 = rowp(M.F1, new1), *dE2 = rowp(M.E2, new2), *dF2 = rowp(M.F2, new2);
		// SEG: the shadow ring, same rows
		const int32_t *tHx = 0, *tHa = 0, *tHb = 0, *tE1 = 0, *tF1 = 0, *tE2 = 0, *tF2 = 0;
		int32_t *uH = 0, *uE1 = 0, *uF1 = 0, *uE2 = 0, *uF2 = 0;
		if (SEG) {
			tHx = M.sH + jx * W, tHa = M.sH + j1 * W, tHb = M.sH + j2 * W;
			tE1 = M.sE1 + r1 * W, tF1 = M.sF1 + r1 * W, tE2 = M.sE2 + r2 * W, tF2 = M.sF2 + r2 * W;
			uH = M.sH + newH * W, uE1 = M.sE1 + new1 * W, uF1 = M.sF1 + new1 * W, uE2 = M.sE2 + new2 * W, uF2 = M.sF2 + new2 * W;
		}
		const bool track_good = (((256 - (s_new & 255)) & 255) < P.nH);
		// every chunk the window touches is stored whole: those columns must map to distinct LDS slots (and at most 64 chunks)
		const bool cur_in_lds = LDS2 && ((hi | 255) - (lo & ~255) + 1) <= cap;
		const int32_t epar = s_new & 1;

		if (tid == 0) {
			sh.rng_lo[newH] = lo, sh.rng_hi[newH] = hi;
			const int32_t nn = npar + 1 == 3 ? 0 : npar + 1; // flags of the NEXT penalty (see forward_pass)
			sh.flags[nn][0] = sh.flags[nn][1] = sh.flags[nn][2] = sh.flags[nn][3] = 0;
			if (TB) M.row_off[s_new - 1] = tb_used, M.row_lo[s_new - 1] = origin;
			if (trace_band && s_new - 1 < fresh(A).dbg_cap) M.dbg[2 * (s_new - 1)] = lo, M.dbg[2 * (s_new - 1) + 1] = hi;
		}

		const int32_t g_first = lo >> 8, g_last = hi >> 8;
		int32_t g = g_first + (wave - g_first % NW + NW) % NW; // this wave's first chunk of the window
		const int32_t g_mine = g;
		// what a chunk reads from HBM; the next chunk's loads are issued before the current chunk's arithmetic
		struct ChunkIn { Raw4 hx4, a4, b4, e14, f14; int32_t va, vb, vg1; };
		auto issue = [&](int32_t gq, ChunkIn &x) {
			const int32_t c0q = gq * kChunk + 4 * lane;
			x.hx4 = ld4(sHx, c0q), x.a4 = ld4(sHa, c0q), x.b4 = ld4(sHb, c0q);
			x.e14 = ld4(sE1, c0q), x.f14 = ld4(sF1, c0q);
			// the neighbouring chunks' outer columns: lane 0 the column to the left (H for the o-lags, E), lane 63 the one to the right (H, F)
			const int32_t ceq = lane == 0 ? max(c0q - 1, 0) : c0q + 4; // column 0 is a pad
			x.va = ld1(sHa, ceq), x.vb = ld1(sHb, ceq), x.vg1 = ld1(lane == 0 ? sE1 : sF1, ceq);
		};
		ChunkIn in_cur, in_next;
		if (g <= g_last) issue(g, in_cur);
		for (; g <= g_last; g += NW, in_cur = in_next) {
			if (g + NW <= g_last) issue(g + NW, in_next);
			const int32_t cb = g * kChunk, c0 = cb + 4 * lane;
			const bool inner = cb >= ilo && cb + kChunk - 1 <= ihi; // uniform: no window test needed anywhere
			if constexpr (H16) {
				// ---- The 16-bit-ring kernel computes on the CODES, two columns per instruction: a register
				// holds columns (c0,c1) [x] or (c2,c3) [y] as they lie in the rows; a live code is k + 3 >= 2, a dead one 0, the order of
				// codes is the order of offsets, so the recurrence is v_pk_max_u16 and a "+1" that keeps dead dead (pk_live_inc).
				const int32_t ONE = 0x00010001, TWO = 0x00020002;
				int32_t HXx = (int32_t)in_cur.hx4.x, HXy = (int32_t)in_cur.hx4.y, O1x = (int32_t)in_cur.a4.x, O1y = (int32_t)in_cur.a4.y;
				int32_t O2x = (int32_t)in_cur.b4.x, O2y = (int32_t)in_cur.b4.y, E1x = (int32_t)in_cur.e14.x, E1y = (int32_t)in_cur.e14.y;
				int32_t F1x = (int32_t)in_cur.f14.x, F1y = (int32_t)in_cur.f14.y, E2x, E2y, F2x, F2y;
				{
					const Raw4 e = (LDS2 && prev_in_lds) ? ld4(lE2, c0 & cap_mask) : ld4(sE2, c0), f = (LDS2 && prev_in_lds) ? ld4(lF2, c0 & cap_mask) : ld4(sF2, c0);
					E2x = (int32_t)e.x, E2y = (int32_t)e.y, F2x = (int32_t)f.x, F2y = (int32_t)f.y;
				}
				const int32_t ce = lane == 0 ? max(c0 - 1, 0) : c0 + 4;
				int32_t va = in_cur.va, vb = in_cur.vb, vg1 = in_cur.vg1, vg2;
				if (LDS2 && prev_in_lds) vg2 = lane == 0 ? e2_edge[epar ^ 1][(g - 1) & 63][1] : e2_edge[epar ^ 1][(g + 1) & 63][0];
				else vg2 = ld1(lane == 0 ? sE2 : sF2, ce);
				// local columns of the two registers' halves: x = (4l, 4l+1), y = (4l+2, 4l+3)
				const int32_t RX = (int32_t)((uint32_t)(4 * lane) | (uint32_t)(4 * lane + 1) << 16), RY = pk_add(RX, TWO);
				int32_t outx = 0, outy = 0; // 0xffff in the halves of columns outside [lo, hi]
				if (!inner) { // reads outside a source window yield dead (what the reference's pads supply, miniwfa.c:96-99): dead is 0, so a mask is an AND
					auto out_of = [&](int32_t wlo, int32_t whi, int32_t &mx, int32_t &my) { // halves of the columns outside [wlo, whi]
						const int32_t lo_r = both16(min(max(wlo - cb, 0), 256)), hi_r1 = both16(min(max(whi - cb + 1, 0), 256));
						mx = pk_nonzero_mask(pk_subsat(lo_r, RX) | pk_subsat(pk_add(RX, ONE), hi_r1));
						my = pk_nonzero_mask(pk_subsat(lo_r, RY) | pk_subsat(pk_add(RY, ONE), hi_r1));
					};
					int32_t mx, my;
					out_of(xlo, xhi, mx, my), HXx &= ~mx, HXy &= ~my;
					out_of(alo, ahi, mx, my), O1x &= ~mx, O1y &= ~my;
					out_of(blo, bhi, mx, my), O2x &= ~mx, O2y &= ~my;
					out_of(p1lo, p1hi, mx, my), E1x &= ~mx, E1y &= ~my, F1x &= ~mx, F1y &= ~my;
					out_of(p2lo, p2hi, mx, my), E2x &= ~mx, E2y &= ~my, F2x &= ~mx, F2y &= ~my;
					out_of(lo, hi, outx, outy);
					const int32_t cn = lane == 0 ? c0 - 1 : c0 + 4;
					va = ((cn >= alo) & (cn <= ahi)) ? va : 0;
					vb = ((cn >= blo) & (cn <= bhi)) ? vb : 0;
					vg1 = ((cn >= p1lo) & (cn <= p1hi)) ? vg1 : 0;
					vg2 = ((cn >= p2lo) & (cn <= p2hi)) ? vg2 : 0;
				}
				// neighbouring columns: the column to the left of (c0,c1) is (c-1,c0) — y of the lane to the left shifted in —, of (c2,c3): (c1,c2);
				// to the right of (c0,c1): (c1,c2), of (c2,c3): (c3,c4).  Lane 0 / lane 63 take the neighbouring chunk's outer column (both halves of the fill).
				const int32_t fa = both16(va), fb = both16(vb), fg1 = both16(vg1), fg2 = both16(vg2);
				const int32_t O1Lx = __builtin_amdgcn_alignbit(O1x, from_left(O1y, fa), 16), O1M = __builtin_amdgcn_alignbit(O1y, O1x, 16), O1Ry = __builtin_amdgcn_alignbit(from_right(O1x, fa), O1y, 16);
				const int32_t O2Lx = __builtin_amdgcn_alignbit(O2x, from_left(O2y, fb), 16), O2M = __builtin_amdgcn_alignbit(O2y, O2x, 16), O2Ry = __builtin_amdgcn_alignbit(from_right(O2x, fb), O2y, 16);
				const int32_t E1Lx = __builtin_amdgcn_alignbit(E1x, from_left(E1y, fg1), 16), E1Ly = __builtin_amdgcn_alignbit(E1y, E1x, 16);
				const int32_t E2Lx = __builtin_amdgcn_alignbit(E2x, from_left(E2y, fg2), 16), E2Ly = __builtin_amdgcn_alignbit(E2y, E2x, 16);
				const int32_t F1Rx = __builtin_amdgcn_alignbit(F1y, F1x, 16), F1Ry = __builtin_amdgcn_alignbit(from_right(F1x, fg1), F1y, 16);
				const int32_t F2Rx = __builtin_amdgcn_alignbit(F2y, F2x, 16), F2Ry = __builtin_amdgcn_alignbit(from_right(F2x, fg2), F2y, 16);
				// ---- recurrence (dev::wf_cell, miniwfa.c:267-278)
				int32_t e1x = pk_maxu(O1Lx, E1Lx), e1y = pk_maxu(O1M, E1Ly), e2x = pk_maxu(O2Lx, E2Lx), e2y = pk_maxu(O2M, E2Ly);
				const int32_t pf1x = pk_maxu(O1M, F1Rx), pf1y = pk_maxu(O1Ry, F1Ry), pf2x = pk_maxu(O2M, F2Rx), pf2y = pk_maxu(O2Ry, F2Ry); // F before its + 1
				int32_t f1x = pk_live_inc(pf1x), f1y = pk_live_inc(pf1y), f2x = pk_live_inc(pf2x), f2y = pk_live_inc(pf2y);
				const int32_t mx_ = pk_live_inc(HXx), my_ = pk_live_inc(HXy);
				int32_t hx_ = pk_maxu(pk_maxu(mx_, pk_maxu(e1x, e2x)), pk_maxu(f1x, f2x)), hy_ = pk_maxu(pk_maxu(my_, pk_maxu(e1y, e2y)), pk_maxu(f1y, f2y));
				uint32_t tbw = 0;
				if (WTB) { // the byte from the results (miniwfa.c:289-306), as in the packed band kernel: z = nm (1 + ne1 (2 + ne2 (2 nf1 - 1))) + extension bits
					const int32_t NEG1 = (int32_t)0xffffffffu, EIGHT = 0x00080008, C16 = 0x00100010, C32 = 0x00200020, C64 = 0x00400040;
					int32_t zx = pk_mad(pk_ne1(hx_, f1x), TWO, NEG1), zy = pk_mad(pk_ne1(hy_, f1y), TWO, NEG1);
					zx = pk_mad(pk_ne1(hx_, e2x), zx, TWO), zy = pk_mad(pk_ne1(hy_, e2y), zy, TWO);
					zx = pk_mad(pk_ne1(hx_, e1x), zx, ONE), zy = pk_mad(pk_ne1(hy_, e1y), zy, ONE);
					zx = pk_mad(pk_ne1(hx_, mx_), zx, 0), zy = pk_mad(pk_ne1(hy_, my_), zy, 0);
					zx = pk_mad(pk_ne1(e1x, O1Lx), EIGHT, zx), zy = pk_mad(pk_ne1(e1y, O1M), EIGHT, zy);
					zx = pk_mad(pk_ne1(pf1x, O1M), C16, zx), zy = pk_mad(pk_ne1(pf1y, O1Ry), C16, zy);
					zx = pk_mad(pk_ne1(e2x, O2Lx), C32, zx), zy = pk_mad(pk_ne1(e2y, O2M), C32, zy);
					zx = pk_mad(pk_ne1(pf2x, O2M), C64, zx), zy = pk_mad(pk_ne1(pf2y, O2Ry), C64, zy);
					tbw = __builtin_amdgcn_perm((uint32_t)zy, (uint32_t)zx, 0x06040200u); // bytes c0 c1 c2 c3 from the low bytes of the four halves
				}
				if (!inner) e1x &= ~outx, e1y &= ~outy, e2x &= ~outx, e2y &= ~outy, f1x &= ~outx, f1y &= ~outy, f2x &= ~outx, f2y &= ~outy, hx_ &= ~outx, hy_ &= ~outy;
				// E/F of this penalty: final, store now
				*(uint2*)(dE1 + ((int64_t)c0 << 1)) = make_uint2((uint32_t)e1x, (uint32_t)e1y);
				*(uint2*)(dF1 + ((int64_t)c0 << 1)) = make_uint2((uint32_t)f1x, (uint32_t)f1y);
				if (LDS2 && cur_in_lds) {
					*(uint2*)(lE2 + ((int64_t)(c0 & cap_mask) << 1)) = make_uint2((uint32_t)e2x, (uint32_t)e2y);
					*(uint2*)(lF2 + ((int64_t)(c0 & cap_mask) << 1)) = make_uint2((uint32_t)f2x, (uint32_t)f2y);
					if (lane == 0) e2_edge[epar][g & 63][0] = (int32_t)((uint32_t)f2x & 0xffffu);
					if (lane == 63) e2_edge[epar][g & 63][1] = (int32_t)((uint32_t)e2y >> 16);
				} else {
					*(uint2*)(dE2 + ((int64_t)c0 << 1)) = make_uint2((uint32_t)e2x, (uint32_t)e2y);
					*(uint2*)(dF2 + ((int64_t)c0 << 1)) = make_uint2((uint32_t)f2x, (uint32_t)f2y);
				}
				// edge rule (miniwfa.c:325-326): H is the max of the five, so "any live" == "H live" == its code is not 0; one lane holds the edge column
				uint32_t lv = 0;
				if ((uint32_t)(lo - cb) < (uint32_t)kChunk) {
					const int32_t rel = lo - cb, w = __builtin_amdgcn_readlane((rel & 2) ? hy_ : hx_, rel >> 2);
					lv |= (((uint32_t)w >> ((rel & 1) ? 16 : 0)) & 0xffffu) ? 1u : 0u;
				}
				if ((uint32_t)(hi - cb) < (uint32_t)kChunk) {
					const int32_t rel = hi - cb, w = __builtin_amdgcn_readlane((rel & 2) ? hy_ : hx_, rel >> 2);
					lv |= (((uint32_t)w >> ((rel & 1) ? 16 : 0)) & 0xffffu) ? 2u : 0u;
				}
				// ---- lane geometry: j = k + 1 = code - 2 may reach rj = min(tl, ql - d); query index = j + d, d = c - 1 - tl (mod 2^16: the true value fits)
				const int32_t x0 = min(cmax - c0, 65535), d0 = c0 - 1 - tl;
				const int32_t Xx = (int32_t)(((uint32_t)x0 & 0xffffu) | (uint32_t)(x0 - 1) << 16), TLp = both16(tl);
				const int32_t rjx = pk_minu(Xx, TLp), rjy = pk_minu(pk_sub(Xx, TWO), TLp);
				const int32_t Dx = (int32_t)(((uint32_t)d0 & 0xffffu) | (uint32_t)(d0 + 1) << 16), Dy = pk_add(Dx, TWO);
				uint32_t gbits = 0;
				if (track_good) { // some array holds an in-matrix offset (miniwfa.c:139-142) <=> j <= rj for a live code (dead: j wraps to 65534)
					auto bad = [&](int32_t v, int32_t rj) { return pk_subsat(pk_sub(v, TWO), rj); }; // zero iff good
					const int32_t bx = pk_minu(pk_minu(bad(hx_, rjx), pk_minu(bad(e1x, rjx), bad(f1x, rjx))), pk_minu(bad(e2x, rjx), bad(f2x, rjx))) | outx;
					const int32_t by = pk_minu(pk_minu(bad(hy_, rjy), pk_minu(bad(e1y, rjy), bad(f1y, rjy))), pk_minu(bad(e2y, rjy), bad(f2y, rjy))) | outy;
					gbits = (uint32_t)((bx & 0xffff) == 0) | (uint32_t)(((uint32_t)bx >> 16) == 0) << 1 | (uint32_t)((by & 0xffff) == 0) << 2 | (uint32_t)(((uint32_t)by >> 16) == 0) << 3;
				}
				// ---- match extension, first probe (sixteen bases of the 2-bit copies): j clamped to rj makes room = rj - j zero for dead and phantom offsets
				const int32_t jx_ = pk_minu(pk_sub(hx_, TWO), rjx), jy_ = pk_minu(pk_sub(hy_, TWO), rjy);
				const int32_t iqx = pk_add(jx_, Dx), iqy = pk_add(jy_, Dy);
				uint32_t cnt[4];
#pragma unroll
				for (int u = 0; u < 4; ++u) {
					const uint32_t J = (uint32_t)((u & 2) ? jy_ : jx_), Q = (uint32_t)((u & 2) ? iqy : iqx);
					const int32_t j = (int32_t)((u & 1) ? J >> 16 : J & 0xffffu), q = (int32_t)((u & 1) ? Q >> 16 : Q & 0xffffu);
					const uint32_t x = seq16g(t2, j) ^ seq16g(q2, q);
					cnt[u] = (uint32_t)(__builtin_ffs((int)x) - 1) >> 1; // huge for "no difference"
				}
				typedef unsigned short us2_t __attribute__((ext_vector_type(2)));
				const int32_t cx = MWF_BC(int32_t, (us2_t)__builtin_amdgcn_cvt_pk_u16(cnt[0], cnt[1])), cy = MWF_BC(int32_t, (us2_t)__builtin_amdgcn_cvt_pk_u16(cnt[2], cnt[3])); // saturating
				const int32_t FULLp = both16(16);
				const int32_t m9x = pk_minu(cx, pk_sub(rjx, jx_)), m9y = pk_minu(cy, pk_sub(rjy, jy_)); // > 16: the whole probe matched, room left
				int32_t nmx = pk_minu(m9x, FULLp), nmy = pk_minu(m9y, FULLp);
				const int32_t pendp = pk_subsat(m9x, FULLp) | pk_subsat(m9y, FULLp);
				if (__ballot(pendp != 0)) { // a run of >= 16 matches continues: its lane walks it 8 bytes per trip for four trips, then the whole wave does
					int32_t hv[4] = {(int32_t)((uint32_t)hx_ & 0xffffu) - 3, (int32_t)((uint32_t)hx_ >> 16) - 3, (int32_t)((uint32_t)hy_ & 0xffffu) - 3, (int32_t)((uint32_t)hy_ >> 16) - 3};
					int32_t nmat[4] = {(int32_t)((uint32_t)nmx & 0xffffu), (int32_t)((uint32_t)nmx >> 16), (int32_t)((uint32_t)nmy & 0xffffu), (int32_t)((uint32_t)nmy >> 16)};
					uint32_t pend = (uint32_t)(((uint32_t)m9x & 0xffffu) > 16u) | (uint32_t)(((uint32_t)m9x >> 16) > 16u) << 1 | (uint32_t)(((uint32_t)m9y & 0xffffu) > 16u) << 2 | (uint32_t)(((uint32_t)m9y >> 16) > 16u) << 3;
					uint32_t open = 0;
					while (pend) {
						const int32_t ii = __builtin_ctz(pend);
						const int32_t hh = pick4(ii, hv[0], hv[1], hv[2], hv[3]);
						int32_t n = 16;
						const int32_t j = hh + 1, q = c0 + ii - 1 - tl + j, rm = min(tl - j, ql - q);
						for (int trip = 0; n < rm; ++trip) {
							if (trip == 4) { open |= 1u << ii; break; }
							const uint64_t x = ld8(M.ts + j + n) ^ ld8(M.qs + q + n);
							if (x) { n += (int32_t)(__builtin_ctzll(x) >> 3); break; }
							n += 8;
						}
						n = min(n, rm);
#pragma unroll
						for (int i = 0; i < 4; ++i) nmat[i] = ii == i ? n : nmat[i];
						pend &= pend - 1;
					}
					for (unsigned long long owners = __ballot(open != 0); owners; owners &= owners - 1) {
						const int32_t src = (int32_t)__builtin_ctzll(owners);
						const int32_t c0s = cb + 4 * src;
						for (uint32_t bits = (uint32_t)__builtin_amdgcn_readlane((int32_t)open, src); bits; bits &= bits - 1) {
							const int32_t ii = (int32_t)__builtin_ctz(bits);
							const int32_t hh = __builtin_amdgcn_readlane(pick4(ii, hv[0], hv[1], hv[2], hv[3]), src);
							const int32_t j = hh + 1, q = c0s + ii - 1 - tl + j, rm = min(tl - j, ql - q);
							const int32_t n = run_wave_g(M, j, q, rm, 48);
#pragma unroll
							for (int i = 0; i < 4; ++i) nmat[i] = (ii == i && lane == src) ? n : nmat[i];
						}
					}
					nmx = (int32_t)(((uint32_t)nmat[0] & 0xffffu) | (uint32_t)nmat[1] << 16), nmy = (int32_t)(((uint32_t)nmat[2] & 0xffffu) | (uint32_t)nmat[3] << 16);
				}
				const int32_t hxx = pk_add(hx_, nmx), hxy = pk_add(hy_, nmy); // extended
				// termination test of the extension sweep (miniwfa.c:405-409): only column ql+1 can hold the end cell
				uint32_t fin = 0;
				int32_t done_info = 0;
				if (cfin >= cb && cfin < cb + kChunk && cfin >= lo && cfin <= hi) { // uniform
					const int32_t rel = cfin - cb, sh16 = (rel & 1) ? 16 : 0;
					const int32_t hv = (int32_t)(((uint32_t)((rel & 2) ? hxy : hxx) >> sh16) & 0xffffu) - 3, nm = (int32_t)(((uint32_t)((rel & 2) ? nmy : nmx) >> sh16) & 0xffffu);
					fin = (uint32_t)(lane == (rel >> 2)) & (uint32_t)(hv == tl - 1) & inm_bit(ql - tl, hv - nm, tl, ql);
					done_info = (fin && nm == 0) ? (int32_t)((tbw >> (8 * (rel & 3))) & 7u) : 0;
				}
				*(uint2*)(dH + ((int64_t)c0 << 1)) = make_uint2((uint32_t)hxx, (uint32_t)hxy);
				if (TB && c0 >= origin && c0 <= hi) *(uint32_t*)(M.tb + tb_used - origin + c0) = tbw;
				if (track_good) {
					unsigned long long *gword = M.good + (int64_t)newH * fresh(A).GW + (int64_t)g * 4;
#pragma unroll
					for (int i = 0; i < 4; ++i) {
						const unsigned long long m = __ballot((gbits >> i) & 1u);
						if (lane == 0) gword[i] = m;
					}
				}
				if (lv & 1u) sh.flags[npar][0] = 1;   // uniform; every lane stores the same word
				if (lv & 2u) sh.flags[npar][1] = 1;
				if (__ballot(fin)) {
					if (fin) sh.flags[npar][2] = 1, sh.flags[npar][3] = done_info;
				}
				continue;
			}
			int32_t hx[4], o1[6], o2[6], e1s[4], f1s[4], e2s[4], f2s[4];
			o1[0] = o1[5] = o2[0] = o2[5] = 0;
			unpack4(in_cur.hx4, hx), unpack4(in_cur.a4, o1 + 1), unpack4(in_cur.b4, o2 + 1), unpack4(in_cur.e14, e1s), unpack4(in_cur.f14, f1s);
			if (LDS2 && prev_in_lds) unpack4(ld4(lE2, c0 & cap_mask), e2s), unpack4(ld4(lF2, c0 & cap_mask), f2s); // (the LDS copy of E2/F2 is coded like the rows)
			else unpack4(ld4(sE2, c0), e2s), unpack4(ld4(sF2, c0), f2s);
			const int32_t ce = lane == 0 ? max(c0 - 1, 0) : c0 + 4;
			int32_t va = in_cur.va, vb = in_cur.vb, vg1 = in_cur.vg1, vg2;
			if (LDS2 && prev_in_lds) vg2 = lane == 0 ? e2_edge[epar ^ 1][(g - 1) & 63][1] : e2_edge[epar ^ 1][(g + 1) & 63][0];
			else vg2 = ld1(lane == 0 ? sE2 : sF2, ce);
			if (!inner) { // reads outside a source window yield NEG_INF (what the reference's pads supply, miniwfa.c:96-99)
#pragma unroll
				for (int i = 0; i < 4; ++i) {
					const int32_t c = c0 + i;
					hx[i] = ((c >= xlo) & (c <= xhi)) ? hx[i] : kNegInf;
					o1[i + 1] = ((c >= alo) & (c <= ahi)) ? o1[i + 1] : kNegInf;
					o2[i + 1] = ((c >= blo) & (c <= bhi)) ? o2[i + 1] : kNegInf;
					const bool in1 = (c >= p1lo) & (c <= p1hi), in2 = (c >= p2lo) & (c <= p2hi);
					e1s[i] = in1 ? e1s[i] : kNegInf, f1s[i] = in1 ? f1s[i] : kNegInf;
					e2s[i] = in2 ? e2s[i] : kNegInf, f2s[i] = in2 ? f2s[i] : kNegInf;
				}
				const int32_t cn = lane == 0 ? c0 - 1 : c0 + 4;
				va = ((cn >= alo) & (cn <= ahi)) ? va : kNegInf;
				vb = ((cn >= blo) & (cn <= bhi)) ? vb : kNegInf;
				vg1 = ((cn >= p1lo) & (cn <= p1hi)) ? vg1 : kNegInf;
				vg2 = ((cn >= p2lo) & (cn <= p2hi)) ? vg2 : kNegInf;
			}
			o1[0] = from_left(o1[4], va), o1[5] = from_right(o1[1], va);
			o2[0] = from_left(o2[4], vb), o2[5] = from_right(o2[1], vb);
			int32_t g1m[4], g1p[4], g2m[4], g2p[4]; // E of column c-1, F of column c+1
			g1m[0] = from_left(e1s[3], vg1), g2m[0] = from_left(e2s[3], vg2);
			g1p[3] = from_right(f1s[0], vg1), g2p[3] = from_right(f2s[0], vg2);
#pragma unroll
			for (int i = 1; i < 4; ++i) g1m[i] = e1s[i - 1], g2m[i] = e2s[i - 1];
#pragma unroll
			for (int i = 0; i < 3; ++i) g1p[i] = f1s[i + 1], g2p[i] = f2s[i + 1];
			// SEG: the provenance of the same nine sources
			int32_t thx[4], to1[6], to2[6], tg1m[4], tg1p[4], tg2m[4], tg2p[4];
			if (SEG) {
				const int4 x4 = *(const int4*)(tHx + c0), a4s = *(const int4*)(tHa + c0), b4s = *(const int4*)(tHb + c0);
				const int4 e1q = *(const int4*)(tE1 + c0), f1q = *(const int4*)(tF1 + c0), e2q = *(const int4*)(tE2 + c0), f2q = *(const int4*)(tF2 + c0);
				int32_t wa = tHa[ce], wb = tHb[ce], wg1 = (lane == 0 ? tE1 : tF1)[ce], wg2 = (lane == 0 ? tE2 : tF2)[ce];
				thx[0] = x4.x, thx[1] = x4.y, thx[2] = x4.z, thx[3] = x4.w;
				to1[1] = a4s.x, to1[2] = a4s.y, to1[3] = a4s.z, to1[4] = a4s.w;
				to2[1] = b4s.x, to2[2] = b4s.y, to2[3] = b4s.z, to2[4] = b4s.w;
				int32_t te1[4] = {e1q.x, e1q.y, e1q.z, e1q.w}, tf1[4] = {f1q.x, f1q.y, f1q.z, f1q.w};
				int32_t te2[4] = {e2q.x, e2q.y, e2q.z, e2q.w}, tf2[4] = {f2q.x, f2q.y, f2q.z, f2q.w};
				if (!inner) {
#pragma unroll
					for (int i = 0; i < 4; ++i) {
						const int32_t c = c0 + i;
						thx[i] = ((c >= xlo) & (c <= xhi)) ? thx[i] : kNegInf;
						to1[i + 1] = ((c >= alo) & (c <= ahi)) ? to1[i + 1] : kNegInf;
						to2[i + 1] = ((c >= blo) & (c <= bhi)) ? to2[i + 1] : kNegInf;
						const bool in1 = (c >= p1lo) & (c <= p1hi), in2 = (c >= p2lo) & (c <= p2hi);
						te1[i] = in1 ? te1[i] : kNegInf, tf1[i] = in1 ? tf1[i] : kNegInf;
						te2[i] = in2 ? te2[i] : kNegInf, tf2[i] = in2 ? tf2[i] : kNegInf;
					}
					const int32_t cn = lane == 0 ? c0 - 1 : c0 + 4;
					wa = ((cn >= alo) & (cn <= ahi)) ? wa : kNegInf;
					wb = ((cn >= blo) & (cn <= bhi)) ? wb : kNegInf;
					wg1 = ((cn >= p1lo) & (cn <= p1hi)) ? wg1 : kNegInf;
					wg2 = ((cn >= p2lo) & (cn <= p2hi)) ? wg2 : kNegInf;
				}
				to1[0] = from_left(to1[4], wa), to1[5] = from_right(to1[1], wa);
				to2[0] = from_left(to2[4], wb), to2[5] = from_right(to2[1], wb);
				tg1m[0] = from_left(te1[3], wg1), tg2m[0] = from_left(te2[3], wg2);
				tg1p[3] = from_right(tf1[0], wg1), tg2p[3] = from_right(tf2[0], wg2);
#pragma unroll
				for (int i = 1; i < 4; ++i) tg1m[i] = te1[i - 1], tg2m[i] = te2[i - 1];
#pragma unroll
				for (int i = 0; i < 3; ++i) tg1p[i] = tf1[i + 1], tg2p[i] = tf2[i + 1];
			}

			// ---- the recurrence, then the first 4-byte probe of the match extension, branch-free for all 4 columns
			int32_t hv[4], nmat[4], ne1[4], nf1[4], ne2[4], nf2[4];
			int32_t uh[4], ue1[4], uf1[4], ue2[4], uf2[4]; // SEG: new provenance values
			uint32_t tbw = 0, pend = 0, live = 0, fin = 0, gbits = 0;
			auto columns = [&](auto inner_c) {
				constexpr bool INNER = decltype(inner_c)::value;
#pragma unroll
				for (int i = 0; i < 4; ++i) {
					const int32_t c = c0 + i, d = c - 1 - tl;
					const uint32_t act = INNER ? 1u : (uint32_t)((c >= lo) & (c <= hi));
					const Cell v = wf_cell<WTB>(hx[i], o1[i], g1m[i], o2[i], g2m[i], o1[i + 2], g1p[i], o2[i + 2], g2p[i]);
					ne1[i] = act ? v.e1 : kNegInf, nf1[i] = act ? v.f1 : kNegInf;
					ne2[i] = act ? v.e2 : kNegInf, nf2[i] = act ? v.f2 : kNegInf;
					if (SEG) { // provenance follows the choices the traceback byte records (miniwfa.c:504-523)
						const Cell u = shadow_cell(v.tb, thx[i], to1[i], tg1m[i], to2[i], tg2m[i], to1[i + 2], tg1p[i], to2[i + 2], tg2p[i]);
						uh[i] = u.h, ue1[i] = u.e1, uf1[i] = u.f1, ue2[i] = u.e2, uf2[i] = u.f2;
					}
					const uint32_t inm = act & inm_bit(d, v.h, tl, ql);
					if (track_good) // uniform
						gbits |= (act & (inm | inm_bit(d, v.e1, tl, ql) | inm_bit(d, v.f1, tl, ql) | inm_bit(d, v.e2, tl, ql) | inm_bit(d, v.f2, tl, ql))) << i;
					const int32_t j = inm ? v.h + 1 : 0, q = inm ? d + v.h + 1 : 0;
					const int32_t room = inm ? min(tl - j, ql - q) : 0;
					if constexpr (H16) {
						const uint32_t x = seq16g(t2, j) ^ seq16g(q2, q);
						nmat[i] = min(min((int32_t)((uint32_t)(__builtin_ffs((int)x) - 1) >> 1), 16), room);
						pend |= ((uint32_t)(x == 0) & (uint32_t)(room > 16)) << i;
					} else {
						const uint32_t x = probe4g(M, j, q);
						nmat[i] = min(min((int32_t)((uint32_t)(__builtin_ffs((int)x) - 1) >> 3), 4), room);
						pend |= ((uint32_t)(x == 0) & (uint32_t)(room > 4)) << i;
					}
					hv[i] = v.h;
					tbw |= v.tb << (8 * i);
				}
			};
			if (inner) columns(std::true_type{});
			else columns(std::false_type{});
			// E/F of this penalty: final, store now
			st4(dE1, c0, ne1);
			st4(dF1, c0, nf1);
			if (SEG) { // (whole chunks are stored: columns outside the window are never read back as anything but NEG_INF)
				*(int4*)(uH + c0) = make_int4(uh[0], uh[1], uh[2], uh[3]);
				*(int4*)(uE1 + c0) = make_int4(ue1[0], ue1[1], ue1[2], ue1[3]);
				*(int4*)(uF1 + c0) = make_int4(uf1[0], uf1[1], uf1[2], uf1[3]);
				*(int4*)(uE2 + c0) = make_int4(ue2[0], ue2[1], ue2[2], ue2[3]);
				*(int4*)(uF2 + c0) = make_int4(uf2[0], uf2[1], uf2[2], uf2[3]);
			}
			if (LDS2 && cur_in_lds) {
				st4(lE2, c0 & cap_mask, ne2);
				st4(lF2, c0 & cap_mask, nf2);
				// (H16: whatever outlives the penalty must be collapsed like the coded rows — a dead -3 that picked up its +1 is -2, and
				// one more +1 next penalty would make it "live")
				if (lane == 0) e2_edge[epar][g & 63][0] = H16 ? dec16(enc16(nf2[0])) : nf2[0];
				if (lane == 63) e2_edge[epar][g & 63][1] = H16 ? dec16(enc16(ne2[3])) : ne2[3];
			} else {
				st4(dE2, c0, ne2);
				st4(dF2, c0, nf2);
			}
			if ((uint32_t)(lo - cb) < (uint32_t)kChunk || (uint32_t)(hi - cb) < (uint32_t)kChunk) // this chunk holds an edge column
#pragma unroll
				for (int i = 0; i < 4; ++i) { // edge rule (miniwfa.c:325-326): H is the max of the five, so "any live" == "H live"
					const uint32_t lv = (uint32_t)(hv[i] >= -1);
					live |= (lv & (uint32_t)(c0 + i == lo)) | ((lv & (uint32_t)(c0 + i == hi)) << 1);
				}
			// a run of >= 4 matches continues: its lane walks it 8 bytes per trip for four trips, then the whole wave does
			if (__ballot(pend != 0)) {
				uint32_t open = 0;
				while (pend) {
					const int32_t ii = __builtin_ctz(pend);
					const int32_t hh = pick4(ii, hv[0], hv[1], hv[2], hv[3]);
					int32_t n = FULLG;
					const int32_t j = hh + 1, q = c0 + ii - 1 - tl + j, rm = min(tl - j, ql - q);
					for (int trip = 0; n < rm; ++trip) {
						if (trip == 4) { open |= 1u << ii; break; }
						const uint64_t x = ld8(M.ts + j + n) ^ ld8(M.qs + q + n);
						if (x) { n += (int32_t)(__builtin_ctzll(x) >> 3); break; }
						n += 8;
					}
					n = min(n, rm);
#pragma unroll
					for (int i = 0; i < 4; ++i) nmat[i] = ii == i ? n : nmat[i];
					pend &= pend - 1;
				}
				for (unsigned long long owners = __ballot(open != 0); owners; owners &= owners - 1) {
					const int32_t src = (int32_t)__builtin_ctzll(owners);
					const int32_t c0s = cb + 4 * src;
					for (uint32_t bits = (uint32_t)__builtin_amdgcn_readlane((int32_t)open, src); bits; bits &= bits - 1) {
						const int32_t ii = (int32_t)__builtin_ctz(bits);
						const int32_t hh = __builtin_amdgcn_readlane(pick4(ii, hv[0], hv[1], hv[2], hv[3]), src);
						const int32_t j = hh + 1, q = c0s + ii - 1 - tl + j, rm = min(tl - j, ql - q);
						const int32_t n = run_wave_g(M, j, q, rm, FULLG + 32);
#pragma unroll
						for (int i = 0; i < 4; ++i) nmat[i] = (ii == i && lane == src) ? n : nmat[i];
					}
				}
			}
			// termination test of the extension sweep (miniwfa.c:405-409): only column ql+1 can hold the end cell
			int32_t done_info = 0;
#pragma unroll
			for (int i = 0; i < 4; ++i) hv[i] += nmat[i];
			if (cfin >= cb && cfin < cb + kChunk && cfin >= lo && cfin <= hi) { // uniform
#pragma unroll
				for (int i = 0; i < 4; ++i) {
					const uint32_t f = (uint32_t)(c0 + i == cfin) & (uint32_t)(hv[i] == tl - 1) & inm_bit(ql - tl, hv[i] - nmat[i], tl, ql);
					fin |= f;
					done_info = f ? (SEG ? uh[i] : (nmat[i] == 0 ? (int32_t)((tbw >> (8 * i)) & 7u) : 0)) : done_info;
				}
			}
			st4(dH, c0, hv);
			if (TB && c0 >= origin && c0 <= hi) *(uint32_t*)(M.tb + tb_used - origin + c0) = tbw;
			if (track_good) {
				unsigned long long *gword = M.good + (int64_t)newH * fresh(A).GW + (int64_t)g * 4;
#pragma unroll
				for (int i = 0; i < 4; ++i) {
					const unsigned long long m = __ballot((gbits >> i) & 1u);
					if (lane == 0) gword[i] = m;
				}
			}
			if (__ballot(live & 1u)) sh.flags[npar][0] = 1;   // uniform branches; every lane stores the same word
			if (__ballot(live & 2u)) sh.flags[npar][1] = 1;
			if (__ballot(fin)) {
				if (fin) sh.flags[npar][2] = 1, sh.flags[npar][3] = done_info;
			}
		}
		// rows written now are read by other waves from the next penalty on: drain, then one barrier
		if constexpr (H16) {
			// ... unless nothing written now is loaded before the penalty after next (every H lag and e1 >= 2, E2/F2 in LDS, no good
			// bits for a shrink to read): the last chunk's youngest stores (vmcnt retires in issue order, and
			// every load of this penalty has been waited for) may then stay in flight across this barrier — they are complete at the next
			const bool relax = MWF_H16_RELAX && min(min(P.x, P.oe1), min(P.oe2, P.e1)) >= 2 && cur_in_lds && !track_good && g_mine <= g_last;
			// (three: every chunk issues at least its E1, F1 and H stores, so whatever a wave leaves in flight here is older than the three
			// youngest operations of its next penalty — or that penalty drains everything)
			if (relax) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
			else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
			__builtin_amdgcn_s_barrier();
			asm volatile("" ::: "memory");
		} else {
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			__syncthreads();
		}

		// ---- bookkeeping, identical on every thread
		if (uni(sh.flags[npar][0])) wf_lo = lo;
		if (uni(sh.flags[npar][1])) wf_hi = hi;
		const int32_t done = uni(sh.flags[npar][2]), payload = uni(sh.flags[npar][3]);
		s = s_new, curH = newH, cur1 = new1, cur2 = new2, par = npar;
		prev_in_lds = cur_in_lds;
		if (TB) tb_used += row_bytes;
		if ((s & 0xff) == 0) { // shrink (reference wf_stripe_shrink, miniwfa.c:144-171) on the interleaved good bits
			if (tid == 0) sh.red[0] = 0x7fffffff, sh.red[1] = -1;
			__syncthreads();
			const int32_t gfirst = wf_lo >> 8, n_words = ((wf_hi >> 8) - gfirst + 1) * 4, GWc = fresh(A).GW;
			for (int32_t q = tid; q < n_words; q += T) {
				const int32_t gg = gfirst + (q >> 2), kq = q & 3, base = gg * kChunk;
				unsigned long long m = 0;
				for (int32_t j = 0; j < P.nH; ++j)
					if (sh.rng_lo[j] <= sh.rng_hi[j] && sh.rng_lo[j] <= base + kChunk - 1 && sh.rng_hi[j] >= base) m |= M.good[(int64_t)j * GWc + (int64_t)gg * 4 + kq];
				m &= lane_mask4(base, kq, wf_lo, wf_hi);
				if (m) {
					atomicMin(&sh.red[0], base + 4 * (int32_t)__builtin_ctzll(m) + kq);
					atomicMax(&sh.red[1], base + 4 * (63 - (int32_t)__builtin_clzll(m)) + kq);
				}
			}
			__syncthreads();
			const int32_t glo = uni(sh.red[0]), ghi = uni(sh.red[1]);
			if (ghi < 0) { R.status = ST_INTERNAL; break; }
			wf_lo = glo, wf_hi = ghi;
		}
		cells += hi - lo + 1;
		if (!SEG && (cells > iter_limit || s > s_limit)) { // miniwfa.c:422-425
			R.status = ST_STOPPED;
			break;
		}
		if (done) {
			R.info = payload;
			break;
		}
	}
	R.s = s, R.cells = cells, R.n_snap = n_snap;
	return R;
}

// STREAM: the four-columns-per-lane passes (two kernels rather than one, so that neither pays for the other's registers)
// MODE: -1 score or traceback by A.want_cigar (one kernel for both); 0 / 1: a kernel for score-only / traceback alone, so that
// neither pays for the other's registers (the kernels with E2/F2 in LDS, whose workgroups share a CU)
template <int T, bool STREAM, bool LDS2, bool H16, int MODE, typename ArgsT>
__device__ void align_pair(const ArgsT &A, Shared &sh, int32_t slot, int32_t pair)
{
	PairMem M;
	pair_mem(A, slot, pair, M);
	const bool trace = A.dbg && pair == A.debug_pair;
	int32_t n_seg = 0, status = ST_OK;
	int64_t cells1 = 0;
	PassResult R;
	R.status = ST_OK, R.s = 0, R.info = 0, R.n_snap = 0, R.cells = 0;
	if (!LDS2 && A.step > 0 && A.want_cigar) { // low-memory first pass (reference mwf_wfa_exact, miniwfa.c:610-611)
		PassResult R1;
		if constexpr (STREAM && !LDS2) R1 = stream_pass<T, false, false, true>(A, M, sh, 0, false);
		else R1 = forward_pass<T, false, true>(A, M, sh, 0, false);
		cells1 = R1.cells;
		status = R1.status;
		if (status == ST_OK) {
			if (threadIdx.x == 0) sh.word[2] = trace_checkpoints(A, M, R1.n_snap, R1.info);
			__syncthreads();
			status = uni(sh.word[2]);
			n_seg = R1.n_snap;
		}
		__syncthreads();
	}
	if (status == ST_OK) {
		if constexpr (STREAM && LDS2 && MODE >= 0) R = stream_pass<T, MODE == 1, LDS2, false, LDS2 && H16>(A, M, sh, MODE == 1 ? n_seg : 0, trace);
		else if (STREAM && LDS2 && H16) R = A.want_cigar ? stream_pass<T, true, LDS2, false, LDS2 && H16>(A, M, sh, n_seg, trace) : stream_pass<T, false, LDS2, false, LDS2 && H16>(A, M, sh, 0, trace);
		else if (STREAM) R = A.want_cigar ? stream_pass<T, true, LDS2>(A, M, sh, n_seg, trace) : stream_pass<T, false, LDS2>(A, M, sh, 0, trace);
		else R = A.want_cigar ? forward_pass<T, true, false>(A, M, sh, n_seg, trace) : forward_pass<T, false, false>(A, M, sh, 0, trace);
		status = R.status;
	}
	finish_pair(fresh(A), M, slot, pair, R, status, cells1);
}

// Persistent workgroups: each pulls pairs from a shared counter until the batch is drained.
// (H16: the kernel with 16-bit ring rows and a 64 KB LDS copy of E2/F2 — two 512-thread workgroups per CU, i.e. 128 VGPRs)
template <int T, bool STREAM, bool LDS2 = false, bool H16 = false, int MODE = -1>
__global__ __launch_bounds__(T, (H16 && T == 512) ? 4 : 1) void wfa_batch_kernel(const BatchArgs)
{
	__shared__ Shared sh;
	// the arguments are read from the kernarg segment where they are used (dev::kernel_args / dev::fresh), never held for the kernel's lifetime
	KArgs &A0 = kernel_args();
	for (;;) {
		KArgs &A = fresh(A0);
		if (threadIdx.x == 0) sh.item = (int32_t)atomicAdd(A.queue, 1);
		__syncthreads();
		const int32_t item = uni(sh.item);
		__syncthreads();
		if (item >= A.n_pairs) break;
		const int32_t pair = A.order ? A.order[item] : item;
		align_pair<T, STREAM, LDS2, H16, MODE>(A, sh, (int32_t)blockIdx.x, pair);
	}
}

// Penalty sets whose ring is deeper than the fast kernels' LDS tables (max(x, o1+e1, o2+e2) >= 256; the reference takes any,
// miniwfa.c:390-393): the one-column-per-lane passes with a window table of kBigRing entries.  Every mode: score, traceback,
// low-memory two-pass.  Rare by nature (gap-open costs in the hundreds), so one plain form.
template <int T>
__global__ __launch_bounds__(T, 1) void wfa_bigring_kernel(const BatchArgs)
{
	__shared__ SharedBig sh;
	KArgs &A0 = kernel_args();
	for (;;) {
		KArgs &A = fresh(A0);
		if (threadIdx.x == 0) sh.item = (int32_t)atomicAdd(A.queue, 1);
		__syncthreads();
		const int32_t item = uni(sh.item);
		__syncthreads();
		if (item >= A.n_pairs) break;
		const int32_t pair = A.order ? A.order[item] : item;
		const int32_t slot = (int32_t)blockIdx.x;
		PairMem M;
		pair_mem(A, slot, pair, M);
		const bool trace = A.dbg && pair == A.debug_pair;
		int32_t n_seg = 0, status = ST_OK;
		int64_t cells1 = 0;
		PassResult R;
		R.status = ST_OK, R.s = 0, R.info = 0, R.n_snap = 0, R.cells = 0;
		if (A.step > 0 && A.want_cigar) { // low-memory first pass (reference mwf_wfa_exact, miniwfa.c:610-611)
			const PassResult R1 = forward_pass<T, false, true>(A, M, sh, 0, false);
			cells1 = R1.cells;
			status = R1.status;
			if (status == ST_OK) {
				if (threadIdx.x == 0) sh.word[2] = trace_checkpoints(A, M, R1.n_snap, R1.info);
				__syncthreads();
				status = uni(sh.word[2]);
				n_seg = R1.n_snap;
			}
			__syncthreads();
		}
		if (status == ST_OK) {
			R = A.want_cigar ? forward_pass<T, true, false>(A, M, sh, n_seg, trace) : forward_pass<T, false, false>(A, M, sh, 0, trace);
			status = R.status;
		}
		finish_pair(fresh(A), M, slot, pair, R, status, cells1);
	}
}

} // namespace

// Start of an align call: every pair "not run" (status -1, s -2: not final), CIGAR pool and work queue at zero.
__global__ void reset_kernel(int32_t *status, int32_t *s, int32_t n, unsigned long long *cig_head, int32_t *queue, int32_t n_queue)
{
	const int32_t i = (int32_t)(blockIdx.x * blockDim.x + threadIdx.x);
	if (i < n) status[i] = -1, s[i] = -2;
	if (i < n_queue) queue[i] = 0; // the work counters of the align call's launches (the grid covers n_queue)
	if (i == 0) cig_head[0] = 0ull, cig_head[1] = 0ull, cig_head[2] = 0ull, cig_head[3] = 0ull; // (the second word: flags of the align's kernels for the host, BatchArgs::cig_head + 1; the third and fourth: BatchArgs::retry_count of the two lane classes)
}

int launch_reset(int32_t *status, int32_t *s, int32_t n, unsigned long long *cig_head, int32_t *queue, int32_t n_queue, void *stream)
{
	const int grid = (std::max(n, n_queue) + 255) / 256 > 0 ? (std::max(n, n_queue) + 255) / 256 : 1;
	hipLaunchKernelGGL(reset_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, status, s, n, cig_head, queue, n_queue);
	return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ---- divergence sketch of a device-resident batch (mwf_gpu_batch_wrap: the host never sees the bytes).  The same statistic as
// host::estimate_divergence (mwf_memory.cpp) computes while it packs a batch from host memory: how many of the 8-mers of a query's
// prefix (up to 1500 bases) occur in its target's prefix — (1 - d)^8 plus chance hits.  One workgroup per sampled pair (at most 16:
// pair k n / samples), the 4^8-bit set in LDS; out[k] = hits.  The reference has no size classes to get wrong (one loop serves any
// divergence, miniwfa.c:396-426): this is what lets the classes of a WRAPPED batch follow its divergence too.
constexpr int kSketchK = 8, kSketchLen = 1500;
__device__ __forceinline__ uint32_t sketch_kmer(const uint8_t *p)
{
	uint32_t h = 0;
#pragma unroll
	for (int i = 0; i < kSketchK; ++i) h = (h << 2) | ((p[i] >> 1) & 3u);
	return h;
}
__global__ __launch_bounds__(256) void sketch_kernel(const uint8_t *seqs, const int64_t *t_off, const int32_t *tl, const int64_t *q_off, const int32_t *ql,
                                                      int32_t n, int32_t samples, int32_t *out)
{
	__shared__ uint32_t bits[(1 << (2 * kSketchK)) / 32];
	__shared__ int32_t hits;
	const int32_t i = (int32_t)((int64_t)blockIdx.x * n / samples);
	const int32_t lt = min(tl[i], kSketchLen), lq = min(ql[i], kSketchLen);
	for (int32_t j = threadIdx.x; j < (1 << (2 * kSketchK)) / 32; j += 256) bits[j] = 0;
	if (threadIdx.x == 0) hits = 0;
	__syncthreads();
	const uint8_t *t = seqs + t_off[i], *q = seqs + q_off[i];
	for (int32_t j = threadIdx.x; j + kSketchK <= lt; j += 256) {
		const uint32_t h = sketch_kmer(t + j);
		atomicOr(&bits[h >> 5], 1u << (h & 31));
	}
	__syncthreads();
	int32_t mine = 0;
	for (int32_t j = threadIdx.x; j + kSketchK <= lq; j += 256) {
		const uint32_t h = sketch_kmer(q + j);
		mine += (int32_t)((bits[h >> 5] >> (h & 31)) & 1u);
	}
	if (mine) atomicAdd(&hits, mine);
	__syncthreads();
	if (threadIdx.x == 0) out[blockIdx.x] = hits;
}

int launch_sketch(const uint8_t *seqs, const int64_t *t_off, const int32_t *tl, const int64_t *q_off, const int32_t *ql, int32_t n, int32_t samples, int32_t *out, void *stream)
{
	hipLaunchKernelGGL(sketch_kernel, dim3(samples), dim3(256), 0, (hipStream_t)stream, seqs, t_off, tl, q_off, ql, n, samples, out);
	return hipGetLastError() == hipSuccess ? 0 : -2;
}

// scalar_generic (comparison / fallback) runs the one-column-per-lane kernel, low-memory mode included
static bool wants_stream(const BatchArgs &a) { return !a.scalar_generic; }

template <bool STREAM>
static int launch_batch_as(const BatchArgs &a, int grid, int block, hipStream_t st)
{
	switch (block) {
	case 64:   hipLaunchKernelGGL((wfa_batch_kernel<64, STREAM>),   dim3(grid), dim3(64),   0, st, a); break;
	case 128:  hipLaunchKernelGGL((wfa_batch_kernel<128, STREAM>),  dim3(grid), dim3(128),  0, st, a); break;
	case 256:  hipLaunchKernelGGL((wfa_batch_kernel<256, STREAM>),  dim3(grid), dim3(256),  0, st, a); break;
	case 512:  hipLaunchKernelGGL((wfa_batch_kernel<512, STREAM>),  dim3(grid), dim3(512),  0, st, a); break;
	case 1024: hipLaunchKernelGGL((wfa_batch_kernel<1024, STREAM>), dim3(grid), dim3(1024), 0, st, a); break;
	default: return -1;
	}
	return hipGetLastError() == hipSuccess ? 0 : -2;
}

// E2/F2 in LDS: one 512-thread workgroup per CU with 2 x lds_e2_cols ints of dynamic LDS
static bool wants_lds2(const BatchArgs &a, int block) { return wants_stream(a) && a.lds_e2_cols > 0 && (block == 512 || block == 768) && a.pen.e2 == 1; }

int launch_batch(const BatchArgs &a, int grid, int block, void *stream)
{
	if (a.pen.nH > kMaxRing) { // a ring deeper than the LDS window tables of every other kernel
		if (a.pen.nH > kBigRing) return -1;
		hipLaunchKernelGGL(wfa_bigring_kernel<256>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
		return hipGetLastError() == hipSuccess ? 0 : -2;
	}
	if (wants_lds2(a, block)) {
		const int lds = a.lds_e2_cols * 2 * (a.ring16 ? 2 : 4); // E2 and F2, as 16-bit codes with the 16-bit ring rows
		const int lds_max = a.lds_e2_cols * 2 * 4;
		auto go = [&](auto kernel, int threads) {
			(void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
			(void)hipGetLastError();
			hipLaunchKernelGGL(kernel, dim3(grid), dim3(threads), lds, (hipStream_t)stream, a);
		};
		if (a.want_cigar) {
			if (a.ring16 && block == 768) go(&wfa_batch_kernel<768, true, true, true, 1>, 768);
			else if (a.ring16) go(&wfa_batch_kernel<512, true, true, true, 1>, 512);
			else if (block == 768) go(&wfa_batch_kernel<768, true, true, false, 1>, 768);
			else go(&wfa_batch_kernel<512, true, true, false, 1>, 512);
		} else {
			if (a.ring16 && block == 768) go(&wfa_batch_kernel<768, true, true, true, 0>, 768);
			else if (a.ring16) go(&wfa_batch_kernel<512, true, true, true, 0>, 512);
			else if (block == 768) go(&wfa_batch_kernel<768, true, true, false, 0>, 768);
			else go(&wfa_batch_kernel<512, true, true, false, 0>, 512);
		}
		return hipGetLastError() == hipSuccess ? 0 : -2;
	}
	return wants_stream(a) ? launch_batch_as<true>(a, grid, block, (hipStream_t)stream) : launch_batch_as<false>(a, grid, block, (hipStream_t)stream);
}

template <bool STREAM>
static int occupancy_as(int block)
{
	int n = 0;
	hipError_t e = hipErrorInvalidValue;
	switch (block) {
	case 64:   e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, wfa_batch_kernel<64, STREAM>, 64, 0); break;
	case 128:  e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, wfa_batch_kernel<128, STREAM>, 128, 0); break;
	case 256:  e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, wfa_batch_kernel<256, STREAM>, 256, 0); break;
	case 512:  e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, wfa_batch_kernel<512, STREAM>, 512, 0); break;
	case 1024: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, wfa_batch_kernel<1024, STREAM>, 1024, 0); break;
	default: break;
	}
	return e == hipSuccess ? n : 0;
}

int bigring_kernel_occupancy()
{
	int n = 0;
	return hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, wfa_bigring_kernel<256>, 256, 0) == hipSuccess ? n : 0;
}

int batch_kernel_occupancy(int block, bool stream_pass, int lds_e2_cols, bool ring16)
{
	if (stream_pass && lds_e2_cols > 0 && (block == 512 || block == 768)) {
		int n = 0;
		const size_t lds = (size_t)lds_e2_cols * (ring16 ? 4 : 8);
		hipError_t e;
		// (the traceback kernels: the score-only ones never hold fewer workgroups)
		if (ring16) e = block == 768 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, wfa_batch_kernel<768, true, true, true, 1>, 768, lds)
		                             : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, wfa_batch_kernel<512, true, true, true, 1>, 512, lds);
		else e = block == 768 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, wfa_batch_kernel<768, true, true, false, 1>, 768, lds)
		                      : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, wfa_batch_kernel<512, true, true, false, 1>, 512, lds);
		return e == hipSuccess ? n : 0;
	}
	return stream_pass ? occupancy_as<true>(block) : occupancy_as<false>(block);
}

} // namespace mwf
