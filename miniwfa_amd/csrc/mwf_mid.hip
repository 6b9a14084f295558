// mwf_mid.hip — one workgroup per pair, one diagonal per lane, every wavefront ring in LDS: the kernel for a FEW mid-size pairs
// (a single mwf_wfa_exact call on a pair of a few thousand bases — the reference's own usage, main.c:67-72 — or a handful of them).
//
// Between the lane kernel (mwf_lane.hip: one wave per pair, pairs of up to 400 bases) and the packed band kernel (mwf_band2.hip: a
// wave computes a 256-column chunk per penalty, H rows in HBM) a lone 2 kb pair cost 1.9 us per penalty: ~800 instructions of one or
// two waves plus the HBM round trip of the rows, whatever the window.  Here the lane kernel's design is spread over the waves of a
// workgroup:
//   * a lane owns ONE column (column = diagonal + tl + 1, as everywhere) of a 64-column group; the groups the window touches are dealt
//     round-robin to the T/64 waves, so a penalty costs a wave the recurrence on one cell per group it holds (reference wf_next_basic,
//     miniwfa.c:261-327) plus the match extension (wf_extend1_padded, :212-226) — a window of up to T columns is one pass;
//   * the H ring (nH rows), the E1/F1 rings (e1 + 1 rows each) and the E2/F2 rings (e2 + 1 rows) are rows of int16 over a SPAN of C
//     columns in LDS (offsets of pairs this short fit; a dead cell is max(v, -32768) exactly as in the packed band kernel), with a pad
//     entry either side that always reads dead (the reference's pads, miniwfa.c:96-99).  A lane reads its neighbours' columns
//     straight from the rows: any penalties are served (no template on e1/e2) and nothing crosses lanes or waves but through the rows;
//   * one spare row per E/F ring: the row a penalty writes is never a row it reads, so ONE s_barrier per penalty orders everything
//     (rows written before the last barrier are read, rows written now are read after the next);
//   * every penalty writes its window AND nH columns either side of it (dead), so a row reads as dead beyond the window it was computed
//     for without any window test — a later window reaches at most nH columns beyond it, shrinks included;
//   * the band shrink every 256 penalties (wf_stripe_shrink, miniwfa.c:144-171) works on ballot good bits kept in LDS per ring row and
//     group, masked by each slice's own window;
//   * both sequences sit in LDS — at 2 bits per base for pairs of plain A/C/G/T (sixteen bases per trip of the extension, two LDS
//     instructions), else as bytes (any alphabet, eight per trip); the wave walks the runs together;
//   * traceback bytes go to the slot's arena as rows of C bytes that all start at the span's first column: the shared traceback
//     (mwf_device.h) finds a byte without reading a row table first.
// A pair whose window leaves the span comes back as ST_BAND_OVERFLOW and is re-run on the packed band kernel (finalize()).
// Results are bit-identical to every other kernel (tests/test_gpu_parity.py::test_mid_kernel_*).
#include "mwf_device.h"

namespace mwf {

using namespace dev;

namespace {

extern __shared__ __attribute__((aligned(16))) uint8_t lds_mid[];

constexpr int32_t kDead16 = -32768;

// bookkeeping words in LDS (behind the rows)
struct MidVars {
	int32_t flags[4];     // per penalty mod 3 (+1 spare): bit 0 new lo edge live, bit 1 new hi edge live, bit 2 end cell reached, bits 4.. payload
	int32_t red[2];       // shrink: first / last good column
	int32_t item, word;
	int32_t far, pad[3];  // furthest offset seen at a forecast penalty (dev::window_forecast)
};

// bits of the 64-column group starting at column w0 that fall inside [lo,hi]
__device__ __forceinline__ unsigned long long group_mask(int32_t w0, int32_t lo, int32_t hi)
{
	if (hi < w0 || lo > w0 + 63 || lo > hi) return 0ull;
	unsigned long long m = ~0ull;
	if (lo > w0) m &= ~0ull << (lo - w0);
	if (hi < w0 + 63) m &= ~0ull >> (w0 + 63 - hi);
	return m;
}

// LDS layout of a launch: rows | good bits | window table | bookkeeping | target bytes | query bytes
struct MidLayout {
	int32_t RL;        // int16 entries per row: pad, C columns, pad (rounded up to a multiple of 8 entries)
	int32_t n_rows;
	int32_t good_off, win_off, vars_off, seq_off; // byte offsets
};
__host__ __device__ inline MidLayout mid_layout(int32_t nH, int32_t e1, int32_t e2, int32_t C)
{
	MidLayout L;
	L.RL = (C + 2 + 7) & ~7;
	L.n_rows = nH + 2 * (e1 + 1) + 2 * (e2 + 1);
	int32_t at = L.n_rows * L.RL * 2;
	at = (at + 15) & ~15;
	L.good_off = at, at += nH * (C / 64) * 8;
	L.win_off = at, at += nH * 8;
	at = (at + 15) & ~15;
	L.vars_off = at, at += (int32_t)sizeof(MidVars);
	at = (at + 15) & ~15;
	L.seq_off = at;
	return L;
}

// FOLD (score-only, o1 == x; the packed band kernel's form, mwf_band2.hip): the E1 / F1 rows hold max(E1, H[s-x]) / max(F1, H[s-x]) and the
// row of lag o1+e1 is not read — two LDS reads less per group.  Every penalty writes the window and nH columns either side, and a window moves
// by one column per penalty (a shrink only cuts it): the columns next to the window of penalty s were written at penalty s - e1.
template <int T, bool TB, bool S2, bool FOLD, typename ArgsT>
__device__ PassResult mid_pass(const ArgsT &A, PairMem &M, const MidLayout &L, const uint8_t *lt, const uint8_t *lq, bool trace_band)
{
	constexpr int NW = T / 64;
	const int32_t tl = M.tl, ql = M.ql, cmax = tl + ql + 1;
	const int32_t tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6);
	const int32_t nH = A.pen.nH, n1 = A.pen.e1 + 1, n2 = A.pen.e2 + 1;
	const int32_t dbg_cap = A.dbg_cap;
	const int64_t iter_limit = A.max_iter > 0 ? A.max_iter : INT64_MAX;
	const int32_t s_limit = A.max_s > 0 ? A.max_s : INT32_MAX;
	const int64_t tb_slot_bytes = A.tb_slot_bytes;
	const int32_t C = A.lane_chunks * 64, RL = L.RL, NG = C / 64;
	// the span: C columns around the middle of the diagonals the alignment path runs between (0 and ql - tl); entry 1 of a row is column `left`
	const int32_t center = tl + 1 + (ql - tl) / 2, left = center - C / 2, right = left + C - 1;
	// rows as byte offsets into the dynamic LDS: ring bases, ring sizes, and the rows of the coming penalty — carried from penalty to penalty
	// (one add and one wrap each) instead of being derived from slot numbers (a dozen multiplies per penalty)
	const int32_t RB = RL * 2;
	const int32_t HB = nH * RB, B1 = n1 * RB, B2 = n2 * RB;
	const int32_t bE1 = HB, bF1 = bE1 + B1, bE2 = bF1 + B1, bF2 = bE2 + B2;
	char *const base = (char*)lds_mid;
	unsigned long long *const good = (unsigned long long*)(lds_mid + L.good_off); // [nH][NG]
	int2 *const win = (int2*)(lds_mid + L.win_off);                                 // [nH]: window of the slice each H slot holds
	MidVars &V = *(MidVars*)(lds_mid + L.vars_off);
	PassResult R;
	R.status = ST_OK, R.s = 0, R.info = 0, R.n_snap = 0, R.cells = 0;

	// ---- penalty 0 (reference wf_stripe_init, miniwfa.c:103-121) and its extension
	{
		const uint32_t dd = 0x80008000u;
		const uint4 dead4 = make_uint4(dd, dd, dd, dd);
		for (int32_t j = tid; j < L.n_rows * RL * 2 / 16; j += T) ((uint4*)lds_mid)[j] = dead4;
		for (int32_t j = tid; j < nH; j += T) win[j] = make_int2(1, 0);
		if (tid < 4) V.flags[tid] = 0;
		if (tid == 0) V.far = -1;
	}
	__syncthreads();
	const int32_t c00 = tl + 1;
	if (c00 < left || c00 > right) { R.status = ST_BAND_OVERFLOW; return R; } // (|ql - tl| beyond the span)
	int32_t k0 = 0;
	if (wave == 0) {
		k0 = (S2 ? lds_extend16(lt, lq, 0, 0, min(tl, ql)) : lds_extend8(lt, lq, 0, 0, min(tl, ql))) - 1;
		if (lane == 0) *(int16_t*)(base + (c00 - left + 1) * 2) = (int16_t)k0, win[0] = make_int2(c00, c00), V.word = k0;
	}
	__syncthreads();
	k0 = uni(V.word);
	if (k0 == tl - 1 && k0 == ql - 1) return R;

	int32_t s = 0, wf_lo = c00, wf_hi = c00;
	int32_t curH = 0, par = 0;
	// byte offsets (within their ring) of the rows penalty 1 writes and reads: H of penalties 1, 1-x, 1-(o1+e1), 1-(o2+e2); E/F of 1 and 1-e
	int32_t oN = RB % HB, oX = ((nH + 1 - A.pen.x) % nH) * RB, oA = ((nH + 1 - A.pen.oe1) % nH) * RB, oB = ((nH + 1 - A.pen.oe2) % nH) * RB;
	int32_t oN1 = RB, oR1 = (2 % n1) * RB, oN2 = RB, oR2 = (2 % n2) * RB;
	int64_t cells = 0, tb_used = 0;
	int32_t est_window = 0;
	if (TB) M.tb_stride = C, M.tb_left = left;
	const int32_t cfin = ql + 1; // the end cell (tl-1, ql-1) lies on diagonal ql-tl, i.e. this column
	const int32_t vb = lane * 2;  // a lane's entry idx - 1 = 64 g + lane of a row, in bytes (+ 128 g): entries idx-1, idx, idx+1 at byte offsets 0, 2, 4
	for (;;) {
#ifdef MWF_MID_TIMING // cycles per penalty of one wave (max_iter = -thread): header | groups | flags .. barrier | bookkeeping; groups this wave ran in bits 28..
		const uint64_t tm0 = __builtin_readcyclecounter();
		int n_groups = 0;
#endif
		const int32_t lo = wf_lo > 1 ? wf_lo - 1 : 1;       // miniwfa.c:417-418
		const int32_t hi = wf_hi < cmax ? wf_hi + 1 : cmax;
		const int32_t s_new = s + 1;
		if (lo < left || hi > right || s_new + tl >= 32760) { R.status = ST_BAND_OVERFLOW; break; } // (an offset — a target index, or past the matrix by one per penalty — must fit 16 bits)
		if (TB && tb_used + C > tb_slot_bytes) { R.status = ST_TB_OVERFLOW; break; }
		const int32_t newH = curH + 1 == nH ? 0 : curH + 1;
		const int32_t npar = par + 1 == 3 ? 0 : par + 1;
		const bool track_good = (((256 - (s_new & 255)) & 255) < nH); // a shrink can still see this slice
		if (wave == 0) { // (the whole wave stores the same words: no exec mask to set up)
			win[newH] = make_int2(lo, hi);
			V.flags[npar + 1 == 3 ? 0 : npar + 1] = 0; // the flag word of the NEXT penalty (its last readers passed the previous barrier)
#ifndef MWF_MID_TIMING
			if (trace_band && s_new - 1 < dbg_cap) M.dbg[2 * (s_new - 1)] = lo, M.dbg[2 * (s_new - 1) + 1] = hi;
#endif
		}
		// columns written: the window and nH either side (dead), clamped to the span
		const int32_t g_first = (max(lo - nH, left) - left) >> 6, g_last = (min(hi + nH, right) - left) >> 6;
		uint32_t flags = 0;
		int32_t fin_info = 0;
		const bool forecast = s_new == 64 || s_new == 256 || s_new == 1024; // uniform: look at how far the pair has come (dev::window_forecast)
		int32_t far = kDead16;
#ifdef MWF_MID_TIMING
		const uint64_t tm1 = __builtin_readcyclecounter();
#endif
		for (int32_t g = g_first + ((wave - g_first) & (NW - 1)); g <= g_last; g += NW) {
#ifdef MWF_MID_TIMING
			++n_groups;
#endif
			const int32_t ga = vb + 128 * g, c = left + 64 * g + lane;
			const int32_t d = c - 1 - tl;
			// sources (reference wf_next_prep, miniwfa.c:252-257)
			const char *const pX = base + ga + oX, *const pA = base + ga + oA, *const pB = base + ga + oB;
			const int32_t hx = *(const int16_t*)(pX + 2), o2m = *(const int16_t*)pB, o2p = *(const int16_t*)(pB + 4);
			const int32_t o1m = FOLD ? kDead16 : *(const int16_t*)pA, o1p = FOLD ? kDead16 : *(const int16_t*)(pA + 4);
			const int32_t g1m = *(const int16_t*)(base + ga + (bE1 + oR1)), g1p = *(const int16_t*)(base + ga + (bF1 + oR1) + 4);
			const int32_t g2m = *(const int16_t*)(base + ga + (bE2 + oR2)), g2p = *(const int16_t*)(base + ga + (bF2 + oR2) + 4);
			const bool act = c >= lo && c <= hi;
			const Cell v = wf_cell<TB>(hx, o1m, g1m, o2m, g2m, o1p, g1p, o2p, g2p);
			if (FOLD) *(int16_t*)(base + ga + (bE1 + oN1) + 2) = (int16_t)max(act ? v.e1 : kDead16, hx), *(int16_t*)(base + ga + (bF1 + oN1) + 2) = (int16_t)max(act ? v.f1 : kDead16, hx);
			else *(int16_t*)(base + ga + (bE1 + oN1) + 2) = (int16_t)(act ? max(v.e1, kDead16) : kDead16), *(int16_t*)(base + ga + (bF1 + oN1) + 2) = (int16_t)(act ? max(v.f1, kDead16) : kDead16);
			*(int16_t*)(base + ga + (bE2 + oN2) + 2) = (int16_t)(act ? max(v.e2, kDead16) : kDead16), *(int16_t*)(base + ga + (bF2 + oN2) + 2) = (int16_t)(act ? max(v.f2, kDead16) : kDead16);
			// match extension (reference wf_extend, miniwfa.c:400-411) of the cells inside the matrix
			const bool inm = act && in_matrix(d, v.h, tl, ql);
			const int32_t j = inm ? v.h + 1 : 0, i = inm ? d + j : 0;
			const int32_t nmat = S2 ? lds_extend16(lt, lq, j, i, inm ? min(tl - j, ql - i) : 0) : lds_extend8(lt, lq, j, i, inm ? min(tl - j, ql - i) : 0);
			const int32_t h = act ? max(v.h + nmat, kDead16) : kDead16;
			*(int16_t*)(base + ga + oN + 2) = (int16_t)h;
			far = max(far, h);
			if (TB && act) M.tb[tb_used + (c - left)] = (uint8_t)v.tb;
			if (track_good) { // some array holds an in-matrix offset here (good_diag, miniwfa.c:139-142)
				const bool gd = act && (inm || in_matrix(d, v.e1, tl, ql) || in_matrix(d, v.f1, tl, ql) || in_matrix(d, v.e2, tl, ql) || in_matrix(d, v.f2, tl, ql));
				const unsigned long long m = __ballot(gd);
				if (lane == 0) good[newH * NG + g] = m;
			}
			// edge rule (miniwfa.c:325-326): H is the max of the five, so "any live" == "H live"
			const uint32_t live = (uint32_t)(h >= -1);
			// termination (miniwfa.c:405-409)
			const bool fin = act && c == cfin && h == tl - 1 && in_matrix(ql - tl, h - nmat, tl, ql);
			flags |= (live & (uint32_t)(c == lo)) | ((live & (uint32_t)(c == hi)) << 1) | ((uint32_t)fin << 2);
			fin_info = fin ? (nmat == 0 ? (int32_t)(v.tb & 7u) : 0) : fin_info;
		}
#ifdef MWF_MID_TIMING
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
		const uint64_t tm2 = __builtin_readcyclecounter();
#endif
		if (__ballot(flags != 0)) { // this wave's share of the three per-penalty flags: one LDS atomic per wave that has any
			const unsigned long long fm = __ballot(flags & 4u);
			uint32_t bits = (__ballot(flags & 1u) ? 1u : 0u) | (__ballot(flags & 2u) ? 2u : 0u);
			if (fm) bits |= 4u | (uint32_t)__builtin_amdgcn_readlane(fin_info, (int32_t)__builtin_ctzll(fm)) << 4;
			if (lane == 0) atomicOr((unsigned int*)&V.flags[npar], bits);
		}
		if (forecast) {
			const int32_t m = wave_max(far);
			if (lane == 0 && m >= 0) atomicMax(&V.far, m);
		}
		// the rows of the coming penalty
		oN = oN + RB == HB ? 0 : oN + RB, oX = oX + RB == HB ? 0 : oX + RB, oA = oA + RB == HB ? 0 : oA + RB, oB = oB + RB == HB ? 0 : oB + RB;
		oN1 = oN1 + RB == B1 ? 0 : oN1 + RB, oR1 = oR1 + RB == B1 ? 0 : oR1 + RB, oN2 = oN2 + RB == B2 ? 0 : oN2 + RB, oR2 = oR2 + RB == B2 ? 0 : oR2 + RB;
		__syncthreads();
#ifdef MWF_MID_TIMING
		const uint64_t tm3 = __builtin_readcyclecounter();
#endif
		const uint32_t fl = (uint32_t)uni(V.flags[npar]);
#ifdef MWF_MID_TIMING
		if (trace_band && tid == (A.max_iter < 0 ? (int32_t)-A.max_iter : 0) && s_new - 1 < dbg_cap) {
			const uint64_t tm4 = __builtin_readcyclecounter();
			M.dbg[2 * (s_new - 1)] = (int32_t)(min((uint32_t)(tm1 - tm0), 65535u) | min((uint32_t)(tm2 - tm1), 65535u) << 16);
			M.dbg[2 * (s_new - 1) + 1] = (int32_t)(min((uint32_t)(tm3 - tm2), 4095u) | min((uint32_t)(tm4 - tm3), 65535u) << 12 | (uint32_t)n_groups << 28);
		}
#endif
		if (fl & 1u) wf_lo = lo;
		if (fl & 2u) wf_hi = hi;
		s = s_new, curH = newH, par = npar;
		if (TB) tb_used += C;
		if ((s & 0xff) == 0) { // shrink (reference wf_stripe_shrink, miniwfa.c:144-171) on the good bits of the slices still in the ring
			if (tid == 0) V.red[0] = 0x7fffffff, V.red[1] = -1;
			__syncthreads();
			const int32_t gA = (wf_lo - left) >> 6, gB = (wf_hi - left) >> 6;
			for (int32_t g = gA + tid; g <= gB; g += T) {
				const int32_t w0 = left + 64 * g;
				unsigned long long m = 0;
				for (int32_t j = 0; j < nH; ++j) {
					const int2 w = win[j];
					m |= good[j * NG + g] & group_mask(w0, w.x, w.y);
				}
				m &= group_mask(w0, wf_lo, wf_hi);
				if (m) {
					atomicMin(&V.red[0], w0 + (int32_t)__builtin_ctzll(m));
					atomicMax(&V.red[1], w0 + 63 - (int32_t)__builtin_clzll(m));
				}
			}
			__syncthreads();
			const int32_t glo = uni(V.red[0]), ghi = uni(V.red[1]);
			if (ghi < 0) { R.status = ST_INTERNAL; break; } // the reference asserts this cannot happen (:157,169)
			wf_lo = glo, wf_hi = ghi;
		}
		cells += hi - lo + 1;
		if (cells > iter_limit || s > s_limit) { // miniwfa.c:422-425
			R.status = ST_STOPPED;
			break;
		}
		if (fl & 4u) { R.info = (int32_t)((fl >> 4) & 7u); break; }
		if (forecast) { // will the window outgrow the span? then hand the pair back now, with the estimate
			est_window = window_forecast(s, uni(V.far), tl, ql, C - 2 * nH - 64);
			if (est_window) { R.status = ST_BAND_OVERFLOW; break; }
		}
	}
	R.s = s, R.cells = est_window ? -(int64_t)est_window : cells;
	return R;
}

template <int T, bool TB, bool S2, bool FOLD = false>
__global__ __launch_bounds__(T, 1) void wfa_mid_kernel(const BatchArgs)
{
	// the arguments are read from the kernarg segment where they are used (mwf_device.h): nothing of them stays in SGPRs across the penalties
	KArgs &A = kernel_args();
	const int32_t tid = threadIdx.x;
	const MidLayout L = mid_layout(A.pen.nH, A.pen.e1, A.pen.e2, A.lane_chunks * 64);
	MidVars &V = *(MidVars*)(lds_mid + L.vars_off);
	uint8_t *lt = lds_mid + L.seq_off;
	// (a follow-up launch behind a kernel that handed pairs back: their number is on the device, BatchArgs::n_pairs_dev)
	const int32_t n_dev = A.n_pairs_dev ? min((int32_t)*A.n_pairs_dev, A.n_pairs) : -1;
	for (int32_t round = 0;; ++round) {
		// a work counter, or — queue == null: a launch of one workgroup per pair — pair blockIdx.x and nothing else (no counter to zero first)
		if (tid == 0) V.item = n_dev >= 0 ? (int32_t)(blockIdx.x + round * gridDim.x) : A.queue ? (int32_t)atomicAdd(A.queue, 1) : (round == 0 ? (int32_t)blockIdx.x : A.n_pairs), V.word = 0;
		__syncthreads();
		const int32_t item = uni(V.item);
		__syncthreads();
		if (item >= (n_dev >= 0 ? n_dev : A.n_pairs)) break;
		const int32_t pair = A.order ? A.order[item] : item;
		PairMem M;
		pair_mem(fresh(A), (int32_t)blockIdx.x, pair, M);
		M.tl = uni(M.tl), M.ql = uni(M.ql);
		uint8_t *lq = S2 ? lt + ((M.tl >> 4) + 2) * 4 : lt + ((M.tl + 7) & ~7) + 16;
		PassResult R;
		R.status = ST_OK, R.s = 0, R.info = 0, R.n_snap = 0, R.cells = 0;
		if (S2) { // 2 bits per base; a base other than A/C/G/T: the host re-runs the pair on a byte-wise copy (ST_ALPHABET)
			uint32_t bad = lds_pack2bit<T>(M.ts, M.tl, lt);
			bad |= lds_pack2bit<T>(M.qs, M.ql, lq);
			if (bad) V.word = 1;
			__syncthreads();
			if (uni(V.word)) R.status = ST_ALPHABET;
			__syncthreads();
		} else { // both sequences into LDS as they are, eight bytes per thread and trip (the packed sequence buffer has 64 bytes of slack behind it)
			for (int32_t j = 8 * tid; j < M.tl; j += 8 * T) *(uint64_t*)(lt + j) = ld8(M.ts + j);
			for (int32_t j = 8 * tid; j < M.ql; j += 8 * T) *(uint64_t*)(lq + j) = ld8(M.qs + j);
			__syncthreads();
		}
		const bool trace = A.dbg && pair == A.debug_pair;
		if (R.status == ST_OK) R = mid_pass<T, TB, S2, FOLD>(fresh(A), M, L, lt, lq, trace);
		if (S2) M.t2 = lt, M.q2 = lq; // the traceback's back-match stays on chip
		finish_pair(fresh(A), M, (int32_t)blockIdx.x, pair, R, R.status, 0);
	}
}

template <int T, bool TB, bool S2, bool FOLD = false>
int launch_v(const BatchArgs &a, int grid, int lds, hipStream_t st)
{
	if constexpr (!TB && !FOLD) {
		if (a.band_fold && a.pen.oe1 - a.pen.x == a.pen.e1) return launch_v<T, TB, S2, true>(a, grid, lds, st);
	}
	// beyond 48 KB of dynamic LDS the runtime wants to be told (per device, and this may run on several host threads: on every launch)
	if (lds > 48 * 1024) {
		(void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wfa_mid_kernel<T, TB, S2, FOLD>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
		(void)hipGetLastError();
	}
	hipLaunchKernelGGL((wfa_mid_kernel<T, TB, S2, FOLD>), dim3(grid), dim3(T), lds, st, a);
	return hipGetLastError() == hipSuccess ? 0 : -2;
}

template <int T>
int launch_t(const BatchArgs &a, int grid, int lds, bool seq2, hipStream_t st)
{
	if (a.want_cigar) return seq2 ? launch_v<T, true, true>(a, grid, lds, st) : launch_v<T, true, false>(a, grid, lds, st);
	return seq2 ? launch_v<T, false, true>(a, grid, lds, st) : launch_v<T, false, false>(a, grid, lds, st);
}

} // namespace

// any penalties whose rings fit (offsets are 16-bit: the host admits pairs with target length + penalty bound < 32760 only)
bool mid_supported(const Penalty &p)
{
	return p.x >= 1 && p.e1 >= 1 && p.e2 >= 1 && p.nH <= 64 && p.e1 <= 8 && p.e2 <= 8;
}

// dynamic LDS of a launch with a span of 64 x `groups` columns, where seq_bytes >= (tl rounded up to 8) + 16 + (ql rounded up to 8) + 32
// for every pair of the launch
int mid_lds_bytes(const Penalty &p, int groups, int64_t seq_bytes)
{
	const MidLayout L = mid_layout(p.nH, p.e1, p.e2, 64 * groups);
	return (int)(((int64_t)L.seq_off + seq_bytes + 64 + 15) / 16 * 16);
}

int launch_mid(const BatchArgs &a, int grid, int block, int lds, bool seq2, void *stream)
{
	if (block == 256) return launch_t<256>(a, grid, lds, seq2, (hipStream_t)stream);
	if (block == 512) return launch_t<512>(a, grid, lds, seq2, (hipStream_t)stream);
	if (block == 1024) return launch_t<1024>(a, grid, lds, seq2, (hipStream_t)stream);
	return -1;
}

} // namespace mwf
