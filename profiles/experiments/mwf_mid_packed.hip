// EXPERIMENT (round 4, not built): the mid kernel with TWO packed columns per lane (128-column blocks, v_pk_* recurrence as in mwf_band2.hip).
// Correct (tests/test_gpu_parity.py::test_mid_kernel_* pass with it) but SLOWER for what the kernel is for: one 500 bp pair 0.177 against 0.121 ms,
// 1 kb 0.308-0.354 against 0.228-0.291, 2 kb 0.646-0.771 against 0.510-0.528 — a lone pair leaves waves idle, so what counts is the length of ONE trip of one
// wave (~430 instructions for two columns against ~290 for one; the compiler also turns the neighbour dwords into ds_bpermute), not the number of trips.

// mwf_mid.hip — one workgroup per pair, one diagonal per lane, every wavefront ring in LDS: the kernel for a FEW mid-size pairs
// (a single mwf_wfa_exact call on a pair of a few thousand bases — the reference's own usage, main.c:67-72 — or a handful of them).
//
// Between the lane kernel (mwf_lane.hip: one wave per pair, pairs of up to 400 bases) and the packed band kernel (mwf_band2.hip: a
// wave computes a 256-column chunk per penalty, H rows in HBM) a lone 2 kb pair cost 1.9 us per penalty: ~800 instructions of one or
// two waves plus the HBM round trip of the rows, whatever the window.  Here the lane kernel's design is spread over the waves of a
// workgroup:
//   * a lane owns TWO neighbouring columns (column = diagonal + tl + 1, as everywhere) of a 128-column block, one register per array, and computes on
//     them with the packed 16-bit instructions of mwf_band2.hip (round 4, second version; the first held one column per lane); the blocks the window touches are dealt
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

// ---- packed 16-bit arithmetic on pairs of columns (as in mwf_band2.hip): a register holds column c in its low half and c + 1 in its high half
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
#define MWF_BC(T, v) __builtin_bit_cast(T, v)
__device__ __forceinline__ int32_t pk_max(int32_t a, int32_t b) { return MWF_BC(int32_t, __builtin_elementwise_max(MWF_BC(s16x2, a), MWF_BC(s16x2, b))); }
__device__ __forceinline__ int32_t pk_add(int32_t a, int32_t b) { return MWF_BC(int32_t, (s16x2)(MWF_BC(s16x2, a) + MWF_BC(s16x2, b))); }
__device__ __forceinline__ int32_t pk_subsat(int32_t a, int32_t b) { return MWF_BC(int32_t, __builtin_elementwise_sub_sat(MWF_BC(u16x2, a), MWF_BC(u16x2, b))); } // max(a - b, 0), unsigned halves
__device__ __forceinline__ int32_t pk_nonzero_mask(int32_t x) // 0xffff in every half that is not zero
{
	int32_t m;
	asm("v_pk_min_u16 %0, %1, 1 op_sel_hi:[1,0]\n\tv_pk_sub_i16 %0, 0, %0 op_sel_hi:[0,1]" : "=&v"(m) : "v"(x));
	return m;
}
__device__ __forceinline__ int32_t pk_ne1(int32_t a, int32_t b) // 1 in every half where a != b
{
	int32_t m;
	asm("v_xor_b32 %0, %1, %2\n\tv_pk_min_u16 %0, %0, 1 op_sel_hi:[1,0]" : "=&v"(m) : "v"(a), "v"(b));
	return m;
}
__device__ __forceinline__ int32_t pk_mad(int32_t a, int32_t b, int32_t c) // a * b + c on unsigned halves
{
	int32_t m;
	asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(m) : "v"(a), "v"(b), "v"(c));
	return m;
}
__device__ __forceinline__ int32_t bfi(int32_t mask, int32_t a, int32_t b) { return (a & mask) | (b & ~mask); }
__device__ __forceinline__ int32_t both16(int32_t v) { return (int32_t)(((uint32_t)v << 16) | ((uint32_t)v & 0xffffu)); }
__device__ __forceinline__ int32_t lo16(int32_t v) { return (int32_t)(int16_t)(v & 0xffff); }
__device__ __forceinline__ int32_t hi16(int32_t v) { return v >> 16; }
__device__ __forceinline__ int32_t pair_of(int32_t lo, int32_t hi) { return (int32_t)(((uint32_t)hi << 16) | ((uint32_t)lo & 0xffffu)); }
constexpr int32_t kDeadPair = (int32_t)0x80008000u;

// lanes of a 128-column block starting at column w0 whose column w0 + 2 lane + k (k = 0: the even columns, 1: the odd ones) lies in [lo,hi]
__device__ __forceinline__ unsigned long long block_mask(int32_t w0, int32_t k, int32_t lo, int32_t hi)
{
	if (lo > hi) return 0ull;
	int32_t lmin = lo - w0 - k, lmax = hi - w0 - k;
	if (lmax < 0) return 0ull;
	lmin = lmin <= 0 ? 0 : (lmin + 1) >> 1;
	lmax = min(lmax >> 1, 63);
	if (lmin > lmax) return 0ull;
	return (~0ull >> (63 - lmax)) & (~0ull << lmin);
}

// Two match extensions per lane, walked together (dev::lds_extend16 / lds_extend8 for one): the LDS reads of both columns travel at once.
template <bool S2>
__device__ __forceinline__ void extend_two(const uint8_t *lt, const uint8_t *lq, int32_t j0, int32_t i0, int32_t room0, int32_t j1, int32_t i1, int32_t room1, int32_t &n0, int32_t &n1)
{
	n0 = 0, n1 = 0;
	bool open0 = room0 > 0, open1 = room1 > 0;
	while (__ballot(open0 || open1)) {
		if (S2) {
			const uint32_t x0 = lds_seq16(lt, j0 + n0) ^ lds_seq16(lq, i0 + n0), x1 = lds_seq16(lt, j1 + n1) ^ lds_seq16(lq, i1 + n1);
			n0 += open0 ? (x0 ? (int32_t)(__builtin_ctz(x0) >> 1) : 16) : 0, n1 += open1 ? (x1 ? (int32_t)(__builtin_ctz(x1) >> 1) : 16) : 0;
			open0 = open0 && x0 == 0 && n0 < room0, open1 = open1 && x1 == 0 && n1 < room1;
		} else {
			const uint64_t x0 = lds_ld8(lt, j0 + n0) ^ lds_ld8(lq, i0 + n0), x1 = lds_ld8(lt, j1 + n1) ^ lds_ld8(lq, i1 + n1);
			n0 += open0 ? (x0 ? (int32_t)(__builtin_ctzll(x0) >> 3) : 8) : 0, n1 += open1 ? (x1 ? (int32_t)(__builtin_ctzll(x1) >> 3) : 8) : 0;
			open0 = open0 && x0 == 0 && n0 < room0, open1 = open1 && x1 == 0 && n1 < room1;
		}
	}
	n0 = max(min(n0, room0), 0), n1 = max(min(n1, room1), 0);
}

// LDS layout of a launch: rows | good bits | window table | bookkeeping | target bytes | query bytes
struct MidLayout {
	int32_t RL;        // int16 entries per row: two pad entries, C columns, two pad entries (rounded up to a multiple of 8 entries)
	int32_t n_rows;
	int32_t good_off, win_off, vars_off, seq_off; // byte offsets
};
__host__ __device__ inline MidLayout mid_layout(int32_t nH, int32_t e1, int32_t e2, int32_t C)
{
	MidLayout L;
	L.RL = (C + 4 + 7) & ~7;
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

template <int T, bool TB, bool S2, typename ArgsT>
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
	const int32_t C = A.lane_chunks * 64, RL = L.RL, NG = C / 64; // (C is a multiple of 128: whole blocks)
	// the span: C columns around the middle of the diagonals the alignment path runs between (0 and ql - tl); entry 2 of a row is column `left`
	const int32_t center = tl + 1 + (ql - tl) / 2, left = center - C / 2, right = left + C - 1;
	// rows as byte offsets into the dynamic LDS: ring bases, ring sizes, and the rows of the coming penalty — carried from penalty to penalty
	// (one add and one wrap each) instead of being derived from slot numbers (a dozen multiplies per penalty)
	const int32_t RB = RL * 2;
	const int32_t HB = nH * RB, B1 = n1 * RB, B2 = n2 * RB;
	const int32_t bE1 = HB, bF1 = bE1 + B1, bE2 = bF1 + B1, bF2 = bE2 + B2;
	char *const base = (char*)lds_mid;
	unsigned long long *const good = (unsigned long long*)(lds_mid + L.good_off); // [nH][NG]: per 128-column block the even columns' bits, then the odd ones'
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
		if (lane == 0) *(int16_t*)(base + (c00 - left + 2) * 2) = (int16_t)k0, win[0] = make_int2(c00, c00), V.word = k0;
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
	// A lane owns TWO neighbouring columns of a 128-column block, one register per array (low half: the even column): its pair of a row
	// is the dword at byte 4 lane + 4 of the block (two pad entries in front), the pairs shifted by one column are that dword and its
	// neighbour through one v_alignbit.
	const int32_t vb = lane * 4 + 4;
	const int32_t ONE = 0x00010001;
	for (;;) {
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
			if (trace_band && s_new - 1 < dbg_cap) M.dbg[2 * (s_new - 1)] = lo, M.dbg[2 * (s_new - 1) + 1] = hi;
		}
		// blocks written: those that meet the window and nH columns either side of it (dead), clamped to the span
		const int32_t b_first = (max(lo - nH, left) - left) >> 7, b_last = (min(hi + nH, right) - left) >> 7;
		const int32_t lo_p = both16(lo - left), hi1_p = both16(hi - left + 1);
		uint32_t flags = 0;
		int32_t fin_info = 0;
		const bool forecast = s_new == 64 || s_new == 256 || s_new == 1024; // uniform: look at how far the pair has come (dev::window_forecast)
		int32_t far = kDead16;
		for (int32_t b = b_first + ((wave - b_first) & (NW - 1)); b <= b_last; b += NW) {
			const int32_t ga = vb + 256 * b, cr = 128 * b + 2 * lane, c = left + cr; // (cr: the even column, relative to the span)
			const int32_t d = c - 1 - tl;
			// sources (reference wf_next_prep, miniwfa.c:252-257): own pair, the pair before, the pair behind
			const char *const pX = base + ga + oX, *const pA = base + ga + oA, *const pB = base + ga + oB;
			const char *const pE1 = base + ga + (bE1 + oR1), *const pF1 = base + ga + (bF1 + oR1), *const pE2 = base + ga + (bE2 + oR2), *const pF2 = base + ga + (bF2 + oR2);
			const int32_t hx = *(const int32_t*)pX;
			const int32_t a0 = *(const int32_t*)(pA - 4), a1 = *(const int32_t*)pA, a2 = *(const int32_t*)(pA + 4);
			const int32_t b0 = *(const int32_t*)(pB - 4), b1 = *(const int32_t*)pB, b2 = *(const int32_t*)(pB + 4);
			const int32_t e10 = *(const int32_t*)(pE1 - 4), e11 = *(const int32_t*)pE1, f11 = *(const int32_t*)pF1, f12 = *(const int32_t*)(pF1 + 4);
			const int32_t e20 = *(const int32_t*)(pE2 - 4), e21 = *(const int32_t*)pE2, f21 = *(const int32_t*)pF2, f22 = *(const int32_t*)(pF2 + 4);
			const int32_t o1m = __builtin_amdgcn_alignbit(a1, a0, 16), o1p = __builtin_amdgcn_alignbit(a2, a1, 16);
			const int32_t o2m = __builtin_amdgcn_alignbit(b1, b0, 16), o2p = __builtin_amdgcn_alignbit(b2, b1, 16);
			const int32_t g1m = __builtin_amdgcn_alignbit(e11, e10, 16), g1p = __builtin_amdgcn_alignbit(f12, f11, 16);
			const int32_t g2m = __builtin_amdgcn_alignbit(e21, e20, 16), g2p = __builtin_amdgcn_alignbit(f22, f21, 16);
			// the recurrence (dev::wf_cell, miniwfa.c:267-278) on the pair
			int32_t ne1 = pk_max(o1m, g1m), ne2 = pk_max(o2m, g2m);
			const int32_t pf1 = pk_max(o1p, g1p), pf2 = pk_max(o2p, g2p); // F before its + 1
			int32_t nf1 = pk_add(pf1, ONE), nf2 = pk_add(pf2, ONE);
			const int32_t m = pk_add(hx, ONE);
			int32_t hh = pk_max(pk_max(m, pk_max(ne1, ne2)), pk_max(nf1, nf2));
			uint32_t tbw = 0;
			if (TB) {
				// the byte from the RESULTS (miniwfa.c:289-306; mwf_band2.hip has the derivation): z = nm (1 + ne1 (2 + ne2 (2 nf1 - 1))) + 8 x_e1 + 16 x_f1 + 32 x_e2 + 64 x_f2
				const int32_t TWO = 0x00020002, NEG1 = (int32_t)0xffffffffu, EIGHT = 0x00080008, C16 = 0x00100010, C32 = 0x00200020, C64 = 0x00400040;
				int32_t z = pk_mad(pk_ne1(hh, nf1), TWO, NEG1);
				z = pk_mad(pk_ne1(hh, ne2), z, TWO);
				z = pk_mad(pk_ne1(hh, ne1), z, ONE);
				z = pk_mad(pk_ne1(hh, m), z, 0);
				z = pk_mad(pk_ne1(ne1, o1m), EIGHT, z);
				z = pk_mad(pk_ne1(pf1, o1p), C16, z);
				z = pk_mad(pk_ne1(ne2, o2m), C32, z);
				z = pk_mad(pk_ne1(pf2, o2p), C64, z);
				tbw = ((uint32_t)z & 0xffu) | (((uint32_t)z >> 16) << 8);
			}
			// columns outside the window are not computed by the reference: dead
			if (!(left + 128 * b >= lo && left + 128 * b + 127 <= hi)) { // uniform
				const int32_t colp = pair_of(cr, cr + 1);
				const int32_t out = pk_nonzero_mask(pk_subsat(lo_p, colp) | pk_subsat(pk_add(colp, ONE), hi1_p));
				hh = bfi(out, kDeadPair, hh), ne1 = bfi(out, kDeadPair, ne1), ne2 = bfi(out, kDeadPair, ne2), nf1 = bfi(out, kDeadPair, nf1), nf2 = bfi(out, kDeadPair, nf2);
			}
			*(int32_t*)(base + ga + (bE1 + oN1)) = ne1, *(int32_t*)(base + ga + (bF1 + oN1)) = nf1;
			*(int32_t*)(base + ga + (bE2 + oN2)) = ne2, *(int32_t*)(base + ga + (bF2 + oN2)) = nf2;
			// match extension (reference wf_extend, miniwfa.c:400-411) of the cells inside the matrix — both columns walked together
			const int32_t h0 = lo16(hh), h1 = hi16(hh);
			const bool inm0 = in_matrix(d, h0, tl, ql), inm1 = in_matrix(d + 1, h1, tl, ql); // (a dead offset is never in the matrix)
			const int32_t j0 = inm0 ? h0 + 1 : 0, i0 = inm0 ? d + j0 : 0, j1 = inm1 ? h1 + 1 : 0, i1 = inm1 ? d + 1 + j1 : 0;
			int32_t n0, n1;
			extend_two<S2>(lt, lq, j0, i0, inm0 ? min(tl - j0, ql - i0) : 0, j1, i1, inm1 ? min(tl - j1, ql - i1) : 0, n0, n1);
			const int32_t hx0 = h0 + n0, hx1 = h1 + n1;
			*(int32_t*)(base + ga + oN) = pair_of(hx0, hx1);
			far = max(far, max(hx0, hx1));
			if (TB) *(uint16_t*)(M.tb + tb_used + cr) = (uint16_t)tbw;
			if (track_good) { // some array holds an in-matrix offset here (good_diag, miniwfa.c:139-142)
				const bool gd0 = inm0 || in_matrix(d, lo16(ne1), tl, ql) || in_matrix(d, lo16(nf1), tl, ql) || in_matrix(d, lo16(ne2), tl, ql) || in_matrix(d, lo16(nf2), tl, ql);
				const bool gd1 = inm1 || in_matrix(d + 1, hi16(ne1), tl, ql) || in_matrix(d + 1, hi16(nf1), tl, ql) || in_matrix(d + 1, hi16(ne2), tl, ql) || in_matrix(d + 1, hi16(nf2), tl, ql);
				const unsigned long long m0 = __ballot(gd0), m1 = __ballot(gd1);
				if (lane == 0) good[newH * NG + 2 * b] = m0, good[newH * NG + 2 * b + 1] = m1;
			}
			// edge rule (miniwfa.c:325-326): H is the max of the five, so "any live" == "H live"; termination (miniwfa.c:405-409)
			const uint32_t live0 = (uint32_t)(hx0 >= -1), live1 = (uint32_t)(hx1 >= -1);
			const bool fin0 = c == cfin && c >= lo && c <= hi && hx0 == tl - 1 && in_matrix(ql - tl, h0, tl, ql);
			const bool fin1 = c + 1 == cfin && c + 1 >= lo && c + 1 <= hi && hx1 == tl - 1 && in_matrix(ql - tl, h1, tl, ql);
			flags |= (live0 & (uint32_t)(c == lo)) | (live1 & (uint32_t)(c + 1 == lo)) | ((live0 & (uint32_t)(c == hi)) << 1) | ((live1 & (uint32_t)(c + 1 == hi)) << 1) | ((uint32_t)(fin0 || fin1) << 2);
			fin_info = fin0 ? (n0 == 0 ? (int32_t)(tbw & 7u) : 0) : fin1 ? (n1 == 0 ? (int32_t)((tbw >> 8) & 7u) : 0) : fin_info;
		}
		if (__ballot(flags != 0)) { // this wave's share of the three per-penalty flags: one LDS atomic per wave that has any
			const unsigned long long fm = __ballot(flags & 4u);
			uint32_t bits = (__ballot(flags & 1u) ? 1u : 0u) | (__ballot(flags & 2u) ? 2u : 0u);
			if (fm) bits |= 4u | (uint32_t)__builtin_amdgcn_readlane(fin_info, (int32_t)__builtin_ctzll(fm)) << 4;
			if (lane == 0) atomicOr((unsigned int*)&V.flags[npar], bits);
		}
		if (forecast) {
			const int32_t mx = wave_max(far);
			if (lane == 0 && mx >= 0) atomicMax(&V.far, mx);
		}
		// the rows of the coming penalty
		oN = oN + RB == HB ? 0 : oN + RB, oX = oX + RB == HB ? 0 : oX + RB, oA = oA + RB == HB ? 0 : oA + RB, oB = oB + RB == HB ? 0 : oB + RB;
		oN1 = oN1 + RB == B1 ? 0 : oN1 + RB, oR1 = oR1 + RB == B1 ? 0 : oR1 + RB, oN2 = oN2 + RB == B2 ? 0 : oN2 + RB, oR2 = oR2 + RB == B2 ? 0 : oR2 + RB;
		__syncthreads();
		const uint32_t fl = (uint32_t)uni(V.flags[npar]);
		if (fl & 1u) wf_lo = lo;
		if (fl & 2u) wf_hi = hi;
		s = s_new, curH = newH, par = npar;
		if (TB) tb_used += C;
		if ((s & 0xff) == 0) { // shrink (reference wf_stripe_shrink, miniwfa.c:144-171) on the good bits of the slices still in the ring
			if (tid == 0) V.red[0] = 0x7fffffff, V.red[1] = -1;
			__syncthreads();
			const int32_t bA = (wf_lo - left) >> 7, bB = (wf_hi - left) >> 7;
			for (int32_t q = 2 * bA + tid; q <= 2 * bB + 1; q += T) { // one word each: block q >> 1, its even (q & 1 == 0) or odd columns
				const int32_t w0 = left + 128 * (q >> 1), k = q & 1;
				unsigned long long m = 0;
				for (int32_t j = 0; j < nH; ++j) {
					const int2 w = win[j];
					m |= good[j * NG + q] & block_mask(w0, k, w.x, w.y);
				}
				m &= block_mask(w0, k, wf_lo, wf_hi);
				if (m) {
					atomicMin(&V.red[0], w0 + k + 2 * (int32_t)__builtin_ctzll(m));
					atomicMax(&V.red[1], w0 + k + 2 * (63 - (int32_t)__builtin_clzll(m)));
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

template <int T, bool TB, bool S2>
__global__ __launch_bounds__(T, 1) void wfa_mid_kernel(const BatchArgs)
{
	// the arguments are read from the kernarg segment where they are used (mwf_device.h): nothing of them stays in SGPRs across the penalties
	KArgs &A = kernel_args();
	const int32_t tid = threadIdx.x;
	const MidLayout L = mid_layout(A.pen.nH, A.pen.e1, A.pen.e2, A.lane_chunks * 64);
	MidVars &V = *(MidVars*)(lds_mid + L.vars_off);
	uint8_t *lt = lds_mid + L.seq_off;
	for (int32_t round = 0;; ++round) {
		// a work counter, or — queue == null: a launch of one workgroup per pair — pair blockIdx.x and nothing else (no counter to zero first)
		if (tid == 0) V.item = A.queue ? (int32_t)atomicAdd(A.queue, 1) : (round == 0 ? (int32_t)blockIdx.x : A.n_pairs), V.word = 0;
		__syncthreads();
		const int32_t item = uni(V.item);
		__syncthreads();
		if (item >= A.n_pairs) break;
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
		if (R.status == ST_OK) R = mid_pass<T, TB, S2>(fresh(A), M, L, lt, lq, trace);
		if (S2) M.t2 = lt, M.q2 = lq; // the traceback's back-match stays on chip
		finish_pair(fresh(A), M, (int32_t)blockIdx.x, pair, R, R.status, 0);
	}
}

template <int T, bool TB, bool S2>
int launch_v(const BatchArgs &a, int grid, int lds, hipStream_t st)
{
	// beyond 48 KB of dynamic LDS the runtime wants to be told (per device, and this may run on several host threads: on every launch)
	if (lds > 48 * 1024) {
		(void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wfa_mid_kernel<T, TB, S2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
		(void)hipGetLastError();
	}
	hipLaunchKernelGGL((wfa_mid_kernel<T, TB, S2>), dim3(grid), dim3(T), lds, st, a);
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
