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
//   * both sequences sit in LDS as bytes (any alphabet); the extension compares 8 bytes per lane and trip, the wave walks together;
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
};

// eight bytes at an arbitrary byte offset of an LDS array (three aligned dwords, two v_alignbyte)
__device__ __forceinline__ uint64_t lds_ld8(const uint8_t *base, int32_t off)
{
	const uint32_t *p = (const uint32_t*)(base + (off & ~3));
	const uint32_t a = p[0], b = p[1], c = p[2];
	const uint32_t lo = __builtin_amdgcn_alignbyte(b, a, (uint32_t)off), hi = __builtin_amdgcn_alignbyte(c, b, (uint32_t)off);
	return (uint64_t)hi << 32 | lo;
}

// Length of the exact-match run t[j..] == q[i..], at most `room` (<= 0: none; j and i must then still be readable offsets).  The wave
// walks together, eight bytes per lane and trip, while any lane's run is open (straight-line trips under one uniform branch).
__device__ __forceinline__ int32_t mid_extend(const uint8_t *lt, const uint8_t *lq, int32_t j, int32_t i, int32_t room)
{
	int32_t n = 0;
	bool open = room > 0;
	while (__ballot(open)) {
		const uint64_t x = lds_ld8(lt, j + n) ^ lds_ld8(lq, i + n);
		const int32_t adv = x ? (int32_t)(__builtin_ctzll(x) >> 3) : 8;
		n += open ? adv : 0;
		open = open && x == 0 && n < room;
	}
	return max(min(n, room), 0);
}

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

template <int T, bool TB, typename ArgsT>
__device__ PassResult mid_pass(const ArgsT &A, PairMem &M, const MidLayout &L, const uint8_t *lt, const uint8_t *lq, bool trace_band)
{
	constexpr int NW = T / 64;
	const int32_t tl = M.tl, ql = M.ql, cmax = tl + ql + 1;
	const int32_t tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6);
	const int32_t nH = A.pen.nH, lagx = A.pen.x, lag1 = A.pen.oe1, lag2 = A.pen.oe2, e1 = A.pen.e1, e2 = A.pen.e2, n1 = e1 + 1, n2 = e2 + 1;
	const int32_t max_s = A.max_s, dbg_cap = A.dbg_cap;
	const int64_t max_iter = A.max_iter;
	const int64_t tb_slot_bytes = A.tb_slot_bytes;
	const int32_t C = A.lane_chunks * 64, RL = L.RL, NG = C / 64;
	// the span: C columns around the middle of the diagonals the alignment path runs between (0 and ql - tl); entry 1 of a row is column `left`
	const int32_t center = tl + 1 + (ql - tl) / 2, left = center - C / 2, right = left + C - 1;
	int16_t *const Hr = (int16_t*)lds_mid, *const E1r = Hr + nH * RL, *const F1r = E1r + n1 * RL, *const E2r = F1r + n1 * RL, *const F2r = E2r + n2 * RL;
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
	}
	__syncthreads();
	const int32_t c00 = tl + 1;
	if (c00 < left || c00 > right) { R.status = ST_BAND_OVERFLOW; return R; } // (cannot happen: the span is centred between the two end diagonals and at least 64 wide ... unless |ql - tl| > C)
	int32_t k0 = 0;
	if (wave == 0) {
		k0 = mid_extend(lt, lq, 0, 0, min(tl, ql)) - 1;
		if (lane == 0) Hr[c00 - left + 1] = (int16_t)k0, win[0] = make_int2(c00, c00), V.word = k0;
	}
	__syncthreads();
	k0 = uni(V.word);
	if (k0 == tl - 1 && k0 == ql - 1) return R;

	int32_t s = 0, wf_lo = c00, wf_hi = c00;
	int32_t curH = 0, cur1 = 0, cur2 = 0, par = 0;
	int64_t cells = 0, tb_used = 0;
	if (TB) M.tb_stride = C, M.tb_left = left;
	const int32_t cfin = ql + 1; // the end cell (tl-1, ql-1) lies on diagonal ql-tl, i.e. this column
	for (;;) {
		const int32_t lo = wf_lo > 1 ? wf_lo - 1 : 1;       // miniwfa.c:417-418
		const int32_t hi = wf_hi < cmax ? wf_hi + 1 : cmax;
		const int32_t s_new = s + 1;
		if (lo < left || hi > right) { R.status = ST_BAND_OVERFLOW; break; }
		if (s_new + tl >= 32760) { R.status = ST_BAND_OVERFLOW; break; } // an offset (a target index, or past the matrix by one per penalty) must fit 16 bits
		if (TB && tb_used + C > tb_slot_bytes) { R.status = ST_TB_OVERFLOW; break; }
		const int32_t newH = curH + 1 == nH ? 0 : curH + 1;
		const int32_t new1 = cur1 + 1 == n1 ? 0 : cur1 + 1, new2 = cur2 + 1 == n2 ? 0 : cur2 + 1;
		const int32_t npar = par + 1 == 3 ? 0 : par + 1;
		int32_t jx = newH - lagx; if (jx < 0) jx += nH;
		int32_t j1 = newH - lag1; if (j1 < 0) j1 += nH;
		int32_t j2 = newH - lag2; if (j2 < 0) j2 += nH;
		const int32_t r1 = new1 + 1 == n1 ? 0 : new1 + 1, r2 = new2 + 1 == n2 ? 0 : new2 + 1; // rows of penalties s_new - e1, s_new - e2
		const bool track_good = (((256 - (s_new & 255)) & 255) < nH); // a shrink can still see this slice
		if (tid == 0) {
			win[newH] = make_int2(lo, hi);
			V.flags[npar + 1 == 3 ? 0 : npar + 1] = 0; // the flag word of the NEXT penalty (its last readers passed the previous barrier)
			if (trace_band && s_new - 1 < dbg_cap) M.dbg[2 * (s_new - 1)] = lo, M.dbg[2 * (s_new - 1) + 1] = hi;
		}
		// columns written: the window and nH either side (dead), clamped to the span
		const int32_t wlo = max(lo - nH, left), whi = min(hi + nH, right);
		const int32_t g_first = (wlo - left) >> 6, g_last = (whi - left) >> 6;
		const int16_t *const hx_base = Hr + jx * RL, *const o1_base = Hr + j1 * RL, *const o2_base = Hr + j2 * RL;
		const int16_t *const e1_base = E1r + r1 * RL, *const f1_base = F1r + r1 * RL, *const e2_base = E2r + r2 * RL, *const f2_base = F2r + r2 * RL;
		uint32_t flags = 0;
		int32_t fin_info = 0;
		for (int32_t g = g_first + (wave - g_first % NW + NW) % NW; g <= g_last; g += NW) {
			const int32_t idx = 64 * g + lane + 1, c = left + 64 * g + lane;
			const int32_t d = c - 1 - tl;
			// sources (reference wf_next_prep, miniwfa.c:252-257)
			const int32_t hx = hx_base[idx], o1m = o1_base[idx - 1], o1p = o1_base[idx + 1], o2m = o2_base[idx - 1], o2p = o2_base[idx + 1];
			const int32_t g1m = e1_base[idx - 1], g1p = f1_base[idx + 1], g2m = e2_base[idx - 1], g2p = f2_base[idx + 1];
			const bool act = c >= lo && c <= hi;
			const Cell v = wf_cell<TB>(hx, o1m, g1m, o2m, g2m, o1p, g1p, o2p, g2p);
			E1r[new1 * RL + idx] = (int16_t)(act ? max(v.e1, kDead16) : kDead16), F1r[new1 * RL + idx] = (int16_t)(act ? max(v.f1, kDead16) : kDead16);
			E2r[new2 * RL + idx] = (int16_t)(act ? max(v.e2, kDead16) : kDead16), F2r[new2 * RL + idx] = (int16_t)(act ? max(v.f2, kDead16) : kDead16);
			// match extension (reference wf_extend, miniwfa.c:400-411) of the cells inside the matrix
			const bool inm = act && in_matrix(d, v.h, tl, ql);
			const int32_t j = inm ? v.h + 1 : 0, i = inm ? d + j : 0;
			const int32_t nmat = mid_extend(lt, lq, j, i, inm ? min(tl - j, ql - i) : 0);
			const int32_t h = act ? max(v.h + nmat, kDead16) : kDead16;
			Hr[newH * RL + idx] = (int16_t)h;
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
		{ // this wave's share of the three per-penalty flags: one LDS atomic per wave that has any
			const unsigned long long fm = __ballot(flags & 4u);
			uint32_t bits = (__ballot(flags & 1u) ? 1u : 0u) | (__ballot(flags & 2u) ? 2u : 0u);
			if (fm) bits |= 4u | (uint32_t)__builtin_amdgcn_readlane(fin_info, (int32_t)__builtin_ctzll(fm)) << 4;
			if (bits && lane == 0) atomicOr((unsigned int*)&V.flags[npar], bits);
		}
		__syncthreads();
		const uint32_t fl = (uint32_t)uni(V.flags[npar]);
		if (fl & 1u) wf_lo = lo;
		if (fl & 2u) wf_hi = hi;
		s = s_new, curH = newH, cur1 = new1, cur2 = new2, par = npar;
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
		if ((max_iter > 0 && cells > max_iter) || (max_s > 0 && s > max_s)) { // miniwfa.c:422-425
			R.status = ST_STOPPED;
			break;
		}
		if (fl & 4u) { R.info = (int32_t)((fl >> 4) & 7u); break; }
	}
	R.s = s, R.cells = cells;
	return R;
}

template <int T, bool TB>
__global__ __launch_bounds__(T, 1) void wfa_mid_kernel(const BatchArgs)
{
	// the arguments are read from the kernarg segment where they are used (mwf_device.h): nothing of them stays in SGPRs across the penalties
	KArgs &A = kernel_args();
	const int32_t tid = threadIdx.x;
	const MidLayout L = mid_layout(A.pen.nH, A.pen.e1, A.pen.e2, A.lane_chunks * 64);
	MidVars &V = *(MidVars*)(lds_mid + L.vars_off);
	uint8_t *lt = lds_mid + L.seq_off;
	for (;;) {
		if (tid == 0) V.item = (int32_t)atomicAdd(A.queue, 1);
		__syncthreads();
		const int32_t item = uni(V.item);
		__syncthreads();
		if (item >= A.n_pairs) break;
		const int32_t pair = A.order ? A.order[item] : item;
		PairMem M;
		pair_mem(fresh(A), (int32_t)blockIdx.x, pair, M);
		M.tl = uni(M.tl), M.ql = uni(M.ql);
		uint8_t *lq = lt + ((M.tl + 7) & ~7) + 16;
		// both sequences into LDS, eight bytes per thread and trip (the packed sequence buffer has 64 bytes of slack behind it)
		for (int32_t j = 8 * tid; j < M.tl; j += 8 * T) *(uint64_t*)(lt + j) = ld8(M.ts + j);
		for (int32_t j = 8 * tid; j < M.ql; j += 8 * T) *(uint64_t*)(lq + j) = ld8(M.qs + j);
		__syncthreads();
		const bool trace = A.dbg && pair == A.debug_pair;
		const PassResult R = mid_pass<T, TB>(fresh(A), M, L, lt, lq, trace);
		finish_pair(fresh(A), M, (int32_t)blockIdx.x, pair, R, R.status, 0);
	}
}

template <int T>
int launch_t(const BatchArgs &a, int grid, int lds, hipStream_t st)
{
	// beyond 48 KB of dynamic LDS the runtime wants to be told (per device, and this may run on several host threads: on every launch)
	if (lds > 48 * 1024) {
		(void)hipFuncSetAttribute(a.want_cigar ? reinterpret_cast<const void*>(&wfa_mid_kernel<T, true>) : reinterpret_cast<const void*>(&wfa_mid_kernel<T, false>),
		                          hipFuncAttributeMaxDynamicSharedMemorySize, lds);
		(void)hipGetLastError();
	}
	if (a.want_cigar) hipLaunchKernelGGL((wfa_mid_kernel<T, true>), dim3(grid), dim3(T), lds, st, a);
	else hipLaunchKernelGGL((wfa_mid_kernel<T, false>), dim3(grid), dim3(T), lds, st, a);
	return hipGetLastError() == hipSuccess ? 0 : -2;
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

int launch_mid(const BatchArgs &a, int grid, int block, int lds, void *stream)
{
	if (block == 256) return launch_t<256>(a, grid, lds, (hipStream_t)stream);
	if (block == 1024) return launch_t<1024>(a, grid, lds, (hipStream_t)stream);
	return -1;
}

} // namespace mwf
