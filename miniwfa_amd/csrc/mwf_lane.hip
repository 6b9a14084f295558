// mwf_lane.hip — one wave per pair, one diagonal per lane, everything on chip: the kernel for SHORT pairs (single calls of
// mwf_wfa_exact on a read-sized pair, batches of short reads).
//
// A 200 bp pair at 5 % divergence ends near penalty 36 with a window of ~45 diagonals.  The packed band kernel (mwf_band2.hip)
// computes a 256-column chunk for it per penalty — ~800 instructions of a lone wave, 1.7 us — and keeps its H rows in HBM.  Here a
// lane owns ONE column per 64-column chunk (column = diagonal + tl + 1, as everywhere), so a penalty is the recurrence on one cell
// plus its match extension, once per chunk the window has reached:
//   * chunk 0 holds the 64 columns around the main diagonal; chunk k >= 1 the 32 columns beyond chunk k-1 on either side (lanes
//     0-31 left, 32-63 right) — a window is symmetric around the main diagonal until the matrix clips it, so a narrow window costs
//     one pass over the recurrence and a window of up to 64 x `lane_chunks` columns (three chunks for pairs of up to 400 bases of
//     target + query, else four: penalties up to ~110 / ~140 with the default gap costs) is held;
//   * the H ring (nH rows), the E1/F1 rings (e1 rows each) and the E2/F2 rings (e2 rows) are rows of int16 in LDS — offsets
//     of pairs this short fit, a dead cell is stored as max(v, -32768) exactly as in the packed band kernel — with one pad column
//     either side that always reads dead (reference pads, miniwfa.c:103-121); a lane reads its neighbours' columns straight from
//     the rows, so any penalties are served (no template on e1/e2) and no cross-lane shuffles are needed;
//   * every column inside a chunk that the window has reached is written every penalty — the offset, or dead outside [lo, hi] — so
//     rows read as dead beyond their window without window tests (windows only grow here: the kernel hands a pair back before the
//     first band shrink);
//   * a wave executes its LDS instructions in order and no other wave shares the rows: no barrier anywhere, one wave per workgroup;
//   * both sequences sit in LDS (8-byte copies), the extension compares 8 bytes per trip;
//   * the traceback bytes go to the slot's arena as rows of 64 x chunks bytes that all start at the leftmost column: the shared
//     traceback (mwf_device.h) finds a byte without reading a row table first — one memory round trip per step instead of two.
// A pair whose window leaves the chunks, or that reaches the first shrink (penalty 256 - nH), comes back as ST_BAND_OVERFLOW and is
// re-run on the packed band kernel (finalize(), mwf_engine.cpp).  Reference: wf_next_basic + wf_extend + the loop of mwf_wfa_core
// (miniwfa.c:252-326, :380-430).
#include "mwf_device.h"

namespace mwf {

using namespace dev;

namespace {

extern __shared__ __attribute__((aligned(16))) uint8_t lds_lane[];

constexpr int32_t kDead16 = -32768;

__device__ __forceinline__ int32_t row_ints(int nc) { return 32 * nc + 2; } // a row: pad, 64 x nc columns, pad as int16, rounded up to dwords

// eight bytes at an arbitrary byte offset of an LDS array (three aligned dwords, two v_alignbyte)
__device__ __forceinline__ uint64_t lds_ld8(const uint8_t *base, int32_t off)
{
	const uint32_t *p = (const uint32_t*)(base + (off & ~3));
	const uint32_t a = p[0], b = p[1], c = p[2];
	const uint32_t lo = __builtin_amdgcn_alignbyte(b, a, (uint32_t)off), hi = __builtin_amdgcn_alignbyte(c, b, (uint32_t)off);
	return (uint64_t)hi << 32 | lo;
}

// Length of the exact-match run t[j..] == q[i..], at most `room` (<= 0: none; j and i must then still be readable offsets).  The
// wave walks together, eight bytes per lane and trip, while any lane's run is open: straight-line trips under one uniform branch
// (a divergent while loop costs ~25 mask instructions per trip).
__device__ __forceinline__ int32_t lane_extend(const uint8_t *lt, const uint8_t *lq, int32_t j, int32_t i, int32_t room)
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

template <bool TB, typename ArgsT>
__device__ PassResult lane_pass(const ArgsT &A, PairMem &M, int16_t *rows, const uint8_t *lt, const uint8_t *lq, bool trace_band)
{
	const int32_t tl = M.tl, ql = M.ql, cmax = tl + ql + 1;
	const int32_t lane = threadIdx.x;
	const int32_t nH = A.pen.nH, lagx = A.pen.x, lag1 = A.pen.oe1, lag2 = A.pen.oe2, e1 = A.pen.e1, e2 = A.pen.e2, n1 = e1, n2 = e2;
	const int32_t max_s = A.max_s, dbg_cap = A.dbg_cap;
	const int64_t max_iter = A.max_iter;
	const int32_t tb_slot_bytes = (int32_t)min(A.tb_slot_bytes, (int64_t)0x7fffffff);
	const int32_t NC = A.lane_chunks, RL = row_ints(NC) * 2; // RL: int16 entries per row
	const int32_t center = tl + 1, left = center - 32 * NC;   // entry 1 of a row is column `left`
	int16_t *const Hr = rows, *const E1r = Hr + nH * RL, *const F1r = E1r + n1 * RL, *const E2r = F1r + n1 * RL, *const F2r = E2r + n2 * RL;
	const int32_t n_rows = nH + 2 * n1 + 2 * n2;
	PassResult R;
	R.status = ST_OK, R.s = 0, R.info = 0, R.n_snap = 0, R.cells = 0;

	// ---- penalty 0 (reference wf_stripe_init, miniwfa.c:103-121) and its extension
	{
		const uint32_t dd = 0x80008000u;
		const uint4 dead4 = make_uint4(dd, dd, dd, dd);
		for (int32_t j = lane; j < (n_rows * RL * 2 + 15) / 16; j += 64) ((uint4*)rows)[j] = dead4;
	}
	__syncthreads(); // (one wave: for the compiler, the int16 accesses below are not reordered with the wide stores)
	const int32_t k0 = lane_extend(lt, lq, 0, 0, min(tl, ql)) - 1;
	if (lane == 0) Hr[center - left + 1] = (int16_t)k0;
	if (k0 == tl - 1 && k0 == ql - 1) return R;

	int32_t s = 0, wf_lo = tl + 1, wf_hi = tl + 1;
	int32_t curH = 0, a1 = 0, a2 = 0; // H slot of penalty s; slots of the E1/F1 and E2/F2 rings the next penalty reads (e1, e2 penalties old) and then overwrites
	int64_t cells = 0;
	int32_t tb_used = 0;
	if (TB) M.tb_stride = 64 * NC, M.tb_left = left;
	const int32_t s_shrink = 256 - nH; // the first penalty whose good bits a shrink would read (wf_stripe_shrink, miniwfa.c:144-171)
	const int32_t cfin = ql + 1;       // the end cell (tl-1, ql-1) lies on diagonal ql-tl, i.e. this column
	for (;;) {
		const int32_t lo = wf_lo > 1 ? wf_lo - 1 : 1;       // miniwfa.c:417-418
		const int32_t hi = wf_hi < cmax ? wf_hi + 1 : cmax;
		const int32_t s_new = s + 1;
		// chunks the window has reached: column c < center lies in chunk (center-1-c)/32, c >= center in chunk (c-center)/32
		const int32_t k_use = max(lo < center ? (center - 1 - lo) >> 5 : 0, hi > center ? (hi - center) >> 5 : 0);
		if (k_use >= NC || s_new >= s_shrink) { R.status = ST_BAND_OVERFLOW; break; }
		const int32_t newH = curH + 1 == nH ? 0 : curH + 1;
		const int32_t b1 = a1, b2 = a2, r1 = a1, r2 = a2; // a ring of exactly e1 (e2) rows: the row read is the row overwritten
		if (TB && tb_used + 64 * NC > tb_slot_bytes) { R.status = ST_TB_OVERFLOW; break; }
		if (trace_band && lane == 0 && s_new - 1 < dbg_cap) M.dbg[2 * (s_new - 1)] = lo, M.dbg[2 * (s_new - 1) + 1] = hi;
		int32_t jx = newH - lagx; if (jx < 0) jx += nH;
		int32_t j1 = newH - lag1; if (j1 < 0) j1 += nH;
		int32_t j2 = newH - lag2; if (j2 < 0) j2 += nH;
		uint32_t flags = 0;    // per lane, over its chunks: 1 = the lo column and live, 2 = the hi column and live, 4 = the end cell, reached
		int32_t fin_info = 0;
		// The E/F row a chunk overwrites is the row the next chunk still reads at the two columns where their blocks touch: the old F of
		// this chunk's first column (lane 0) and the old E of its last (lane 63) travel to the next chunk in scalars.
		int32_t cE1 = 0, cE2 = 0, cF1 = 0, cF2 = 0;
		for (int32_t k = 0; k <= k_use; ++k) {
			const int32_t c = lane < 32 ? center - 32 * (k + 1) + lane : center + 32 * k + lane - 32;
			const int32_t d = c - center, idx = c - left + 1;
			// sources (reference wf_next_prep, miniwfa.c:252-257)
			const int16_t *hx_row = Hr + jx * RL + idx, *o1_row = Hr + j1 * RL + idx, *o2_row = Hr + j2 * RL + idx;
			const int32_t hx = hx_row[0], o1m = o1_row[-1], o1p = o1_row[1], o2m = o2_row[-1], o2p = o2_row[1];
			int32_t g1m = E1r[r1 * RL + idx - 1], g1p = F1r[r1 * RL + idx + 1], g2m = E2r[r2 * RL + idx - 1], g2p = F2r[r2 * RL + idx + 1];
			if (k > 0) { // lane 31: the column left of the previous chunk's first; lane 32: the column right of its last
				g1p = lane == 31 ? cF1 : g1p, g2p = lane == 31 ? cF2 : g2p;
				g1m = lane == 32 ? cE1 : g1m, g2m = lane == 32 ? cE2 : g2m;
			}
			if (k < k_use) {
				const int32_t oE1 = E1r[r1 * RL + idx], oF1 = F1r[r1 * RL + idx], oE2 = E2r[r2 * RL + idx], oF2 = F2r[r2 * RL + idx];
				cE1 = __builtin_amdgcn_readlane(oE1, 63), cE2 = __builtin_amdgcn_readlane(oE2, 63);
				cF1 = __builtin_amdgcn_readlane(oF1, 0), cF2 = __builtin_amdgcn_readlane(oF2, 0);
			}
			const bool act = c >= lo && c <= hi;
			const Cell v = wf_cell<TB>(hx, o1m, g1m, o2m, g2m, o1p, g1p, o2p, g2p);
			E1r[b1 * RL + idx] = (int16_t)(act ? max(v.e1, kDead16) : kDead16), F1r[b1 * RL + idx] = (int16_t)(act ? max(v.f1, kDead16) : kDead16);
			E2r[b2 * RL + idx] = (int16_t)(act ? max(v.e2, kDead16) : kDead16), F2r[b2 * RL + idx] = (int16_t)(act ? max(v.f2, kDead16) : kDead16);
			// match extension (reference wf_extend, miniwfa.c:208-246) of the cells inside the matrix
			const bool inm = act && in_matrix(d, v.h, tl, ql);
			const int32_t j = inm ? v.h + 1 : 0, i = inm ? d + j : 0;
			const int32_t nmat = lane_extend(lt, lq, j, i, inm ? min(tl - j, ql - i) : 0);
			const int32_t h = act ? max(v.h + nmat, kDead16) : kDead16;
			Hr[newH * RL + idx] = (int16_t)h;
			if (TB && act) M.tb[tb_used + (c - left)] = (uint8_t)v.tb;
			// edge rule (miniwfa.c:325-326): H is the max of the five, so "any live" == "H live"
			const uint32_t live = (uint32_t)(h >= -1);
			// termination (miniwfa.c:405-409)
			const bool fin = act && c == cfin && h == tl - 1 && in_matrix(ql - tl, h - nmat, tl, ql);
			flags |= (live & (uint32_t)(c == lo)) | ((live & (uint32_t)(c == hi)) << 1) | ((uint32_t)fin << 2);
			fin_info = fin ? (nmat == 0 ? (int32_t)(v.tb & 7u) : 0) : fin_info;
		}
		if (__ballot(flags & 1u)) wf_lo = lo;
		if (__ballot(flags & 2u)) wf_hi = hi;
		const unsigned long long fm = __ballot(flags & 4u);
		s = s_new, curH = newH;
		a1 = a1 + 1 == n1 ? 0 : a1 + 1, a2 = a2 + 1 == n2 ? 0 : a2 + 1;
		if (TB) tb_used += 64 * NC;
		cells += hi - lo + 1;
		if ((max_iter > 0 && cells > max_iter) || (max_s > 0 && s > max_s)) { // miniwfa.c:422-425
			R.status = ST_STOPPED;
			break;
		}
		if (fm) { R.info = __builtin_amdgcn_readlane(fin_info, (int32_t)__builtin_ctzll(fm)); break; }
	}
	R.s = s, R.cells = cells;
	return R;
}

template <bool TB>
__global__ __launch_bounds__(64) void wfa_lane_kernel(const BatchArgs)
{
	// the arguments are read from the kernarg segment where they are used (mwf_device.h): nothing of them stays in SGPRs across the penalties
	KArgs &A = kernel_args();
	const int32_t lane = threadIdx.x;
	const int32_t n_rows = A.pen.nH + 2 * A.pen.e1 + 2 * A.pen.e2;
	int16_t *rows = (int16_t*)lds_lane;
	uint8_t *lt = lds_lane + (n_rows * row_ints(A.lane_chunks) * 4 + 15) / 16 * 16;
	for (;;) {
		int32_t item = 0;
		if (lane == 0) item = (int32_t)atomicAdd(A.queue, 1);
		item = uni(item);
		if (item >= A.n_pairs) break;
		const int32_t pair = A.order ? A.order[item] : item;
		PairMem M;
		pair_mem(fresh(A), (int32_t)blockIdx.x, pair, M);
		M.tl = uni(M.tl), M.ql = uni(M.ql);
		uint8_t *lq = lt + ((M.tl + 7) & ~7) + 16;
		// both sequences into LDS, eight bytes per lane and trip (the packed sequence buffer has 64 bytes of slack behind it)
		for (int32_t j = 8 * lane; j < M.tl; j += 512) *(uint64_t*)(lt + j) = ld8(M.ts + j);
		for (int32_t j = 8 * lane; j < M.ql; j += 512) *(uint64_t*)(lq + j) = ld8(M.qs + j);
		__syncthreads(); // (one wave: orders the copies before the dword reads of the extension for the compiler)
		const bool trace = A.dbg && pair == A.debug_pair;
		const PassResult R = lane_pass<TB>(fresh(A), M, rows, lt, lq, trace);
		finish_pair(fresh(A), M, (int32_t)blockIdx.x, pair, R, R.status, 0);
	}
}

} // namespace

bool lane_supported(const Penalty &p)
{
	return p.x >= 1 && p.e1 >= 1 && p.e2 >= 1 && p.nH + 2 * p.e1 + 2 * p.e2 <= 96 && p.nH < 128;
}

// dynamic LDS of a launch: the rings plus the sequence copy, where seq_bytes >= tl + ql + 24 for every pair of the launch (the copy
// needs (tl rounded up to 8) + 16 + (ql rounded up to 8) + 32 bytes)
int lane_lds_bytes(const Penalty &p, int chunks, int64_t seq_bytes)
{
	const int64_t rings = ((int64_t)(p.nH + 2 * p.e1 + 2 * p.e2) * (32 * chunks + 2) * 4 + 15) / 16 * 16;
	return (int)((rings + seq_bytes + 64 + 15) / 16 * 16);
}

int launch_lane(const BatchArgs &a, int grid, int lds, void *stream)
{
	// deep rings (large gap-open costs) or a raised lane_max_len: beyond 48 KB of dynamic LDS the runtime wants to be told (the attribute is
	// per device and this may run on several host threads: set on every launch that needs it, as the band kernels do)
	if (lds > 48 * 1024) {
		(void)hipFuncSetAttribute(a.want_cigar ? reinterpret_cast<const void*>(&wfa_lane_kernel<true>) : reinterpret_cast<const void*>(&wfa_lane_kernel<false>),
		                          hipFuncAttributeMaxDynamicSharedMemorySize, lds);
		(void)hipGetLastError();
	}
	if (a.want_cigar) hipLaunchKernelGGL(wfa_lane_kernel<true>, dim3(grid), dim3(64), lds, (hipStream_t)stream, a);
	else hipLaunchKernelGGL(wfa_lane_kernel<false>, dim3(grid), dim3(64), lds, (hipStream_t)stream, a);
	return hipGetLastError() == hipSuccess ? 0 : -2;
}

int lane_kernel_occupancy(int lds, bool cigar)
{
	int n = 0;
	const hipError_t e = cigar ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, wfa_lane_kernel<true>, 64, lds)
	                           : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, wfa_lane_kernel<false>, 64, lds);
	return e == hipSuccess ? n : 0;
}

} // namespace mwf
