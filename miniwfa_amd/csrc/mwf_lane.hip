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
//   * both sequences sit in LDS — at 2 bits per base for pairs of plain A/C/G/T (sixteen bases per trip of the extension, two LDS
//     instructions), else as bytes (eight per trip; a pair outside A/C/G/T comes back as ST_ALPHABET and is re-run byte-wise);
//   * the traceback bytes go to the slot's arena as rows of 64 x chunks bytes that all start at the leftmost column: the shared
//     traceback (mwf_device.h) finds a byte without reading a row table first — one memory round trip per step instead of two.
// A pair whose window leaves the chunks, or that reaches the first shrink (penalty 256 - nH), comes back as ST_BAND_OVERFLOW and is
// re-run on the packed band kernel (finalize(), mwf_plan.cpp).  Reference: wf_next_basic + wf_extend + the loop of mwf_wfa_core
// (miniwfa.c:252-326, :380-430).
#include "mwf_device.h"

namespace mwf {

using namespace dev;

namespace {

extern __shared__ __attribute__((aligned(16))) uint8_t lds_lane[];

constexpr int32_t kDead16 = -32768;

__device__ __forceinline__ int32_t row_ints(int nc) { return 32 * nc + 2; } // a row: pad, 64 x nc columns, pad as int16, rounded up to dwords

// FOLD (score-only, o1 == x as in the default penalties; the packed band kernel's form, mwf_band2.hip): the E1 / F1 rows hold
// max(E1, H[s-x]) / max(F1, H[s-x]) — what the reference computes e1 penalties later from the row of lag o1+e1 (miniwfa.c:267-278) — and that
// row is not read: two LDS reads less per chunk.  Windows never shrink here (the kernel hands a pair back before the first shrink), so a
// column that held a live H was computed at every later penalty.
template <bool TB, bool S2, bool FOLD, typename ArgsT>
__device__ PassResult lane_pass(const ArgsT &A, PairMem &M, int16_t *rows, const uint8_t *lt, const uint8_t *lq, bool trace_band)
{
	const int32_t tl = M.tl, ql = M.ql, cmax = tl + ql + 1;
	const int32_t lane = threadIdx.x;
	const int32_t nH = A.pen.nH, n1 = A.pen.e1, n2 = A.pen.e2;
	const int32_t dbg_cap = A.dbg_cap;
	const int64_t iter_limit = A.max_iter > 0 ? A.max_iter : INT64_MAX;
	const int32_t s_limit = A.max_s > 0 ? A.max_s : INT32_MAX;
	const int32_t tb_slot_bytes = (int32_t)min(A.tb_slot_bytes, (int64_t)0x7fffffff);
	const int32_t NC = A.lane_chunks, RL = row_ints(NC) * 2; // RL: int16 entries per row
	const int32_t center = tl + 1, left = center - 32 * NC;   // entry 1 of a row is column `left`
	// rows as byte offsets: ring bases, ring sizes, and the rows of the coming penalty carried from penalty to penalty (one add and one
	// wrap each) instead of being derived from slot numbers (a dozen multiplies per penalty)
	const int32_t RB = RL * 2, HB = nH * RB, B1 = n1 * RB, B2 = n2 * RB;
	const int32_t bE1 = HB, bF1 = bE1 + B1, bE2 = bF1 + B1, bF2 = bE2 + B2;
	char *const base = (char*)rows;
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
	const int32_t k0 = (S2 ? lds_extend16(lt, lq, 0, 0, min(tl, ql)) : lds_extend8(lt, lq, 0, 0, min(tl, ql))) - 1;
	if (lane == 0) *(int16_t*)(base + (center - left + 1) * 2) = (int16_t)k0;
	if (k0 == tl - 1 && k0 == ql - 1) return R;

	int32_t s = 0, wf_lo = tl + 1, wf_hi = tl + 1;
	// byte offsets (within their ring) of the rows penalty 1 writes and reads: H of penalties 1, 1-x, 1-(o1+e1), 1-(o2+e2); the E/F rings
	// have exactly e1 (e2) rows: the row read (e penalties old) is the row overwritten
	int32_t oN = RB % HB, oX = ((nH + 1 - A.pen.x) % nH) * RB, oA = ((nH + 1 - A.pen.oe1) % nH) * RB, oB = ((nH + 1 - A.pen.oe2) % nH) * RB;
	int32_t o1 = 0, o2 = 0;
	int64_t cells = 0;
	int32_t tb_used = 0;
	if (TB) M.tb_stride = 64 * NC, M.tb_left = left;
	const int32_t s_shrink = 256 - nH; // the first penalty whose good bits a shrink would read (wf_stripe_shrink, miniwfa.c:144-171)
	const int32_t cfin = ql + 1;       // the end cell (tl-1, ql-1) lies on diagonal ql-tl, i.e. this column
	// a lane's entry of chunk k: column c = center + lane + dk with dk = -32 (k + 1) (lanes 0-31) or 32 (k - 1) (lanes 32-63); entry
	// idx = c - left + 1, so entry idx - 1 lies at byte 2 (32 NC + lane) + 2 dk of a row
	const int32_t vb = 2 * (32 * NC + lane);
	const bool lower = lane < 32;
	// (Measured and dropped, round 4: penalties nobody can score — with the default costs thirteen of the first 25, a third of what a 150 bp read at 5 % runs
	// through — written dead without loads, recurrence or extension: which rows can hold a live cell follows from the penalties alone.  40 000 x 150 bp
	// 0.555 against 0.535 ms, one 200 bp call 55.8 against 54.9 us: the scalar bookkeeping costs what the skipped vector work saves.)
	for (;;) {
#ifdef MWF_LANE_TIMING // cycles per penalty of the traced pair's wave: header | chunks | footer (+ chunks run, bit 31: a skipped penalty)
		const uint64_t tm0 = __builtin_readcyclecounter();
#endif
		const int32_t lo = wf_lo > 1 ? wf_lo - 1 : 1;       // miniwfa.c:417-418
		const int32_t hi = wf_hi < cmax ? wf_hi + 1 : cmax;
		const int32_t s_new = s + 1;
		// chunks the window has reached: column c < center lies in chunk (center-1-c)/32, c >= center in chunk (c-center)/32
		const int32_t k_use = max(lo < center ? (center - 1 - lo) >> 5 : 0, hi > center ? (hi - center) >> 5 : 0);
		if (k_use >= NC || s_new >= s_shrink) { R.status = ST_BAND_OVERFLOW; break; }
		if (TB && tb_used + 64 * NC > tb_slot_bytes) { R.status = ST_TB_OVERFLOW; break; }
#ifndef MWF_LANE_TIMING
		if (trace_band && lane == 0 && s_new - 1 < dbg_cap) M.dbg[2 * (s_new - 1)] = lo, M.dbg[2 * (s_new - 1) + 1] = hi;
#endif
		uint32_t flags = 0;    // per lane, over its chunks: 1 = the lo column and live, 2 = the hi column and live, 4 = the end cell, reached
		int32_t fin_info = 0;
#ifdef MWF_LANE_TIMING
		const uint64_t tm1 = __builtin_readcyclecounter();
#endif
		{
		// The E/F row a chunk overwrites is the row the next chunk still reads at the two columns where their blocks touch: the old F of
		// this chunk's first column (lane 0) and the old E of its last (lane 63) travel to the next chunk in scalars.
		int32_t cE1 = 0, cE2 = 0, cF1 = 0, cF2 = 0;
		for (int32_t k = 0; k <= k_use; ++k) {
			const int32_t dk = lower ? -32 * (k + 1) : 32 * (k - 1);
			const int32_t c = center + lane + dk, ga = vb + 2 * dk;
			const int32_t d = c - center;
			// sources (reference wf_next_prep, miniwfa.c:252-257): entries idx-1, idx, idx+1 at byte offsets 0, 2, 4 of `ga` in a row
			const char *const pX = base + ga + oX, *const pA = base + ga + oA, *const pB = base + ga + oB;
			const int32_t hx = *(const int16_t*)(pX + 2), o2m = *(const int16_t*)pB, o2p = *(const int16_t*)(pB + 4);
			const int32_t o1m = FOLD ? kDead16 : *(const int16_t*)pA, o1p = FOLD ? kDead16 : *(const int16_t*)(pA + 4);
			char *const pE1 = base + ga + (bE1 + o1), *const pF1 = base + ga + (bF1 + o1), *const pE2 = base + ga + (bE2 + o2), *const pF2 = base + ga + (bF2 + o2);
			int32_t g1m = *(const int16_t*)pE1, g1p = *(const int16_t*)(pF1 + 4), g2m = *(const int16_t*)pE2, g2p = *(const int16_t*)(pF2 + 4);
			if (k > 0) { // lane 31: the column left of the previous chunk's first; lane 32: the column right of its last
				g1p = lane == 31 ? cF1 : g1p, g2p = lane == 31 ? cF2 : g2p;
				g1m = lane == 32 ? cE1 : g1m, g2m = lane == 32 ? cE2 : g2m;
			}
			if (k < k_use) {
				const int32_t oE1 = *(const int16_t*)(pE1 + 2), oF1 = *(const int16_t*)(pF1 + 2), oE2 = *(const int16_t*)(pE2 + 2), oF2 = *(const int16_t*)(pF2 + 2);
				cE1 = __builtin_amdgcn_readlane(oE1, 63), cE2 = __builtin_amdgcn_readlane(oE2, 63);
				cF1 = __builtin_amdgcn_readlane(oF1, 0), cF2 = __builtin_amdgcn_readlane(oF2, 0);
			}
			const bool act = c >= lo && c <= hi;
			const Cell v = wf_cell<TB>(hx, o1m, g1m, o2m, g2m, o1p, g1p, o2p, g2p);
			if (FOLD) *(int16_t*)(pE1 + 2) = (int16_t)max(act ? v.e1 : kDead16, hx), *(int16_t*)(pF1 + 2) = (int16_t)max(act ? v.f1 : kDead16, hx); // (hx >= kDead16: it was read as an int16)
			else *(int16_t*)(pE1 + 2) = (int16_t)(act ? max(v.e1, kDead16) : kDead16), *(int16_t*)(pF1 + 2) = (int16_t)(act ? max(v.f1, kDead16) : kDead16);
			*(int16_t*)(pE2 + 2) = (int16_t)(act ? max(v.e2, kDead16) : kDead16), *(int16_t*)(pF2 + 2) = (int16_t)(act ? max(v.f2, kDead16) : kDead16);
			// match extension (reference wf_extend, miniwfa.c:208-246) of the cells inside the matrix
			const bool inm = act && in_matrix(d, v.h, tl, ql);
			const int32_t j = inm ? v.h + 1 : 0, i = inm ? d + j : 0;
			const int32_t nmat = S2 ? lds_extend16(lt, lq, j, i, inm ? min(tl - j, ql - i) : 0) : lds_extend8(lt, lq, j, i, inm ? min(tl - j, ql - i) : 0);
			const int32_t h = act ? max(v.h + nmat, kDead16) : kDead16;
			*(int16_t*)(base + ga + oN + 2) = (int16_t)h;
			if (TB && act) M.tb[tb_used + (c - left)] = (uint8_t)v.tb;
			// edge rule (miniwfa.c:325-326): H is the max of the five, so "any live" == "H live"
			const uint32_t live = (uint32_t)(h >= -1);
			// termination (miniwfa.c:405-409)
			const bool fin = act && c == cfin && h == tl - 1 && in_matrix(ql - tl, h - nmat, tl, ql);
			flags |= (live & (uint32_t)(c == lo)) | ((live & (uint32_t)(c == hi)) << 1) | ((uint32_t)fin << 2);
			fin_info = fin ? (nmat == 0 ? (int32_t)(v.tb & 7u) : 0) : fin_info;
		}
		}
#ifdef MWF_LANE_TIMING
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
		const uint64_t tm2 = __builtin_readcyclecounter();
#endif
		if (__ballot(flags & 1u)) wf_lo = lo;
		if (__ballot(flags & 2u)) wf_hi = hi;
		const unsigned long long fm = __ballot(flags & 4u);
		s = s_new;
		oN = oN + RB == HB ? 0 : oN + RB, oX = oX + RB == HB ? 0 : oX + RB, oA = oA + RB == HB ? 0 : oA + RB, oB = oB + RB == HB ? 0 : oB + RB;
		o1 = o1 + RB == B1 ? 0 : o1 + RB, o2 = o2 + RB == B2 ? 0 : o2 + RB;
		if (TB) tb_used += 64 * NC;
		cells += hi - lo + 1;
		if (cells > iter_limit || s > s_limit) { // miniwfa.c:422-425
			R.status = ST_STOPPED;
			break;
		}
#ifdef MWF_LANE_TIMING
		if (trace_band && lane == 0 && s - 1 < dbg_cap) {
			const uint64_t tm3 = __builtin_readcyclecounter();
			M.dbg[2 * (s - 1)] = (int32_t)(min((uint32_t)(tm1 - tm0), 65535u) | min((uint32_t)(tm2 - tm1), 65535u) << 16);
			M.dbg[2 * (s - 1) + 1] = (int32_t)(min((uint32_t)(tm3 - tm2), 65535u) | (uint32_t)(k_use + 1) << 16);
		}
#endif
		if (fm) { R.info = __builtin_amdgcn_readlane(fin_info, (int32_t)__builtin_ctzll(fm)); break; }
	}
	R.s = s, R.cells = cells; // (no early hand-back here: a forecast after two dozen penalties is noise, and this kernel's whole run is ~100 penalties)
	return R;
}

template <bool TB, bool S2, bool FOLD = false>
__global__ __launch_bounds__(64) void wfa_lane_kernel(const BatchArgs)
{
	// the arguments are read from the kernarg segment where they are used (mwf_device.h): nothing of them stays in SGPRs across the penalties
	KArgs &A = kernel_args();
	const int32_t lane = threadIdx.x;
	const int32_t n_rows = A.pen.nH + 2 * A.pen.e1 + 2 * A.pen.e2;
	int16_t *rows = (int16_t*)lds_lane;
	uint8_t *lt = lds_lane + (n_rows * row_ints(A.lane_chunks) * 4 + 15) / 16 * 16;
	CigLocal cig_loc;
	cig_loc.base = 0, cig_loc.left = 0;
	for (int32_t round = 0;; ++round) {
		// Work counters — a set of 64, each on a cache line of its own: counter c deals the pairs c, c + 64, c + 128 ... of the order to the waves
		// with blockIdx % 64 == c — or, queue == null: a launch of one wave per pair.  ONE global counter is what bounded this kernel through
		// round 4: 40 000 read pairs are 40 000 atomics on one address, ~12.7 ns each — 0.51 of the 0.61 ms, whatever the reads (identical
		// reads, which end at penalty 0: 0.508 ms); a static deal of the pairs to the waves removes the atomics but not the imbalance (reads at
		// 2 %: 0.546 -> 0.321 ms, at 5 %: 0.603 -> 0.728; profiles/r04/lane_counter.txt).
		int32_t item = 0;
		if (A.queue) {
			const int32_t nc = min(A.queue_parts, (int32_t)gridDim.x), c = (int32_t)blockIdx.x % nc; // (a launch of fewer waves than counters: as many counters as waves)
			if (lane == 0) item = c + nc * (int32_t)atomicAdd(A.queue + 32 * c, 1);
			item = uni(item);
		} else item = round == 0 ? (int32_t)blockIdx.x : A.n_pairs;
		if (item >= A.n_pairs) break;
		const int32_t pair = A.order ? A.order[item] : item;
		PairMem M;
		pair_mem(fresh(A), (int32_t)blockIdx.x, pair, M);
		M.tl = uni(M.tl), M.ql = uni(M.ql);
		uint8_t *lq = S2 ? lt + ((M.tl >> 4) + 2) * 4 : lt + ((M.tl + 7) & ~7) + 16;
		PassResult R;
		R.status = ST_OK, R.s = 0, R.info = 0, R.n_snap = 0, R.cells = 0;
		if (S2) { // 2 bits per base; a base other than A/C/G/T: the host re-runs the pair on a byte-wise copy (ST_ALPHABET)
			uint32_t bad = lds_pack2bit<64>(M.ts, M.tl, lt);
			bad |= lds_pack2bit<64>(M.qs, M.ql, lq);
			if (__ballot(bad != 0)) R.status = ST_ALPHABET;
		} else { // both sequences into LDS as they are, eight bytes per lane and trip (the packed sequence buffer has 64 bytes of slack behind it)
			for (int32_t j = 8 * lane; j < M.tl; j += 512) *(uint64_t*)(lt + j) = ld8(M.ts + j);
			for (int32_t j = 8 * lane; j < M.ql; j += 512) *(uint64_t*)(lq + j) = ld8(M.qs + j);
		}
		__syncthreads(); // (one wave: orders the copies before the dword reads of the extension for the compiler)
		const bool trace = A.dbg && pair == A.debug_pair;
		if (R.status == ST_OK) R = lane_pass<TB, S2, FOLD>(fresh(A), M, rows, lt, lq, trace);
		if (S2) M.t2 = lt, M.q2 = lq; // the traceback's back-match stays on chip
		finish_pair(fresh(A), M, (int32_t)blockIdx.x, pair, R, R.status, 0, &cig_loc);
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

int launch_lane(const BatchArgs &a, int grid, int lds, bool seq2, void *stream)
{
	// deep rings (large gap-open costs) or a raised lane_max_len: beyond 48 KB of dynamic LDS the runtime wants to be told (the attribute is
	// per device and this may run on several host threads: set on every launch that needs it, as the band kernels do)
	const bool fold = !a.want_cigar && a.band_fold && a.pen.oe1 - a.pen.x == a.pen.e1;
	const void *fn = fold ? (seq2 ? reinterpret_cast<const void*>(&wfa_lane_kernel<false, true, true>) : reinterpret_cast<const void*>(&wfa_lane_kernel<false, false, true>)) : a.want_cigar ? (seq2 ? reinterpret_cast<const void*>(&wfa_lane_kernel<true, true>) : reinterpret_cast<const void*>(&wfa_lane_kernel<true, false>))
	                              : (seq2 ? reinterpret_cast<const void*>(&wfa_lane_kernel<false, true>) : reinterpret_cast<const void*>(&wfa_lane_kernel<false, false>));
	if (lds > 48 * 1024) {
		(void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
		(void)hipGetLastError();
	}
	if (a.want_cigar) {
		if (seq2) hipLaunchKernelGGL((wfa_lane_kernel<true, true>), dim3(grid), dim3(64), lds, (hipStream_t)stream, a);
		else hipLaunchKernelGGL((wfa_lane_kernel<true, false>), dim3(grid), dim3(64), lds, (hipStream_t)stream, a);
	} else if (fold) {
		if (seq2) hipLaunchKernelGGL((wfa_lane_kernel<false, true, true>), dim3(grid), dim3(64), lds, (hipStream_t)stream, a);
		else hipLaunchKernelGGL((wfa_lane_kernel<false, false, true>), dim3(grid), dim3(64), lds, (hipStream_t)stream, a);
	} else {
		if (seq2) hipLaunchKernelGGL((wfa_lane_kernel<false, true>), dim3(grid), dim3(64), lds, (hipStream_t)stream, a);
		else hipLaunchKernelGGL((wfa_lane_kernel<false, false>), dim3(grid), dim3(64), lds, (hipStream_t)stream, a);
	}
	return hipGetLastError() == hipSuccess ? 0 : -2;
}

int lane_kernel_occupancy(int lds, bool cigar)
{
	int n = 0;
	const hipError_t e = cigar ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, wfa_lane_kernel<true, true>, 64, lds)
	                           : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, wfa_lane_kernel<false, true>, 64, lds);
	return e == hipSuccess ? n : 0;
}

} // namespace mwf
