// mwf_coop.hip — one sequence pair across the whole device, the per-penalty hand-off form of rounds 1-2.  Since round 3 every pass
// but ONE runs on the systolic kernel (mwf_sys.hip); what is left here is the provenance pass of the two-pass low-memory mode
// (coop_pass<.., SEG>, wfa_coop_seg_kernel) and the walk back through its snapshots (coop_trace_kernel).  The description below is
// that of the whole family; the traceback / plain instantiations are no longer built.
//
// A single pair is a strictly sequential chain of penalties (reference miniwfa.c:397-426), so the only
// parallelism is across the diagonals of one wavefront — tens to hundreds of thousands of them for these
// pairs.  This kernel is the band kernel (mwf_band.hip) stretched over every CU:
//
//   * all waves of all workgroups form one wave array; 256-column chunk g belongs to wave (g mod waves),
//     slot (g / waves) mod 2 — for ever, so the H rows of a chunk are only ever touched by one CU and can
//     use ordinary cached 16-byte loads/stores, and the E/F wavefronts stay in that wave's registers;
//   * what crosses waves — the outer columns of each chunk (E1/E2 of the last, F1/F2 of the first, and the
//     H of both for the three lags) — travels as granules: aligned 8-byte {value, penalty tag} words in a ring
//     in HBM, written by one agent-scope (sc1, write-through) store and re-loaded by the one neighbour that
//     wants them until the tag is the penalty it is waiting for.  That wait is the only synchronisation
//     between neighbouring waves: no fences, no drains, no grid barrier per penalty;
//   * the only thing everybody needs per penalty is the fate of the two edge columns (does the band grow?)
//     and of the end cell: their owners publish three tagged words into a flag ring, thread 0 of every
//     workgroup polls them.  Full grid barriers remain only around the shrink, every 256 penalties.  Every
//     wait is bounded, so a lost workgroup yields an error (and a re-run on the one-workgroup kernel), not a hang;
//   * long exact-match runs (these pairs are ~99 % identical) are walked by the whole wave: 64 lanes x
//     4 bytes per trip for the lane that owns the diagonal, instead of one lane 4 bytes at a time;
//   * low-memory mode (opt.step > 0, reference miniwfa.c:437-601): with 288 GB of HBM the full 1-byte-per-
//     cell traceback of the first pass fits on the device (50.6 GB for the MHC pair), so the checkpoints
//     are read off it by walking the recorded choices back from the end cell — the same (penalty, diagonal)
//     chain the reference's provenance stripe reports — and the second pass with band resets runs exactly
//     as the reference's does (same n_iter, same CIGAR).  See checkpoint_walk().
//
// A pass is one launch; pass results travel to the next launch (walk, second pass, traceback) through
// coop_state in HBM, so the host never synchronises in between.
#include "mwf_device.h"

namespace mwf {

using namespace dev;

namespace {

constexpr int kT = 512;          // threads per workgroup: 8 waves, up to 256 VGPRs each
constexpr int kNW = kT / 64;
constexpr int kK = 2;            // chunks per wave
constexpr int kChunk = 256;
// (the bound on every wait between workgroups is A.coop_spin_limit polls: a lost workgroup yields an error, never a hang)

__device__ __forceinline__ int32_t from_left(int32_t v, int32_t fill) { return __builtin_amdgcn_update_dpp(fill, v, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ int32_t from_right(int32_t v, int32_t fill) { return __builtin_amdgcn_update_dpp(fill, v, 0x130, 0xf, 0xf, false); }

__device__ __forceinline__ int32_t ld_ag(const int32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_ag(int32_t *p, int32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Granule: one naturally aligned 8-byte {value, tag} word written by ONE agent-scope (write-through) store and read by
// agent-scope loads: the reader knows the value is the one it wants because the tag is the penalty it was computed at.
// No ordering between different granules is ever relied upon.
// (Measured and rejected: routing the seven-in-eight neighbour relations that stay inside a workgroup through LDS instead
// — the critical path runs through the workgroup boundaries either way, and the extra selects cost 5 %.)
typedef unsigned long long gran_t;
__device__ __forceinline__ gran_t ld_gran(const gran_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_gran(gran_t *p, int32_t v, int32_t tag)
{
	__hip_atomic_store(p, (gran_t)(uint32_t)v | (gran_t)(uint32_t)tag << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int32_t gran_val(gran_t g) { return (int32_t)(uint32_t)g; }
__device__ __forceinline__ int32_t gran_tag(gran_t g) { return (int32_t)(uint32_t)(g >> 32); }

constexpr int kFlagCopies = 32;  // every flag word is written to this many cache lines: 256 pollers on one line starve the line's writer
constexpr int kFlagRing = 64;    // penalties of edge/end flags kept; workgroups never drift further apart than 32 penalties
constexpr int kDriftCheck = 16;  // every so many penalties a workgroup waits until all have finished the penalty 16 back

__device__ __forceinline__ uint32_t probe4g(const PairMem &M, int32_t j, int32_t i)
{
	uint32_t a, b;
	__builtin_memcpy(&a, M.ts + j, 4);
	__builtin_memcpy(&b, M.qs + i, 4);
	return a ^ b;
}

__device__ __forceinline__ uint32_t inm_bit(int32_t d, int32_t k, int32_t tl, int32_t ql)
{
	return (uint32_t)((uint32_t)(k + 1) < (uint32_t)(tl + 1)) & (uint32_t)((uint32_t)(d + k + 1) < (uint32_t)(ql + 1));
}

__device__ __forceinline__ int32_t pick4(int32_t i, int32_t a0, int32_t a1, int32_t a2, int32_t a3)
{
	return i == 0 ? a0 : i == 1 ? a1 : i == 2 ? a2 : a3;
}

__device__ __forceinline__ unsigned long long lane_mask(int32_t base, int32_t k, int32_t a, int32_t b)
{
	int32_t lmin = a - base - k, lmax = b - base - k;
	if (lmax < 0) return 0ull;
	lmin = lmin <= 0 ? 0 : (lmin + 3) >> 2;
	lmax = min(lmax >> 2, 63);
	if (lmin > lmax) return 0ull;
	return (~0ull >> (63 - lmax)) & (~0ull << lmin);
}

// The whole wave walks one diagonal: t[j+n..] vs q[i+n..], up to `room` bytes, starting after n0 matched bytes.
// Every argument is wave-uniform; returns the total number of matching bytes (<= room).
__device__ __forceinline__ int32_t lcp_wave(const PairMem &M, int32_t j, int32_t i, int32_t room, int32_t n0)
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
		const unsigned long long stop = __ballot(m < 4); // lanes beyond `room` have m == 0 and stop the scan too
		if (stop == 0) { n += 256; continue; }
		const int32_t first = (int32_t)__builtin_ctzll(stop);
		n += 4 * first + __builtin_amdgcn_readlane(m, first);
		break;
	}
	return min(n, room);
}

// Device-wide barrier that also reduces the three per-penalty flags.
//
// Arrivals are counted on two levels: workgroups with the same (blockIdx mod 8) — which the dispatcher is observed to
// place on one XCD, though nothing here depends on it — count on their own word; the last of a group to arrive counts on
// the top word, waits there for all groups and then releases its group through the group's 64-bit generation word.
// The top word is 64 bits: arrivals in the low half, flag bits in the high half: bit 0 TOGGLES when the new low edge
// is live, bit 1 when the new high edge is (exactly one wave owns an edge column, so each toggles at most once per
// penalty and the change against the previous value is the flag); bits 2-5 become 1 | last_state<<1 once the end cell is
// reached.  A workgroup that has something to report XORs it into the high half BEFORE it arrives (returning atomic, so
// it is performed first); whoever releases a group copies the high half it saw next to the epoch, so every workgroup
// leaves the barrier knowing the flags — no agent-scope flag loads afterwards.
// `flags`: this workgroup's four LDS flag words of the penalty (or null).  `vm_keep`: how many of this wave's youngest
// memory operations may still be in flight (0 = drain everything; see the call site).  FENCE: release/acquire pair for
// data written with ordinary stores (only the shrink needs it).
// (`sync` and `lb`: the barrier words and the block index of this pair's group of workgroups — several pairs can run side
// by side on disjoint groups, each with its own words)
template <bool FENCE>
__device__ __forceinline__ bool grid_sync(const BatchArgs &A, unsigned *sync, unsigned lb, Shared &sh, unsigned &epoch, unsigned n_groups,
                                          const int32_t *flags, unsigned &cum, int32_t vm_keep)
{
	switch (vm_keep) { // the count must be an immediate
	case 3:  asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory"); break;
	case 4:  asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory"); break;
	case 8:  asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory"); break;
	case 9:  asm volatile("s_waitcnt vmcnt(9) lgkmcnt(0)" ::: "memory"); break;
	default: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); break;
	}
	__syncthreads();
	++epoch;
	if (threadIdx.x == 0) {
		if (FENCE) {
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		}
		unsigned long long *const top = (unsigned long long*)sync;                        // [0]
		int32_t *const abort_flag = (int32_t*)sync - 256 + 15;                            // this group's flags[15] (the barrier words sit 1024 bytes behind the flags)
		unsigned *const grp_cnt = sync + 16;                                               // [16 + 8*g]
		unsigned long long *const grp_gen = (unsigned long long*)(sync + 96);              // [96 + 8*g] (8-byte aligned)
		const unsigned grp = lb & 7u, n_grp = n_groups < 8u ? n_groups : 8u;
		const unsigned gsize = (n_groups - grp + 7u) / 8u;
		unsigned spins = 0;
		int32_t ok = 1;
		unsigned long long seen = 0;
		unsigned mine = 0;
		if (flags) mine = (unsigned)(flags[0] != 0) | (unsigned)(flags[1] != 0) << 1 | (flags[2] != 0 ? (1u | (unsigned)flags[3] << 1) << 2 : 0u);
		if (mine) (void)__hip_atomic_fetch_xor(top, (unsigned long long)mine << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		const unsigned old = __hip_atomic_fetch_add(&grp_cnt[8 * grp], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (old + 1 == gsize * epoch) {
			(void)__hip_atomic_fetch_add(top, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			for (;;) {
				seen = __hip_atomic_load(top, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				if ((unsigned)(seen & 0xffffffffu) >= n_grp * epoch) break;
				__builtin_amdgcn_s_sleep(1);
				if (++spins > A.coop_spin_limit || ((spins & 255u) == 0 && ld_ag(abort_flag))) { ok = 0; break; }
			}
			// (a leader that gave up publishes a POISONED generation: its members leave the barrier knowing that it did not complete)
			__hip_atomic_store(&grp_gen[4 * grp], (seen & 0xffffffff00000000ull) | (ok ? epoch : (epoch | 0x80000000u)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		} else {
			for (;;) {
				seen = __hip_atomic_load(&grp_gen[4 * grp], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				if ((unsigned)(seen & 0xffffffffu) >= epoch) { if (seen & 0x80000000ull) ok = 0; break; }
				__builtin_amdgcn_s_sleep(1);
				if (++spins > A.coop_spin_limit || ((spins & 255u) == 0 && ld_ag(abort_flag))) { ok = 0; break; }
			}
		}
		if (FENCE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
		if (!ok) st_ag(abort_flag, 1); // whoever gives up first releases everybody else at once
		sh.word[3] = ok;
		sh.word[2] = (int32_t)(seen >> 32);
	}
	__syncthreads();
	cum = (unsigned)uni(sh.word[2]);
	return uni(sh.word[3]) != 0;
}

// grp / lb / G: this pair's group of workgroups, this workgroup's index in it, the group's size (the whole grid when one pair
// has the device to itself)
// SEG: first pass of the true low-memory mode (reference mwf_wfa_seg, miniwfa.c:551-601): no traceback bytes are stored;
// every wavefront value carries the index of the cell its optimal predecessor chain went through at the last snapshot
// (shadow registers / shadow H rows / shadow granules, moved by the choices the traceback byte records, :495-526), and
// every `step` penalties the shadow ring is flattened into a snapshot and renumbered (:451-474), see coop_snapshot().
template <int E1, int E2, bool TB, bool SEG = false>
__device__ PassResult coop_pass(const BatchArgs &A, const PairMem &M, Shared &sh, int32_t n_seg, int32_t grp, int32_t lb, int32_t G)
{
	static_assert(!(TB && SEG), "the first pass of the low-memory mode stores no traceback");
	constexpr bool WTB = TB || SEG; // the recurrence yields the traceback byte
	const int32_t NWt = G * kNW, TC = NWt * kK;
	const int32_t tl = M.tl, ql = M.ql, cmax = tl + ql + 1;
	const int32_t tid = threadIdx.x, lane = tid & 63, gw = uni(lb * kNW + (tid >> 6));
	const bool lead = lb == 0 && tid == 0;
	char *const misc = (char*)A.coop_flags + (int64_t)grp * A.coop_misc_stride; // this group's flags | barrier words | pass state | flag ring
	unsigned *const sync = (unsigned*)(misc + 1024);
	const int64_t W = A.W;
	const int32_t nH = A.pen.nH, lagx = A.pen.x, lag1 = A.pen.oe1, lag2 = A.pen.oe2;
	const bool relaxed_stores = min(lagx, min(lag1, lag2)) >= 3;
	// some H lag is 1 (edit-distance preset: x = o+e = 1): the rows of the next penalty include the one this penalty writes,
	// so a chunk's next rows are requested only after its own store has completed
	const bool lag_one = min(lagx, min(lag1, lag2)) < 2;
	int32_t *const H = M.H;
	// What crosses waves.  granule(row, r, side, which): per H slot (= penalty mod nH), chunk slot r and side (0: the chunk's
	// last column, written by lane 63; 1: its first column, lane 0): which 0 = E1 | F1, 1 = E2 | F2, 2 = H after extension.
	gran_t *const grans = (gran_t*)(A.coop_edge + (int64_t)grp * A.coop_edge_stride);
	auto granule = [&](int32_t row, int32_t r, int32_t side, int32_t which) -> gran_t* { return grans + ((((int64_t)row * TC + r) * 2 + side) * 4 + which); };
	// SEG: the same for the provenance values, in a second array of the same shape
	gran_t *const sgrans = (gran_t*)(A.coop_edge + (int64_t)grp * A.coop_edge_stride + A.coop_sedge_off);
	auto sgranule = [&](int32_t row, int32_t r, int32_t side, int32_t which) -> gran_t* { return sgrans + ((((int64_t)row * TC + r) * 2 + side) * 4 + which); };
	int32_t *const sH = M.sH;
	int32_t *const gflags = (int32_t*)misc;                // [12..14]: origin offset, shrink reduction; [15]: set by the first workgroup that gives up a wait
	// Per penalty (mod kFlagRing): "new low edge live", "new high edge live", "end cell reached | last state << 1", each as
	// penalty << 4 | value, written by the one wave that owns the column in question.
	int32_t *const fring = (int32_t*)misc + 1024;
	auto flag_entry = [&](int32_t pen, int32_t copy) -> int32_t* { return fring + ((pen & (kFlagRing - 1)) * kFlagCopies + copy) * 32; }; // one 128-byte line each
	unsigned long long *const arrived = (unsigned long long*)(sync + 200); // workgroup-penalties finished (drift bound)
	PassResult R;
	R.status = ST_OK, R.s = 0, R.info = 0, R.n_snap = 0, R.cells = 0;
	unsigned epoch = 0, cum = 0; // the host zeroes the barrier words before every pass

	int32_t e1h[E1][kK][4], f1h[E1][kK][4], e2h[E2][kK][4], f2h[E2][kK][4];
#pragma unroll
	for (int k = 0; k < kK; ++k)
#pragma unroll
		for (int i = 0; i < 4; ++i) {
#pragma unroll
			for (int a = 0; a < E1; ++a) e1h[a][k][i] = f1h[a][k][i] = kNegInf;
#pragma unroll
			for (int a = 0; a < E2; ++a) e2h[a][k][i] = f2h[a][k][i] = kNegInf;
		}
	int4 phx[kK], po1[kK], po2[kK];
	gran_t ph1[kK], ph2[kK];       // lanes 0 / 63: H granules of the neighbouring chunk's adjacent column for lags o1+e1, o2+e2
	// SEG: provenance of the same values (dead code otherwise)
	int32_t se1h[E1][kK][4], sf1h[E1][kK][4], se2h[E2][kK][4], sf2h[E2][kK][4];
	int4 sphx[kK], spo1[kK], spo2[kK];
	gran_t sph1[kK], sph2[kK];
	if (SEG) {
#pragma unroll
		for (int k = 0; k < kK; ++k)
#pragma unroll
			for (int i = 0; i < 4; ++i) {
#pragma unroll
				for (int a = 0; a < E1; ++a) se1h[a][k][i] = sf1h[a][k][i] = kNegInf;
#pragma unroll
				for (int a = 0; a < E2; ++a) se2h[a][k][i] = sf2h[a][k][i] = kNegInf;
			}
	}

	// ---- penalty 0: origin and its extension (the first wave of workgroup 0 walks it cooperatively)
	if (tid == 0) {
		for (int32_t j = 0; j < nH; ++j) sh.rng_lo[j] = 1, sh.rng_hi[j] = 0;
		sh.rng_lo[0] = sh.rng_hi[0] = tl + 1;
		sh.red[0] = 0; // set by a wave whose wait for a neighbour ran into the spin limit
	}
	if (lb == 0 && tid < 64) {
		const int32_t k0 = lcp_wave(M, 0, 0, min(tl, ql), 0) - 1;
		if (tid == 0) {
			const int32_t c0 = tl + 1;
			st_ag(&H[c0], k0); // its owner is some other workgroup's wave: write through
			st_ag(&gflags[12], k0);
			// the origin's chunk edges for the first lagged reads
			const int32_t r0 = (c0 >> 8) % TC;
			if ((c0 & 255) == 0) st_gran(granule(0, r0, 1, 2), k0, 0);
			if ((c0 & 255) == 255) st_gran(granule(0, r0, 0, 2), k0, 0);
			if (SEG) { // the origin's provenance is -1 (miniwfa.c:119, :542)
				st_ag(&sH[c0], -1);
				if ((c0 & 255) == 0) st_gran(sgranule(0, r0, 1, 2), -1, 0);
				if ((c0 & 255) == 255) st_gran(sgranule(0, r0, 0, 2), -1, 0);
			}
		}
	}
	if (tid == 0) for (int32_t j = 0; j < 12; ++j) (&sh.flags[0][0])[j] = 0;
	if (!grid_sync<false>(A, sync, (unsigned)lb, sh, epoch, G, nullptr, cum, 0)) { R.status = ST_INTERNAL; return R; }
	{
		const int32_t k0 = uni(ld_ag(&gflags[12]));
		if (k0 == tl - 1 && k0 == ql - 1) { R.cells = 0; R.info = SEG ? -1 : 0; return R; }
	}

	int32_t s = 0, wf_lo = tl + 1, wf_hi = tl + 1;
	int32_t curH = 0, par = 0, sid = 0;
	int64_t cells = 0, tb_used = 0;
	int32_t snap_ctr = A.step == 1 ? 0 : 1, n_snap = 0; // SEG: (s+1) % step, snapshots taken
	int64_t snap_used = 0;

	auto prefetch = [&](int k, int32_t slotH, int32_t phi, int32_t g_lo) {
		int32_t jx = slotH - lagx; if (jx < 0) jx += nH;
		int32_t j1 = slotH - lag1; if (j1 < 0) j1 += nH;
		int32_t j2 = slotH - lag2; if (j2 < 0) j2 += nH;
		const int32_t r = gw + NWt * k;
		int32_t g = g_lo - g_lo % TC + r;
		if (g < g_lo) g += TC;
		const bool in = g * kChunk <= phi;
		const int32_t c0 = in ? g * kChunk + 4 * lane : 0;
		phx[k] = *(const int4*)(H + (jx * W + c0));
		po1[k] = *(const int4*)(H + (j1 * W + c0));
		po2[k] = *(const int4*)(H + (j2 * W + c0));
		// neighbours' outer columns: lane 0 wants the LAST column of the chunk to the left, lane 63 the FIRST of the one to the right
		const int32_t rn = lane == 0 ? (r == 0 ? TC - 1 : r - 1) : (r + 1 == TC ? 0 : r + 1);
		const int32_t side = lane == 0 ? 0 : 1;
		ph1[k] = ld_gran(granule(j1, rn, side, 2));
		ph2[k] = ld_gran(granule(j2, rn, side, 2));
		if (SEG) {
			sphx[k] = *(const int4*)(sH + (jx * W + c0));
			spo1[k] = *(const int4*)(sH + (j1 * W + c0));
			spo2[k] = *(const int4*)(sH + (j2 * W + c0));
			sph1[k] = ld_gran(sgranule(j1, rn, side, 2));
			sph2[k] = ld_gran(sgranule(j2, rn, side, 2));
		}
	};
	int32_t gl;
	{
		const int32_t lo1 = wf_lo > 1 ? wf_lo - 1 : 1, hi1 = wf_hi < cmax ? wf_hi + 1 : cmax;
		gl = lo1 >> 8;
#pragma unroll
		for (int k = 0; k < kK; ++k) prefetch(k, 1, hi1, gl);
	}

#ifdef MWF_BAND_TIMING
	unsigned long long t_acc[4] = {0, 0, 0, 0}, t_steps = 0, t_active = 0;
#endif
	for (;;) {
#ifdef MWF_BAND_TIMING
		const unsigned long long t_a = __builtin_readcyclecounter();
#endif
		if (TB && sid < n_seg) { // checkpoint reset of the second pass (miniwfa.c:413-416)
			if (uni(M.seg[2 * sid]) == s) {
				const int32_t c = uni(M.seg[2 * sid + 1]);
				if (c < wf_lo || c > wf_hi) { R.status = ST_INTERNAL; break; }
				wf_lo = wf_hi = c;
				++sid;
			}
		}
		if (SEG) { // snapshot when (s+1) % step == 0, before slice s+1 is computed (miniwfa.c:585-586)
			if (snap_ctr == 0) {
				// ---- flatten the shadow ring and renumber it (reference wf_snapshot1, miniwfa.c:451-474).  Index of a cell:
				// slice id * Wspan + (column - slo), slice ids: H ring slot j -> j, then E1, F1, E2, F2 by age.
				if (!grid_sync<true>(A, sync, (unsigned)lb, sh, epoch, G, nullptr, cum, 0)) { R.status = ST_INTERNAL; break; }
				int32_t slo = 0x7fffffff, shi = -1;
				for (int32_t j = 0; j < nH; ++j) {
					const int32_t l = uni(sh.rng_lo[j]), h = uni(sh.rng_hi[j]);
					if (l <= h) slo = min(slo, l), shi = max(shi, h);
				}
				const int32_t Wspan = shi - slo + 1;
				constexpr int32_t NS_EF = 2 * E1 + 2 * E2;
				const int64_t total = (int64_t)(nH + NS_EF) * Wspan;
				if ((int64_t)(n_snap + 1) * 8 > A.snap_meta_slot || snap_used + total > A.snap_slot_ints || total > 0x7fffffffLL) { R.status = ST_SNAP_OVERFLOW; break; }
				int32_t *const x = M.snap + snap_used;
				if (lead) {
					int32_t *meta = M.snap_meta + (int64_t)n_snap * 8;
					meta[0] = (int32_t)(snap_used & 0xffffffff), meta[1] = (int32_t)(snap_used >> 32);
					meta[2] = s, meta[3] = curH, meta[4] = slo, meta[5] = Wspan;
				}
				const int32_t gbase_s = gl - gl % TC;
#pragma unroll
				for (int k = 0; k < kK; ++k) {
					const int32_t r = gw + NWt * k;
					int32_t g = gbase_s + r;
					if (g < gl) g += TC;
					const int32_t cb = g * kChunk, c0 = cb + 4 * lane;
					if (cb > shi || cb + kChunk - 1 < slo) continue; // uniform
					// H rows of the ring
					for (int32_t j = 0; j < nH; ++j) {
						const int32_t l = uni(sh.rng_lo[j]), h = uni(sh.rng_hi[j]);
						if (l > h || cb > h || cb + kChunk - 1 < l) continue; // uniform
						int32_t age = curH - j; if (age < 0) age += nH;
						int4 v = *(const int4*)(sH + (j * W + c0));
						int32_t vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
						for (int i = 0; i < 4; ++i) {
							const int32_t c = c0 + i;
							if (c >= l && c <= h) {
								const int32_t t = j * Wspan + (c - slo);
								x[t] = vv[i];
								vv[i] = t;
							}
						}
						*(int4*)(sH + (j * W + c0)) = make_int4(vv[0], vv[1], vv[2], vv[3]);
						// the copies of the outer columns the neighbouring waves read
						if (lane == 63 && c0 + 3 >= l && c0 + 3 <= h) st_gran(sgranule(j, r, 0, 2), vv[3], s - age);
						if (lane == 0 && c0 >= l && c0 <= h) st_gran(sgranule(j, r, 1, 2), vv[0], s - age);
					}
					// E/F registers: age a holds penalty s - a, whose window is that of H ring slot curH - a
					auto flat = [&](int32_t (&reg)[4], int32_t id, int32_t a, int32_t which, bool is_e) {
						int32_t j = curH - a; if (j < 0) j += nH;
						if (s - a < 0) return;
						const int32_t l = uni(sh.rng_lo[j]), h = uni(sh.rng_hi[j]);
#pragma unroll
						for (int i = 0; i < 4; ++i) {
							const int32_t c = c0 + i;
							if (c >= l && c <= h) {
								const int32_t t = id * Wspan + (c - slo);
								x[t] = reg[i];
								reg[i] = t;
							}
						}
						if (is_e && lane == 63 && c0 + 3 >= l && c0 + 3 <= h) st_gran(sgranule(j, r, 0, which), reg[3], s - a);
						if (!is_e && lane == 0 && c0 >= l && c0 <= h) st_gran(sgranule(j, r, 1, which), reg[0], s - a);
					};
#pragma unroll
					for (int a = 0; a < E1; ++a) flat(se1h[a][k], nH + a, a, 0, true), flat(sf1h[a][k], nH + E1 + a, a, 0, false);
#pragma unroll
					for (int a = 0; a < E2; ++a) flat(se2h[a][k], nH + 2 * E1 + a, a, 1, true), flat(sf2h[a][k], nH + 2 * E1 + E2 + a, a, 1, false);
				}
				snap_used += total, ++n_snap;
				if (!grid_sync<true>(A, sync, (unsigned)lb, sh, epoch, G, nullptr, cum, 0)) { R.status = ST_INTERNAL; break; }
				// what was requested for the next penalty before the renumbering is stale: request it again
				{
					const int32_t lo_n = wf_lo > 1 ? wf_lo - 1 : 1, hi_n = wf_hi < cmax ? wf_hi + 1 : cmax;
					const int32_t nH1 = curH + 1 == nH ? 0 : curH + 1;
#pragma unroll
					for (int k = 0; k < kK; ++k) prefetch(k, nH1, hi_n, gl);
					(void)lo_n;
				}
			}
			snap_ctr = snap_ctr + 1 == A.step ? 0 : snap_ctr + 1;
		}
		const int32_t lo = wf_lo > 1 ? wf_lo - 1 : 1;
		const int32_t hi = wf_hi < cmax ? wf_hi + 1 : cmax;
		const int32_t s_new = s + 1;
		const int32_t newH = curH + 1 == nH ? 0 : curH + 1;
		const int32_t npar = par + 1 == 3 ? 0 : par + 1;
		const int32_t origin = lo & ~3;
		const int32_t row_bytes = (hi | 3) - origin + 1;
		if (TB) {
			if (s_new - 1 >= A.rows_slot) { R.status = ST_ROWS_OVERFLOW; break; }
			if (tb_used + row_bytes > A.tb_slot_bytes) { R.status = ST_TB_OVERFLOW; break; }
		}
		const int32_t plo = lo > 1 ? lo - 1 : 1, phi = hi < cmax ? hi + 1 : cmax;
		const int32_t gl_next = plo >> 8;
		if ((phi >> 8) - gl_next + 1 > TC - 1) { R.status = ST_BAND_OVERFLOW; break; }
		const int32_t nextH = newH + 1 == nH ? 0 : newH + 1;
		int32_t jx = newH - lagx; if (jx < 0) jx += nH;
		int32_t j1 = newH - lag1; if (j1 < 0) j1 += nH;
		int32_t j2 = newH - lag2; if (j2 < 0) j2 += nH;
		int32_t jg1 = newH - E1;  if (jg1 < 0) jg1 += nH;
		int32_t jg2 = newH - E2;  if (jg2 < 0) jg2 += nH;
		const int32_t xlo = uni(sh.rng_lo[jx]), xhi = uni(sh.rng_hi[jx]);
		const int32_t alo = uni(sh.rng_lo[j1]), ahi = uni(sh.rng_hi[j1]);
		const int32_t blo = uni(sh.rng_lo[j2]), bhi = uni(sh.rng_hi[j2]);
		const int32_t p1lo = uni(sh.rng_lo[jg1]), p1hi = uni(sh.rng_hi[jg1]); // window of penalty s_new-e1 (E1/F1 sources)
		const int32_t p2lo = uni(sh.rng_lo[jg2]), p2hi = uni(sh.rng_hi[jg2]);
		const int32_t ilo = max(max(lo, xlo), max(alo, blo) + 1), ihi = min(min(hi, xhi), min(ahi, bhi) - 1);
		const bool track_good = (((256 - (s_new & 255)) & 255) < nH);
		const int32_t cfin = ql + 1; // the end cell (tl-1, ql-1) lies on diagonal ql-tl, i.e. in this column

		if (tid == 0) {
			sh.rng_lo[newH] = lo, sh.rng_hi[newH] = hi;
		}
		if (lead) {
			if (TB) M.row_off[s_new - 1] = tb_used, M.row_lo[s_new - 1] = origin;
#ifndef MWF_BAND_TIMING // (the timing build keeps hand-off timestamps in the trace buffer instead)
			if (A.dbg && s_new - 1 < A.dbg_cap) M.dbg[2 * (s_new - 1)] = lo, M.dbg[2 * (s_new - 1) + 1] = hi;
#endif
		}

#ifdef MWF_BAND_TIMING
		const unsigned long long t_b = __builtin_readcyclecounter();
#endif
		const int32_t gbase = gl - gl % TC;
		bool act0 = false, act1 = false;
#pragma unroll
		for (int k = 0; k < kK; ++k) {
			const int32_t r = gw + NWt * k;
			int32_t g = gbase + r;
			if (g < gl) g += TC;
			const int32_t cb = g * kChunk;
			const bool active = cb <= hi && cb + kChunk - 1 >= lo;
			int32_t ne1[4], nf1[4], ne2[4], nf2[4];
			int32_t sne1s[4], snf1s[4], sne2s[4], snf2s[4]; // SEG: the new provenance values of this chunk
			if (active) {
				if (k == 0) act0 = true; else act1 = true;
				const int32_t c0 = cb + 4 * lane;
				const bool inner = cb >= ilo && cb + kChunk - 1 <= ihi;
				int32_t hx[4] = {phx[k].x, phx[k].y, phx[k].z, phx[k].w};
				int32_t o1[6], o2[6];
				o1[1] = po1[k].x, o1[2] = po1[k].y, o1[3] = po1[k].z, o1[4] = po1[k].w;
				o2[1] = po2[k].x, o2[2] = po2[k].y, o2[3] = po2[k].z, o2[4] = po2[k].w;
				gran_t gh1 = ph1[k], gh2 = ph2[k];
				int32_t shx[4], so1[6], so2[6];
				gran_t sgh1 = 0, sgh2 = 0;
				if (SEG) {
					shx[0] = sphx[k].x, shx[1] = sphx[k].y, shx[2] = sphx[k].z, shx[3] = sphx[k].w;
					so1[1] = spo1[k].x, so1[2] = spo1[k].y, so1[3] = spo1[k].z, so1[4] = spo1[k].w;
					so2[1] = spo2[k].x, so2[2] = spo2[k].y, so2[3] = spo2[k].z, so2[4] = spo2[k].w;
					sgh1 = sph1[k], sgh2 = sph2[k];
				}
				// What the neighbouring chunks computed for the column next to this chunk (lane 0: left neighbour's last column,
				// lane 63: right neighbour's first): E1|F1 of penalty s_new-e1, E2|F2 of s_new-e2, H of s_new-lag1 and s_new-lag2.
				// A neighbour column outside the window of that penalty was never computed: NEG_INF.  Otherwise wait for the
				// granule carrying that penalty's tag — this wait is the only synchronisation between neighbouring waves.
				const int32_t rl = r == 0 ? TC - 1 : r - 1, rr = r + 1 == TC ? 0 : r + 1;
				int32_t xg1, xg2, v1, v2;
				int32_t sxg1 = kNegInf, sxg2 = kNegInf, sv1 = kNegInf, sv2 = kNegInf;
				{
					const int32_t nb = lane == 0 ? rl : rr, side = lane == 0 ? 0 : 1, cn = lane == 0 ? cb - 1 : cb + kChunk;
					const bool edge_lane = lane == 0 || lane == 63;
					const bool need_e1 = edge_lane & (cn >= p1lo) & (cn <= p1hi), need_e2 = edge_lane & (cn >= p2lo) & (cn <= p2hi);
					const bool need_h1 = edge_lane & (cn >= alo) & (cn <= ahi), need_h2 = edge_lane & (cn >= blo) & (cn <= bhi);
					// (Loading these at the end of the previous penalty instead, to overlap the wait for the flags, measured slower:
					// memory operations return in issue order, so an early agent-scope load holds up whatever is issued behind it.)
#ifdef MWF_BAND_TIMING
					const unsigned long long t_g0 = __builtin_readcyclecounter();
#endif
					gran_t ge1 = ld_gran(granule(jg1, nb, side, 0)), ge2 = ld_gran(granule(jg2, nb, side, 1));
					gran_t sge1 = 0, sge2 = 0;
					if (SEG) sge1 = ld_gran(sgranule(jg1, nb, side, 0)), sge2 = ld_gran(sgranule(jg2, nb, side, 1));
					for (unsigned spins = 0;; ++spins) {
						bool late = (need_e1 && gran_tag(ge1) != s_new - E1) || (need_e2 && gran_tag(ge2) != s_new - E2) ||
						            (need_h1 && gran_tag(gh1) != s_new - lag1) || (need_h2 && gran_tag(gh2) != s_new - lag2);
						if (SEG) late = late || (need_e1 && gran_tag(sge1) != s_new - E1) || (need_e2 && gran_tag(sge2) != s_new - E2) ||
						                (need_h1 && gran_tag(sgh1) != s_new - lag1) || (need_h2 && gran_tag(sgh2) != s_new - lag2);
						if (!__ballot(late)) break;
						if (spins > A.coop_spin_limit || ((spins & 255u) == 255u && uni(ld_ag(&gflags[15])))) {
							if (lane == 0) sh.red[0] = 1, st_ag(&gflags[15], 1);
							break;
						}
						__builtin_amdgcn_s_sleep(1);
						ge1 = ld_gran(granule(jg1, nb, side, 0)), ge2 = ld_gran(granule(jg2, nb, side, 1));
						gh1 = ld_gran(granule(j1, nb, side, 2)), gh2 = ld_gran(granule(j2, nb, side, 2));
						if (SEG) {
							sge1 = ld_gran(sgranule(jg1, nb, side, 0)), sge2 = ld_gran(sgranule(jg2, nb, side, 1));
							sgh1 = ld_gran(sgranule(j1, nb, side, 2)), sgh2 = ld_gran(sgranule(j2, nb, side, 2));
						}
					}
#ifdef MWF_BAND_TIMING
					if (__ballot(gran_tag(ge1) == 0x7fffffff) == 0) t_acc[3] += __builtin_readcyclecounter() - t_g0; // (forces the wait here)
#endif
					xg1 = need_e1 ? gran_val(ge1) : kNegInf, xg2 = need_e2 ? gran_val(ge2) : kNegInf;
					v1 = need_h1 ? gran_val(gh1) : kNegInf, v2 = need_h2 ? gran_val(gh2) : kNegInf;
					if (SEG) {
						sxg1 = need_e1 ? gran_val(sge1) : kNegInf, sxg2 = need_e2 ? gran_val(sge2) : kNegInf;
						sv1 = need_h1 ? gran_val(sgh1) : kNegInf, sv2 = need_h2 ? gran_val(sgh2) : kNegInf;
					}
				}
				if (!inner) {
#pragma unroll
					for (int i = 0; i < 4; ++i) {
						const int32_t c = c0 + i;
						hx[i] = ((c >= xlo) & (c <= xhi)) ? hx[i] : kNegInf;
						o1[i + 1] = ((c >= alo) & (c <= ahi)) ? o1[i + 1] : kNegInf;
						o2[i + 1] = ((c >= blo) & (c <= bhi)) ? o2[i + 1] : kNegInf;
						if (SEG) {
							shx[i] = ((c >= xlo) & (c <= xhi)) ? shx[i] : kNegInf;
							so1[i + 1] = ((c >= alo) & (c <= ahi)) ? so1[i + 1] : kNegInf;
							so2[i + 1] = ((c >= blo) & (c <= bhi)) ? so2[i + 1] : kNegInf;
						}
					}
				}
				o1[0] = from_left(o1[4], v1), o1[5] = from_right(o1[1], v1);
				o2[0] = from_left(o2[4], v2), o2[5] = from_right(o2[1], v2);
				int32_t g1m[4], g1p[4], g2m[4], g2p[4];
				g1m[0] = from_left(e1h[E1 - 1][k][3], xg1);   // lane 0 keeps its own xg1: the left neighbour's E1
				g2m[0] = from_left(e2h[E2 - 1][k][3], xg2);
				g1p[3] = from_right(f1h[E1 - 1][k][0], xg1);  // lane 63 keeps its own xg1: the right neighbour's F1
				g2p[3] = from_right(f2h[E2 - 1][k][0], xg2);
#pragma unroll
				for (int i = 1; i < 4; ++i) g1m[i] = e1h[E1 - 1][k][i - 1], g2m[i] = e2h[E2 - 1][k][i - 1];
#pragma unroll
				for (int i = 0; i < 3; ++i) g1p[i] = f1h[E1 - 1][k][i + 1], g2p[i] = f2h[E2 - 1][k][i + 1];
				int32_t sg1m[4], sg1p[4], sg2m[4], sg2p[4];
				if (SEG) {
					so1[0] = from_left(so1[4], sv1), so1[5] = from_right(so1[1], sv1);
					so2[0] = from_left(so2[4], sv2), so2[5] = from_right(so2[1], sv2);
					sg1m[0] = from_left(se1h[E1 - 1][k][3], sxg1), sg2m[0] = from_left(se2h[E2 - 1][k][3], sxg2);
					sg1p[3] = from_right(sf1h[E1 - 1][k][0], sxg1), sg2p[3] = from_right(sf2h[E2 - 1][k][0], sxg2);
#pragma unroll
					for (int i = 1; i < 4; ++i) sg1m[i] = se1h[E1 - 1][k][i - 1], sg2m[i] = se2h[E2 - 1][k][i - 1];
#pragma unroll
					for (int i = 0; i < 3; ++i) sg1p[i] = sf1h[E1 - 1][k][i + 1], sg2p[i] = sf2h[E2 - 1][k][i + 1];
				}

				int32_t hv[4], room[4], nmat[4];
				int32_t sne1[4], snf1[4], sne2[4], snf2[4], shv[4];
				uint32_t tbw = 0, pend = 0, live = 0, fin = 0, gbits = 0;
#pragma unroll
				for (int i = 0; i < 4; ++i) {
					const int32_t c = c0 + i, d = c - 1 - tl;
					const uint32_t act = inner ? 1u : (uint32_t)((c >= lo) & (c <= hi));
					const Cell v = wf_cell<WTB>(hx[i], o1[i], g1m[i], o2[i], g2m[i], o1[i + 2], g1p[i], o2[i + 2], g2p[i]);
					ne1[i] = act ? v.e1 : kNegInf, nf1[i] = act ? v.f1 : kNegInf;
					ne2[i] = act ? v.e2 : kNegInf, nf2[i] = act ? v.f2 : kNegInf;
					if (SEG) { // provenance follows the choices the traceback byte records (miniwfa.c:504-523)
						const Cell u = shadow_cell(v.tb, shx[i], so1[i], sg1m[i], so2[i], sg2m[i], so1[i + 2], sg1p[i], so2[i + 2], sg2p[i]);
						sne1[i] = act ? u.e1 : kNegInf, snf1[i] = act ? u.f1 : kNegInf;
						sne2[i] = act ? u.e2 : kNegInf, snf2[i] = act ? u.f2 : kNegInf;
						shv[i] = act ? u.h : kNegInf;
					}
					const uint32_t inm = act & inm_bit(d, v.h, tl, ql);
					if (track_good)
						gbits |= (act & (inm | inm_bit(d, v.e1, tl, ql) | inm_bit(d, v.f1, tl, ql) | inm_bit(d, v.e2, tl, ql) | inm_bit(d, v.f2, tl, ql))) << i;
					const uint32_t lv = act & (uint32_t)(v.h >= -1);
					live |= (lv & (uint32_t)(c == lo)) | ((lv & (uint32_t)(c == hi)) << 1);
					const int32_t j = inm ? v.h + 1 : 0, q = inm ? d + v.h + 1 : 0;
					room[i] = inm ? min(tl - j, ql - q) : 0;
					const uint32_t x = probe4g(M, j, q);
					nmat[i] = min(x ? (int32_t)(__builtin_ctz(x) >> 3) : 4, room[i]);
					pend |= ((uint32_t)(x == 0) & (uint32_t)(room[i] > 4)) << i;
					hv[i] = v.h;
					tbw |= v.tb << (8 * i);
				}
				// E/F of the outer columns and the edge flags are final: publish them now, the write-through overlaps the probes
				if (lane == 63) st_gran(granule(newH, r, 0, 0), ne1[3], s_new), st_gran(granule(newH, r, 0, 1), ne2[3], s_new);
				if (lane == 0) st_gran(granule(newH, r, 1, 0), nf1[0], s_new), st_gran(granule(newH, r, 1, 1), nf2[0], s_new);
				if (SEG) {
					if (lane == 63) st_gran(sgranule(newH, r, 0, 0), sne1[3], s_new), st_gran(sgranule(newH, r, 0, 1), sne2[3], s_new);
					if (lane == 0) st_gran(sgranule(newH, r, 1, 0), snf1[0], s_new), st_gran(sgranule(newH, r, 1, 1), snf2[0], s_new);
#pragma unroll
					for (int i = 0; i < 4; ++i) sne1s[i] = sne1[i], snf1s[i] = snf1[i], sne2s[i] = sne2[i], snf2s[i] = snf2[i];
				}
				if ((uint32_t)(lo - cb) < (uint32_t)kChunk) { // this chunk holds the low edge column
					const int32_t lv = __ballot(live & 1u) != 0;
					if (lane < kFlagCopies) st_ag(flag_entry(s_new, lane) + 0, s_new << 4 | lv);
#ifdef MWF_BAND_TIMING
					if (lane == 0 && A.dbg && s_new < A.dbg_cap / 4) ((unsigned long long*)A.dbg)[4 * s_new + 0] = __builtin_amdgcn_s_memrealtime(); // low edge published
#endif
				}
				if ((uint32_t)(hi - cb) < (uint32_t)kChunk) {
					const int32_t lv = __ballot(live & 2u) != 0;
					if (lane < kFlagCopies) st_ag(flag_entry(s_new, lane) + 1, s_new << 4 | lv);
				}
				if (!lag_one) prefetch(k, nextH, phi, gl_next); // the next penalty's rows: only now, so that nothing queues in front of what was just published
				// a run of >= 4 matches continues: the wave walks it together, one owning lane and column at a time
				unsigned long long owners = __ballot(pend != 0);
				while (owners) {
					const int32_t src = (int32_t)__builtin_ctzll(owners);
					owners &= owners - 1;
					uint32_t bits = (uint32_t)__builtin_amdgcn_readlane((int32_t)pend, src);
					const int32_t c0s = __builtin_amdgcn_readlane(c0, src);
					while (bits) {
						const int32_t ii = (int32_t)__builtin_ctz(bits);
						bits &= bits - 1;
						const int32_t hh = __builtin_amdgcn_readlane(pick4(ii, hv[0], hv[1], hv[2], hv[3]), src);
						const int32_t rm = __builtin_amdgcn_readlane(pick4(ii, room[0], room[1], room[2], room[3]), src);
						const int32_t j = hh + 1, q = c0s + ii - 1 - tl + j;
						const int32_t n = lcp_wave(M, j, q, rm, 4);
#pragma unroll
						for (int i = 0; i < 4; ++i) nmat[i] = (ii == i && lane == src) ? n : nmat[i];
					}
				}
				int32_t done_info = 0;
#pragma unroll
				for (int i = 0; i < 4; ++i) {
					const int32_t d = c0 + i - 1 - tl, kk = hv[i] + nmat[i];
					const uint32_t act = inner ? 1u : (uint32_t)((c0 + i >= lo) & (c0 + i <= hi));
					const uint32_t f = act & inm_bit(d, hv[i], tl, ql) & (uint32_t)(kk == tl - 1) & (uint32_t)(d + kk == ql - 1);
					fin |= f;
					done_info = f ? (SEG ? shv[i] : (nmat[i] == 0 ? (int32_t)((tbw >> (8 * i)) & 7u) : 0)) : done_info;
					hv[i] = kk;
				}
				*(int4*)(H + (newH * W + c0)) = make_int4(hv[0], hv[1], hv[2], hv[3]);
				if (SEG) {
					*(int4*)(sH + (newH * W + c0)) = make_int4(shv[0], shv[1], shv[2], shv[3]);
					if (lane == 63) st_gran(sgranule(newH, r, 0, 2), shv[3], s_new);
					if (lane == 0) st_gran(sgranule(newH, r, 1, 2), shv[0], s_new);
				}
				if (TB && c0 >= origin && c0 <= hi) *(uint32_t*)(M.tb + tb_used - origin + c0) = tbw;
				if (track_good) {
					unsigned long long *gword = M.good + (int64_t)newH * A.GW + (int64_t)g * 4;
#pragma unroll
					for (int i = 0; i < 4; ++i) {
						const unsigned long long m = __ballot((gbits >> i) & 1u);
						if (lane == 0) gword[i] = m;
					}
				}
				// H of the outer columns: read again no sooner than min-lag penalties from now
				if (lane == 63) st_gran(granule(newH, r, 0, 2), hv[3], s_new);
				if (lane == 0) st_gran(granule(newH, r, 1, 2), hv[0], s_new);
				if ((uint32_t)(cfin - cb) < (uint32_t)kChunk && cfin >= lo && cfin <= hi) { // this chunk holds the end diagonal
					const unsigned long long fm = __ballot(fin);
					int32_t val = 0;
					if (fm && SEG) { // the end cell's provenance does not fit the flag word: its own word, visible before the flag is
						const int32_t prov = __builtin_amdgcn_readlane(done_info, (int32_t)__builtin_ctzll(fm));
						if (lane == 0) st_ag(&gflags[16], prov);
						__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
						asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
						val = 1;
					} else if (fm) val = 1 | __builtin_amdgcn_readlane(done_info, (int32_t)__builtin_ctzll(fm)) << 1;
					if (lane < kFlagCopies) st_ag(flag_entry(s_new, lane) + 2, s_new << 4 | val);
				}
				if (lag_one) {
					asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
					prefetch(k, nextH, phi, gl_next);
				}

			} else {
				prefetch(k, nextH, phi, gl_next);
#pragma unroll
				for (int i = 0; i < 4; ++i) ne1[i] = nf1[i] = ne2[i] = nf2[i] = kNegInf;
			}
			if (SEG) {
#pragma unroll
				for (int i = 0; i < 4; ++i) {
#pragma unroll
					for (int a = E1 - 1; a > 0; --a) se1h[a][k][i] = se1h[a - 1][k][i], sf1h[a][k][i] = sf1h[a - 1][k][i];
#pragma unroll
					for (int a = E2 - 1; a > 0; --a) se2h[a][k][i] = se2h[a - 1][k][i], sf2h[a][k][i] = sf2h[a - 1][k][i];
					se1h[0][k][i] = active ? sne1s[i] : kNegInf, sf1h[0][k][i] = active ? snf1s[i] : kNegInf;
					se2h[0][k][i] = active ? sne2s[i] : kNegInf, sf2h[0][k][i] = active ? snf2s[i] : kNegInf;
				}
			}

#pragma unroll
			for (int i = 0; i < 4; ++i) {
#pragma unroll
				for (int a = E1 - 1; a > 0; --a) e1h[a][k][i] = e1h[a - 1][k][i], f1h[a][k][i] = f1h[a - 1][k][i];
#pragma unroll
				for (int a = E2 - 1; a > 0; --a) e2h[a][k][i] = e2h[a - 1][k][i], f2h[a][k][i] = f2h[a - 1][k][i];
				e1h[0][k][i] = ne1[i], f1h[0][k][i] = nf1[i], e2h[0][k][i] = ne2[i], f2h[0][k][i] = nf2[i];
			}
		}

		// Late stores of a chunk (H row, traceback dword, the two H edge words) are not read by anybody for at least
		// min-lag - 1 penalties; when every lag is >= 3 they may stay in flight across this barrier.  They are the youngest
		// operations of the wave, except that an idle second chunk still issues its 5 dummy prefetch loads
		// after them.  (A smaller count than the truth only waits for more.)
		int32_t vm_keep = 0;
		if (relaxed_stores && !track_good && !SEG) {
			if (act1) vm_keep = 3 + (TB ? 1 : 0);
			else if (act0) vm_keep = 8 + (TB ? 1 : 0);
		}

#ifdef MWF_BAND_TIMING
		const unsigned long long t_c = __builtin_readcyclecounter();
#endif
		// ---- end of the penalty.  No grid barrier: neighbours synchronise through the granules above; what every workgroup
		// needs before it can go on is the fate of the two edge columns (the next window) and of the end cell, which their
		// owners publish into the flag ring.  Thread 0 waits for them while the other waves wait at the workgroup barrier.
		// Timestamps (profiles/handoff_probe.py, C4-like pair): the period is 5.2 us; an edge flag is seen 0.9 us after its
		// store, the end-cell flag (known only after the alignment path's match extension) 1.5 us later — but reading that one
		// a penalty late leaves the period at 5.2 us: it is every workgroup's own chunk work (3.7 us from flags seen to chunks
		// done) plus one poll round trip after it.  Measured and rejected: a chunk-less wave that polls from the start of the
		// penalty so that the round trip overlaps the chunk work (1.43 -> 1.50 s, with and without the late end-cell flag).
		switch (vm_keep) { // own H rows are read back min-lag - 1 penalties from now: keep at most this penalty's late stores in flight
		case 3:  asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
		case 4:  asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
		case 8:  asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
		case 9:  asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
		default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
		}
		if (tid == 0) {
			(void)__hip_atomic_fetch_add(arrived, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			const int32_t *fr = flag_entry(s_new, lb & (kFlagCopies - 1));
			const bool want_fin = cfin >= lo && cfin <= hi;
			int32_t w0 = 0, w1 = 0, w2 = 0, ok = 1;
			// each look is ONE 16-byte agent-scope load of this workgroup's copy of the flag words
			for (unsigned spins = 0;; ++spins) {
#ifdef MWF_BAND_TIMING
				t_acc[3] += 1;
#endif
				int4 w;
				asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(w) : "v"(fr) : "memory");
				w0 = w.x, w1 = w.y, w2 = want_fin ? w.z : s_new << 4;
#ifdef MWF_BAND_TIMING
				if (lb == 5 && A.dbg && s_new < A.dbg_cap / 4) { // profiles/handoff_probe.py
					unsigned long long *tt = (unsigned long long*)A.dbg + 4 * s_new;
					if (spins == 0) tt[1] = __builtin_amdgcn_s_memrealtime();                          // workgroup 5 starts looking
					if ((w0 >> 4) == s_new && tt[2] == 0) tt[2] = __builtin_amdgcn_s_memrealtime();     // ... sees the low edge flag
					if ((w0 >> 4) == s_new && (w1 >> 4) == s_new && (w2 >> 4) == s_new) tt[3] = __builtin_amdgcn_s_memrealtime(); // ... sees all
				}
#endif
				if ((w0 >> 4) == s_new && (w1 >> 4) == s_new && (w2 >> 4) == s_new) break;
				if (spins > A.coop_spin_limit || ((spins & 255u) == 255u && ld_ag(&gflags[15]))) { ok = 0; break; }
				__builtin_amdgcn_s_sleep(1);
			}
			// drift bound: the flag ring holds kFlagRing penalties, so nobody may run more than that ahead of the slowest workgroup
			if (ok && s_new >= kDriftCheck && (s_new & (kDriftCheck - 1)) == 0) {
				const unsigned long long want = (unsigned long long)G * (unsigned long long)(s_new - kDriftCheck);
				for (unsigned spins = 0; __hip_atomic_load(arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want; ++spins) {
					if (spins > A.coop_spin_limit || ((spins & 255u) == 255u && ld_ag(&gflags[15]))) { ok = 0; break; }
					__builtin_amdgcn_s_sleep(2);
				}
			}
			if (!ok) st_ag(&gflags[15], 1); // release everybody else at once
			sh.flags[npar][0] = (w0 & 1) | (w1 & 1) << 1 | (w2 & 15) << 2;
			sh.flags[npar][1] = ok;
		}
		__syncthreads();
#ifdef MWF_BAND_TIMING
		{
			const unsigned long long t_d = __builtin_readcyclecounter();
			t_acc[0] += t_b - t_a, t_acc[1] += t_c - t_b, t_acc[2] += t_d - t_c;
			t_steps += 1, t_active += (unsigned long long)(act0 ? 1 : 0) + (act1 ? 1 : 0);
		}
#endif

		// ---- bookkeeping, identical on every thread of every workgroup
		const int32_t fbits = uni(sh.flags[npar][0]);
		if (uni(sh.flags[npar][1]) == 0 || uni(sh.red[0]) != 0) { R.status = ST_INTERNAL; break; }
		if (fbits & 1) wf_lo = lo;
		if (fbits & 2) wf_hi = hi;
		const int32_t done = (fbits >> 2) & 1, payload = (fbits >> 3) & 7;
		s = s_new, curH = newH, par = npar, gl = gl_next;
		if (TB) tb_used += row_bytes;
		if ((s & 0xff) == 0) { // shrink (miniwfa.c:144-171): the good bits were written with ordinary stores by every CU
			if (lead) st_ag(&gflags[13], 0x7fffffff), st_ag(&gflags[14], -1);
			if (!grid_sync<true>(A, sync, (unsigned)lb, sh, epoch, G, nullptr, cum, 0)) { R.status = ST_INTERNAL; break; }
			const int32_t gfirst = wf_lo >> 8, n_words = ((wf_hi >> 8) - gfirst + 1) * 4;
			int32_t mylo = 0x7fffffff, myhi = -1;
			for (int32_t q = lb * kT + tid; q < n_words; q += G * kT) {
				const int32_t gg = gfirst + (q >> 2), kq = q & 3, base = gg * kChunk;
				unsigned long long m = 0;
				for (int32_t j = 0; j < nH; ++j)
					if (sh.rng_lo[j] <= sh.rng_hi[j] && sh.rng_lo[j] <= base + kChunk - 1 && sh.rng_hi[j] >= base) m |= M.good[(int64_t)j * A.GW + (int64_t)gg * 4 + kq];
				m &= lane_mask(base, kq, wf_lo, wf_hi);
				if (m) {
					mylo = min(mylo, base + 4 * (int32_t)__builtin_ctzll(m) + kq);
					myhi = max(myhi, base + 4 * (63 - (int32_t)__builtin_clzll(m)) + kq);
				}
			}
			if (myhi >= 0) {
				__hip_atomic_fetch_min(&gflags[13], mylo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				__hip_atomic_fetch_max(&gflags[14], myhi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
			if (!grid_sync<false>(A, sync, (unsigned)lb, sh, epoch, G, nullptr, cum, 0)) { R.status = ST_INTERNAL; break; }
			const int32_t glo = uni(ld_ag(&gflags[13])), ghi = uni(ld_ag(&gflags[14]));
			if (ghi < 0) { R.status = ST_INTERNAL; break; }
			wf_lo = glo, wf_hi = ghi;
		}
		cells += hi - lo + 1;
		// the low-memory first pass has no stop rules and is not counted in n_iter (miniwfa.c:569-589)
		if (A.coop_pass != 1 && A.coop_pass != 3 && ((A.max_iter > 0 && cells > A.max_iter) || (A.max_s > 0 && s > A.max_s))) {
			R.status = ST_STOPPED;
			break;
		}
		if (done) {
			R.info = payload;
			if (SEG) {
				__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
				R.info = uni(ld_ag(&gflags[16]));
			}
			break;
		}
	}
	R.n_snap = n_snap;
#ifdef MWF_BAND_TIMING
	if (lane == 0 && ((tid >> 6) < 2 || (tid >> 6) == 7) && (blockIdx.x == 0 || blockIdx.x == 72 || blockIdx.x == 73 || blockIdx.x == 74 || blockIdx.x == 200))
		printf("wg %3d wave %d steps %llu active-slots %llu | per step: header %.0f  slots %.0f  drain+flags+barrier %.0f cycles; granule wait per active slot %.0f (watcher: polls per step %.2f)\n", (int)blockIdx.x, tid >> 6,
		       t_steps, t_active, (double)t_acc[0] / t_steps, (double)t_acc[1] / t_steps, (double)t_acc[2] / t_steps, (double)t_acc[3] / (t_active ? t_active : 1), (double)t_acc[3] / t_steps);
#endif
	R.s = s, R.cells = cells;
	return R;
}

// which pair a group of workgroups works on, and where its pass state lives
__device__ __forceinline__ int32_t group_pair(const BatchArgs &A, int32_t grp) { return A.coop_pair_ids ? A.coop_pair_ids[grp] : A.coop_pair; }
__device__ __forceinline__ int32_t *group_state(const BatchArgs &A, int32_t grp) { return (int32_t*)((char*)A.coop_flags + (int64_t)grp * A.coop_misc_stride + 2048); }

// The first pass of the true low-memory mode: a kernel of its own, so that the other passes keep their register allocation.
template <int E1, int E2>
__global__ __launch_bounds__(kT) void wfa_coop_seg_kernel(const BatchArgs A)
{
	__shared__ Shared sh;
	const int32_t G = A.coop_group_size, grp = (int32_t)blockIdx.x / G, lb = (int32_t)blockIdx.x % G;
	PairMem M;
	pair_mem(A, grp, group_pair(A, grp), M);
	const PassResult R = coop_pass<E1, E2, false, true>(A, M, sh, 0, grp, lb, G);
	if (lb == 0 && threadIdx.x == 0) {
		int32_t *st = group_state(A, grp);
		st[0] = R.status, st[1] = R.s, st[2] = R.info;
		st[4] = (int32_t)(R.cells & 0xffffffff), st[5] = (int32_t)(R.cells >> 32);
		st[6] = R.n_snap;
	}
}

// Checkpoints of the true low-memory mode: chase the provenance of the end cell back through the snapshots (reference
// wf_traceback_seg, miniwfa.c:528-549).  An index decodes to (slice id, column) with the snapshot's Wspan and slo; the
// slice id gives the penalty: H ring slot j held penalty S - ((curH - j) mod nH), an E/F register of age a penalty S - a.
__global__ void coop_trace_kernel(const BatchArgs A)
{
	if (threadIdx.x != 0) return;
	const int32_t grp = (int32_t)blockIdx.x; // one block per pair
	int32_t *st = group_state(A, grp);
	st[3] = 0;
	if (st[0] != ST_OK) return;
	const Penalty &P = A.pen;
	PairMem M;
	pair_mem(A, grp, group_pair(A, grp), M);
	const int32_t n_snap = st[6];
	if (n_snap > A.seg_slot) { st[0] = ST_SNAP_OVERFLOW; return; }
	int32_t last = st[2];
	for (int32_t j = n_snap - 1; j >= 0; --j) {
		const int32_t *meta = M.snap_meta + (int64_t)j * 8;
		const int64_t base = (int64_t)(uint32_t)meta[0] | (int64_t)meta[1] << 32;
		const int32_t S = meta[2], curH = meta[3], slo = meta[4], Wspan = meta[5];
		if (last < 0 || Wspan <= 0) { st[0] = ST_INTERNAL; return; }
		const int32_t id = last / Wspan, col = slo + last % Wspan;
		int32_t age;
		if (id < P.nH) { age = curH - id; if (age < 0) age += P.nH; }
		else {
			const int32_t q = id - P.nH;
			if (q < P.e1) age = q;
			else if (q < 2 * P.e1) age = q - P.e1;
			else if (q < 2 * P.e1 + P.e2) age = q - 2 * P.e1;
			else if (q < 2 * P.e1 + 2 * P.e2) age = q - 2 * P.e1 - P.e2;
			else { st[0] = ST_INTERNAL; return; }
		}
		M.seg[2 * j] = S - age, M.seg[2 * j + 1] = col;
		last = M.snap[base + last];
	}
	if (last != -1) { st[0] = ST_INTERNAL; return; } // the chain must end at the origin (reference asserts, miniwfa.c:542,547)
	st[3] = n_snap;
}

template <int E1, int E2>
int launch_pass(const BatchArgs &a, int grid, hipStream_t st)
{
	if (a.coop_pass != 3) return -1; // every other pass runs on the systolic kernel (mwf_sys.hip)
	hipLaunchKernelGGL((wfa_coop_seg_kernel<E1, E2>), dim3(grid), dim3(kT), 0, st, a);
	return hipGetLastError() == hipSuccess ? 0 : -2;
}

} // namespace

bool coop_supported(const Penalty &p)
{
	return ((p.e1 == 2 && p.e2 == 1) || (p.e1 == 2 && p.e2 == 2) || (p.e1 == 1 && p.e2 == 1)) && p.nH <= kMaxRing; // (per-slot window history in LDS)
}

int64_t coop_chunk_slots(int grid) { return (int64_t)grid * kNW * kK; }

int coop_max_grid(bool)
{
	int dev = 0, n_cu = 0, per = 0;
	if (hipGetDevice(&dev) != hipSuccess) return 0;
	if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
	if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, wfa_coop_seg_kernel<2, 1>, kT, 0) != hipSuccess || per < 1) return 0;
	return n_cu; // one workgroup per CU: every one of them is resident, which the grid barrier relies on
}

int launch_coop_pass(const BatchArgs &a, int grid, void *stream)
{
	if (a.pen.e1 == 2 && a.pen.e2 == 1) return launch_pass<2, 1>(a, grid, (hipStream_t)stream);
	if (a.pen.e1 == 2 && a.pen.e2 == 2) return launch_pass<2, 2>(a, grid, (hipStream_t)stream);
	if (a.pen.e1 == 1 && a.pen.e2 == 1) return launch_pass<1, 1>(a, grid, (hipStream_t)stream);
	return -1;
}

int launch_coop_trace(const BatchArgs &a, void *stream)
{
	hipLaunchKernelGGL(coop_trace_kernel, dim3(a.coop_groups > 0 ? a.coop_groups : 1), dim3(64), 0, (hipStream_t)stream, a);
	return hipGetLastError() == hipSuccess ? 0 : -2;
}

} // namespace mwf
