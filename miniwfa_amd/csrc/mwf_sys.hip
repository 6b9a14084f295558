// mwf_sys.hip — one sequence pair across the whole device, systolic form (BASELINE configs 2 and 4: a 150 kb pair, a 5 Mb pair).
//
// A single pair is a strictly sequential chain of penalties (reference miniwfa.c:397-426); the only parallelism is across the
// diagonals of one wavefront, and the only dependence of lag 1 is E2/F2 of the neighbouring diagonal.  Rounds 1-2 (mwf_coop.hip, removed) exchanged
// the outer columns of every 256-column chunk between neighbouring waves at EVERY penalty and polled the fate of the two edge
// columns at every penalty: 5-6 us per penalty, whatever the window, because a penalty costs two dependent cross-CU round trips.
// Here the exchange happens once per block of P penalties (P = 8):
//   * a chunk slot computes 256 columns but owns only the inner 256 - 2P; the P columns on either side are a halo that
//     duplicates the neighbours' outer columns.  After the hand-off every column of the slot is exact; each penalty computed
//     without one loses one column per side (a cell reads columns c-1, c, c+1 of older wavefronts), so after P penalties exactly
//     the owned columns are still exact — nothing is masked, the garbage in the halo simply never reaches an owned column;
//   * every slot has a PRIVATE ring of H rows ([nH][256], halo included) and keeps E/F in registers: between hand-offs a wave
//     talks to nobody — no granule waits, no flag polls, no workgroup barrier; waves of one workgroup drift freely;
//   * at the end of a block a slot publishes what its two neighbours' halos must become — H of its outer P owned columns for
//     the block's P penalties (stored as it goes), their E/F registers, and its view of the window edges — with write-through
//     stores, then a progress word; it then waits for both neighbours' progress words and refreshes its own halo;
//   * the window (reference wf_lo/wf_hi, grown by the liveness of the edge cells, miniwfa.c:417-418, :325-326) is tracked by
//     every slot for itself: the slot that OWNS an edge column computes its liveness exactly and logs the new edge; a slot whose
//     valid columns do not contain the edge cannot be affected by it within the block; at the hand-off the exact chain of the
//     block's edges is re-derived from the owners' published values (see refresh());
//   * everything global happens once per EPOCH of 256 penalties, where the reference shrinks the band (miniwfa.c:144-171) and a
//     device-wide barrier is needed anyway: which slots take part in the next epoch (the window can grow by at most one column
//     per penalty and side), the end cell / stop rules (acted upon at the epoch's end: at most 255 surplus penalties), n_iter
//     (from the edge log), the traceback layout.
// Traceback bytes are laid out per epoch and slot (256 bytes per slot and penalty, dev::tb_byte); the second pass of the
// low-memory mode collapses the window at its checkpoints exactly as the reference does (miniwfa.c:413-416) — every slot
// knows the checkpoints in advance.  The provenance pass of the two-pass low-memory mode (SEG, round 5) runs here as well: shadow
// registers, a shadow H ring and shadow halves of the hand-off boxes; snapshots need nothing global (see sys_pass).
// Results are bit-identical to the other kernels (tests/test_gpu_parity.py, tests/test_long_pairs.py).
#include "mwf_device.h"

namespace mwf {

using namespace dev;

namespace {

constexpr int kT = 512;          // threads per workgroup: 8 waves, up to 256 VGPRs each
constexpr int kNW = kT / 64;
constexpr int kK = 2;            // chunk slots per wave
constexpr int kEpoch = 256;      // penalties between two band shrinks (reference miniwfa.c:429)
constexpr int kMaxP = 16;

__device__ __forceinline__ int32_t from_left(int32_t v, int32_t fill) { return __builtin_amdgcn_update_dpp(fill, v, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ int32_t from_right(int32_t v, int32_t fill) { return __builtin_amdgcn_update_dpp(fill, v, 0x130, 0xf, 0xf, false); }

__device__ __forceinline__ int32_t ld_ag(const int32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_ag(int32_t *p, int32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// hand-off payload: naturally aligned 8-byte words, write-through stores and L2-served loads (sc1) on both sides
typedef unsigned long long u64;
__device__ __forceinline__ u64 ld2_ag(const u64 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st2_ag(u64 *p, int32_t a, int32_t b) { __hip_atomic_store(p, (u64)(uint32_t)a | (u64)(uint32_t)b << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int32_t lo32(u64 v) { return (int32_t)(uint32_t)v; }
__device__ __forceinline__ int32_t hi32(u64 v) { return (int32_t)(uint32_t)(v >> 32); }
template <int C>
__device__ __forceinline__ void ld_box(const int32_t *p, int32_t (&v)[C])
{
	if constexpr (C == 4) { const u64 a = ld2_ag((const u64*)p), b = ld2_ag((const u64*)p + 1); v[0] = lo32(a), v[1] = hi32(a), v[2] = lo32(b), v[3] = hi32(b); }
	else if constexpr (C == 2) { const u64 a = ld2_ag((const u64*)p); v[0] = lo32(a), v[1] = hi32(a); }
	else v[0] = ld_ag(p);
}
template <int C>
__device__ __forceinline__ void st_box(int32_t *p, const int32_t (&v)[C])
{
	if constexpr (C == 4) st2_ag((u64*)p, v[0], v[1]), st2_ag((u64*)p + 1, v[2], v[3]);
	else if constexpr (C == 2) st2_ag((u64*)p, v[0], v[1]);
	else st_ag(p, v[0]);
}

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

// C = columns per lane (4, 2 or 1): a slot computes 64 C columns.  The fewer, the fewer instructions a wave issues per penalty —
// which is what a chain of penalties on few, narrow chunks waits for — and the more slots (and halo) a window of a given width costs.
template <int C>
__device__ __forceinline__ int32_t pickc(int32_t i, const int32_t (&a)[C])
{
	if constexpr (C == 4) return pick4(i, a[0], a[1], a[2], a[3]);
	else if constexpr (C == 2) return i == 0 ? a[0] : a[1];
	else return a[0];
}
template <int C>
__device__ __forceinline__ void ld_cols(const int32_t *p, int32_t (&v)[C])
{
	if constexpr (C == 4) { const int4 x = *(const int4*)p; v[0] = x.x, v[1] = x.y, v[2] = x.z, v[3] = x.w; }
	else if constexpr (C == 2) { const int2 x = *(const int2*)p; v[0] = x.x, v[1] = x.y; }
	else v[0] = *p;
}
template <int C>
__device__ __forceinline__ void st_cols(int32_t *p, const int32_t (&v)[C])
{
	if constexpr (C == 4) *(int4*)p = make_int4(v[0], v[1], v[2], v[3]);
	else if constexpr (C == 2) *(int2*)p = make_int2(v[0], v[1]);
	else *p = v[0];
}

// The recurrence with its traceback byte (dev::wf_cell, miniwfa.c:267-278, :289-306), the byte read off the RESULTS so that few
// values are live at once: H is the maximum of m, e1, e2, f1, f2 and the reference's tie-breaking (mismatch, then E1, E2, F1,
// F2) is the first of them that equals it; a gap state was extended iff it differs from what opening it would have given.
template <bool WANT_TB>
__device__ __forceinline__ Cell sys_cell(int32_t hx, int32_t o1m, int32_t g1m, int32_t o2m, int32_t g2m, int32_t o1p, int32_t g1p, int32_t o2p, int32_t g2p)
{
	if (!WANT_TB) return wf_cell<false>(hx, o1m, g1m, o2m, g2m, o1p, g1p, o2p, g2p);
	Cell c;
	c.e1 = max(o1m, g1m);
	c.e2 = max(o2m, g2m);
	c.f1 = max(o1p, g1p) + 1;
	c.f2 = max(o2p, g2p) + 1;
	const int32_t m = hx + 1;
	c.h = max(max(m, max(c.e1, c.e2)), max(c.f1, c.f2));
	const uint32_t z = c.h == m ? 0u : c.h == c.e1 ? 1u : c.h == c.e2 ? 3u : c.h == c.f1 ? 2u : 4u;
	c.tb = z | ((uint32_t)(c.e1 != o1m) << 3) | ((uint32_t)(c.f1 != o1p + 1) << 4) | ((uint32_t)(c.e2 != o2m) << 5) | ((uint32_t)(c.f2 != o2p + 1) << 6);
	return c;
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

// per slot, in LDS (a wave works on one slot at a time; the other one's registers are parked in HBM, see make_resident)
struct SlotVars {
	int32_t g;          // chunk the slot holds in this epoch
	int32_t part;       // takes part in this epoch
	int32_t fresh;      // has just joined: nothing live, registers start dead
	int32_t wl, wh;     // the slot's view of wf_lo / wf_hi
	int32_t fin_seen;
	int32_t cover_bad;  // the latest penalty whose window (this slot's view) does not cover all of the slot's 256 columns
	int32_t pad[1];
};

struct SysLds {
	int32_t word[8];
	int32_t red[2];
	SlotVars sv[kNW * kK];
	int2 hist[kNW * kK][kMaxRing];     // per slot and H ring row: the slot's view {lo, hi} of that slice's window, clamped to its columns
	int32_t mywl[kNW * kK][kMaxP], mywh[kNW * kK][kMaxP]; // per slot: its view of wf_lo / wf_hi after every penalty of the current block
};

// Device-wide barrier of this pair's group of workgroups, with a release/acquire pair for data written with ordinary
// stores (counters on two levels: workgroups with the same index mod 8 share a word).
__device__ __forceinline__ bool sys_grid_sync(uint32_t spin_limit, unsigned *sync, int32_t *abort_flag, unsigned lb, SysLds &L, unsigned &epoch, unsigned n_wg)
{
	asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
	__syncthreads();
	++epoch;
	if (threadIdx.x == 0) {
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		unsigned long long *const top = (unsigned long long*)sync;                        // [0]
		unsigned *const grp_cnt = sync + 16;                                               // [16 + 8*g]
		unsigned long long *const grp_gen = (unsigned long long*)(sync + 96);              // [96 + 8*g] (8-byte aligned)
		const unsigned grp = lb & 7u, n_grp = n_wg < 8u ? n_wg : 8u;
		const unsigned gsize = (n_wg - grp + 7u) / 8u;
		unsigned spins = 0;
		int32_t ok = 1;
		unsigned long long seen = 0;
		const unsigned old = __hip_atomic_fetch_add(&grp_cnt[8 * grp], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (old + 1 == gsize * epoch) {
			(void)__hip_atomic_fetch_add(top, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			for (;;) {
				seen = __hip_atomic_load(top, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				if ((unsigned)(seen & 0xffffffffu) >= n_grp * epoch) break;
				if (spins < 32) __builtin_amdgcn_s_sleep(2); else __builtin_amdgcn_s_sleep(100); // (idle workgroups wait here for a whole epoch: they must not hammer the fabric the hand-offs travel on)
				if (++spins > spin_limit || ((spins & 255u) == 0 && ld_ag(abort_flag))) { ok = 0; break; }
			}
			// (a leader that gave up publishes a POISONED generation: its members leave the barrier knowing that it did not complete)
			__hip_atomic_store(&grp_gen[4 * grp], (unsigned long long)(ok ? epoch : (epoch | 0x80000000u)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		} else {
			for (;;) {
				seen = __hip_atomic_load(&grp_gen[4 * grp], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				if ((unsigned)(seen & 0xffffffffu) >= epoch) { if (seen & 0x80000000ull) ok = 0; break; }
				if (spins < 32) __builtin_amdgcn_s_sleep(2); else __builtin_amdgcn_s_sleep(100);
				if (++spins > spin_limit || ((spins & 255u) == 0 && ld_ag(abort_flag))) { ok = 0; break; }
			}
		}
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
		if (!ok) st_ag(abort_flag, 1); // whoever gives up first releases everybody else at once
		L.word[3] = ok;
	}
	__syncthreads();
	return uni(L.word[3]) != 0;
}

// Checkpoints the second pass really applies: the reference looks at ONE checkpoint per penalty, `seg[sid].s == s` (miniwfa.c:413),
// so a checkpoint whose penalty is not above its predecessor's (tiny steps: several snapshots can map to one cell) is never
// reached and blocks every later one — the applied ones are the strictly increasing prefix.
__device__ __forceinline__ int32_t seg_effective(const int32_t *seg, int32_t n_seg)
{
	if (n_seg < 2) return n_seg;
	int32_t j = 1;
	while (j < n_seg && seg[2 * j] > seg[2 * (j - 1)]) ++j;
	return uni(j);
}

__device__ __forceinline__ int32_t floordiv(int32_t a, int32_t b) { return a >= 0 ? a / b : -((-a + b - 1) / b); }

// grp / lb / G: this pair's group of workgroups, this workgroup's index in it, the group's size
// SEG (round 5): the first pass of the true low-memory mode (reference mwf_wfa_seg, miniwfa.c:551-601) — no traceback byte is stored; every
// wavefront value carries the index of the cell its predecessor chain went through at the last snapshot (shadow registers, a shadow H ring,
// shadow halves of the hand-off boxes: moved by the choices the traceback byte records, miniwfa.c:495-526), and whenever (s + 1) % step == 0
// every slot flattens the provenance of its owned columns into the snapshot and renumbers all its columns (miniwfa.c:451-474).  Nothing global
// is needed for that: the index of a cell is a function of its array-slice, its column and the epoch's chunk range alone.
template <int E1, int E2, bool TB, int P, bool DEFER, int C, bool SEG = false>
__device__ PassResult sys_pass(const BatchArgs &A, const PairMem &M, SysLds &L, int32_t n_seg, int32_t grp, int32_t lb, int32_t G)
{
	constexpr int kW = 64 * C;                // columns a slot computes
	constexpr int PL = P / C;                 // halo lanes per side
	constexpr int OW = kW - 2 * P;            // columns a slot owns
	constexpr int NEF = 2 * E1 + 2 * E2;      // E/F register arrays per column
	constexpr bool WTB = TB || SEG;           // the recurrence yields the traceback byte
	static_assert(!(TB && SEG), "the first pass of the low-memory mode stores no traceback");
	constexpr int SH_OFF = (P + NEF) * C;     // SEG: a lane's provenance values sit behind its wavefront values
	constexpr int LANE_INTS = (P + NEF) * C * (SEG ? 2 : 1);  // ints one outer lane publishes per block
	constexpr int WIN_OFF = 2 * PL * LANE_INTS; // ints in front of a box's window views
	static_assert(P % C == 0 && (C == 1 || C == 2 || C == 4), "columns per lane");
	constexpr int BOX_INTS = (WIN_OFF + 2 * P + 31) / 32 * 32;
	constexpr int NBLK = kEpoch / P;
	static_assert(P == 4 || P == 8 || P == 16, "block length");
	const int32_t NWt = G * kNW, TC = NWt * kK;
	const int32_t tl = M.tl, ql = M.ql, cmax = tl + ql + 1;
	const int32_t tid = threadIdx.x, lane = tid & 63, wv = uni(tid >> 6), gw = uni(A.sys_spread ? wv * G + lb : lb * kNW + wv);
	const bool lead = lb == 0 && tid == 0;
	char *const misc = (char*)A.coop_flags + (int64_t)grp * A.coop_misc_stride; // this group's flags | barrier words | pass state
	unsigned *const sync = (unsigned*)(misc + 1024);
	int32_t *const gflags = (int32_t*)misc;   // [12]: origin offset; [13..14] and [17..18]: shrink reduction (two parities); [15]: abort; [20..21]: end cell (penalty, last state)
	const int32_t nH = A.pen.nH, lagx = A.pen.x, lag1 = A.pen.oe1, lag2 = A.pen.oe2;
	const uint32_t spin_limit = A.coop_spin_limit;
	const bool lag_one = min(lagx, min(lag1, lag2)) < 2; // a row this penalty writes is read at the next one: no loads ahead of the store
	int32_t *const ring = M.H;                                   // [slot][nH][256]
	int32_t *const box = A.sys_box + (int64_t)grp * A.sys_box_stride;
	u64 *const prog = A.sys_prog + (int64_t)grp * A.sys_prog_stride;
	int32_t *const logL = A.sys_log + (int64_t)grp * A.sys_log_stride, *const logH = logL + A.sys_log_stride / 2;
	int32_t *const park = A.sys_park + (int64_t)grp * A.sys_park_stride;    // [slot][NEF (SEG: 2 NEF)][64 lanes][4]
	int32_t *const sring = SEG ? M.sH : nullptr;                            // [slot][nH][256]: provenance of the H ring
	PassResult R;
	R.status = ST_OK, R.s = 0, R.info = 0, R.n_snap = 0, R.cells = 0;
	unsigned epoch = 0; // the host zeroes the barrier words before every pass

	// E/F wavefronts of the RESIDENT slot (four columns per lane; age 0 is the previous penalty) and its prefetched H rows
	int32_t e1h[E1][C], f1h[E1][C], e2h[E2][C], f2h[E2][C];
	int32_t phx[C], po1[C], po2[C];
	// SEG: their provenance (dead code otherwise); the origin's is -1, where the chain through the snapshots ends (miniwfa.c:119, :542)
	int32_t se1h[E1][C], sf1h[E1][C], se2h[E2][C], sf2h[E2][C];
	int32_t sphx[C], spo1[C], spo2[C];
	int32_t res = -1; // slot of this wave whose E/F are in the registers
	auto set_dead = [&]() {
#pragma unroll
		for (int i = 0; i < C; ++i) {
#pragma unroll
			for (int a = 0; a < E1; ++a) e1h[a][i] = f1h[a][i] = kNegInf, se1h[a][i] = sf1h[a][i] = -1;
#pragma unroll
			for (int a = 0; a < E2; ++a) e2h[a][i] = f2h[a][i] = kNegInf, se2h[a][i] = sf2h[a][i] = -1;
		}
	};
	set_dead();
	// A wave holds up to kK slots but works on one at a time, a whole block of penalties each: the other slot's registers rest
	// in HBM (six 16-byte words per lane; only waves whose two slots are both inside the window ever swap)
	auto park_ptr = [&](int32_t r, int32_t a) -> int32_t* { return park + (((int64_t)r * (SEG ? 2 * NEF : NEF) + a) * 64 + lane) * C; };
	auto make_resident = [&](int32_t k) {
		if (res == k) return;
		if (res >= 0 && uni(L.sv[wv * kK + res].part)) {
			const int32_t r = gw + NWt * res;
			int a = 0;
#pragma unroll
			for (int q = 0; q < E1; ++q, ++a) st_cols<C>(park_ptr(r, a), e1h[q]);
#pragma unroll
			for (int q = 0; q < E1; ++q, ++a) st_cols<C>(park_ptr(r, a), f1h[q]);
#pragma unroll
			for (int q = 0; q < E2; ++q, ++a) st_cols<C>(park_ptr(r, a), e2h[q]);
#pragma unroll
			for (int q = 0; q < E2; ++q, ++a) st_cols<C>(park_ptr(r, a), f2h[q]);
			if (SEG) {
#pragma unroll
				for (int q = 0; q < E1; ++q, ++a) st_cols<C>(park_ptr(r, a), se1h[q]);
#pragma unroll
				for (int q = 0; q < E1; ++q, ++a) st_cols<C>(park_ptr(r, a), sf1h[q]);
#pragma unroll
				for (int q = 0; q < E2; ++q, ++a) st_cols<C>(park_ptr(r, a), se2h[q]);
#pragma unroll
				for (int q = 0; q < E2; ++q, ++a) st_cols<C>(park_ptr(r, a), sf2h[q]);
			}
		}
		res = k;
		if (uni(L.sv[wv * kK + k].fresh)) {
			set_dead();
			if (lane == 0) L.sv[wv * kK + k].fresh = 0;
			return;
		}
		const int32_t r = gw + NWt * k;
		int a = 0;
#pragma unroll
		for (int q = 0; q < E1; ++q, ++a) ld_cols<C>(park_ptr(r, a), e1h[q]);
#pragma unroll
		for (int q = 0; q < E1; ++q, ++a) ld_cols<C>(park_ptr(r, a), f1h[q]);
#pragma unroll
		for (int q = 0; q < E2; ++q, ++a) ld_cols<C>(park_ptr(r, a), e2h[q]);
#pragma unroll
		for (int q = 0; q < E2; ++q, ++a) ld_cols<C>(park_ptr(r, a), f2h[q]);
		if (SEG) {
#pragma unroll
			for (int q = 0; q < E1; ++q, ++a) ld_cols<C>(park_ptr(r, a), se1h[q]);
#pragma unroll
			for (int q = 0; q < E1; ++q, ++a) ld_cols<C>(park_ptr(r, a), sf1h[q]);
#pragma unroll
			for (int q = 0; q < E2; ++q, ++a) ld_cols<C>(park_ptr(r, a), se2h[q]);
#pragma unroll
			for (int q = 0; q < E2; ++q, ++a) ld_cols<C>(park_ptr(r, a), sf2h[q]);
		}
	};

	// ---- penalty 0: origin and its extension (the first wave of workgroup 0 walks it cooperatively)
	if (lb == 0 && tid < 64) {
		const int32_t k0 = lcp_wave(M, 0, 0, min(tl, ql), 0) - 1;
		if (tid == 0) st_ag(&gflags[12], k0), st_ag(&logL[0], tl + 1), st_ag(&logH[0], tl + 1);
	}
	if (lead) st_ag(&gflags[13], 0x7fffffff), st_ag(&gflags[14], -1), st_ag(&gflags[17], 0x7fffffff), st_ag(&gflags[18], -1), st_ag(&gflags[20], 0x7fffffff);
	if (lane < kK) {
		SlotVars z;
		z.g = -1, z.part = 0, z.fresh = 1, z.wl = z.wh = 0, z.fin_seen = 0, z.cover_bad = 0, z.pad[0] = 0;
		L.sv[wv * kK + lane] = z;
	}
	if (!sys_grid_sync(spin_limit, sync, &gflags[15], (unsigned)lb, L, epoch, G)) { R.status = ST_INTERNAL; return R; }
	const int32_t k0 = uni(ld_ag(&gflags[12]));
	if (k0 == tl - 1 && k0 == ql - 1) {
		if (SEG) R.info = -1; // the end cell IS the origin, whose provenance is -1 (miniwfa.c:119): the checkpoint trace expects the chain to end there
		return R;
	}

	int32_t s = 0, wf_lo = tl + 1, wf_hi = tl + 1;
	if (TB) n_seg = seg_effective(M.seg, n_seg);
	int32_t sid = 0, sid_blk = 0;
	int32_t seg_s = TB && n_seg > 0 ? uni(M.seg[0]) : -1, seg_c = TB && n_seg > 0 ? uni(M.seg[1]) : 0; // the next checkpoint
	int64_t cells = 0, tb_used = 0;
	// SEG: the next snapshot is due when `snap_next` penalties are done, i.e. (s + 1) % step == 0 (miniwfa.c:585-586); it will be the snap_idx-th;
	// snap_used ints of the arena lie in front of this epoch's snapshots.  Every wave keeps the same count (idle ones included).
	int32_t snap_next_blk = SEG ? A.step - 1 : 0x7fffffff, snap_idx_blk = 0;
	int64_t snap_used = 0;
	int32_t pgA = 1, pgB = 0; // chunks that took part in the previous epoch
	const int32_t cfin = ql + 1; // the end cell (tl-1, ql-1) lies on diagonal ql-tl, i.e. in this column
	const int32_t gmax = cmax / OW;

	auto row_ptr = [&](int32_t r, int32_t j) -> int32_t* { return ring + (((int64_t)r * nH + j) * kW + C * lane); };
	auto srow_ptr = [&](int32_t r, int32_t j) -> int32_t* { return sring + (((int64_t)r * nH + j) * kW + C * lane); };
	auto prefetch = [&](int32_t r, int32_t slotH) { // the three H rows the penalty that writes ring row slotH reads
		int32_t jx = slotH - lagx; if (jx < 0) jx += nH;
		int32_t j1 = slotH - lag1; if (j1 < 0) j1 += nH;
		int32_t j2 = slotH - lag2; if (j2 < 0) j2 += nH;
		ld_cols<C>(row_ptr(r, jx), phx);
		ld_cols<C>(row_ptr(r, j1), po1);
		ld_cols<C>(row_ptr(r, j2), po2);
		if (SEG) ld_cols<C>(srow_ptr(r, jx), sphx), ld_cols<C>(srow_ptr(r, j1), spo1), ld_cols<C>(srow_ptr(r, j2), spo2);
	};

#ifdef MWF_SYS_TIMING
	unsigned long long t_acc[6] = {0, 0, 0, 0, 0, 0}, t_blocks = 0, t_runs = 0, t_st[4] = {0, 0, 0, 0}, t_ee[3] = {0, 0, 0};
#define MWF_T(x) const unsigned long long x = __builtin_readcyclecounter()
#else
#define MWF_T(x)
#endif
	for (;;) { // ---- one epoch: penalties s+1 .. s+256
		MWF_T(tt_e0);
		const int32_t ep = s >> 8;
		// chunks that take part: their owned columns meet [wf_lo - 257 - P, wf_hi + 257 + P]; then the chunk beyond the outermost
		// one — which does not take part — holds no column the window can reach before the next shrink, and neither do the
		// outermost chunk's own outer P columns (that chunk's halo): a chunk that joins later starts from nothing
		const int32_t gA = max(0, floordiv(wf_lo - (kEpoch + 1 + P), OW)), gB = min(gmax, (wf_hi + kEpoch + 1 + P) / OW);
		const int32_t n_ep = gB - gA + 1;
		if (n_ep > TC - 1) { R.status = ST_BAND_OVERFLOW; break; }
		if (s + 1 > A.rows_slot) { R.status = ST_ROWS_OVERFLOW; break; } // (the log and the epoch table hold rows_slot + 256 penalties)
		const int64_t ep_base = tb_used;
		// SEG: the snapshots that fall into this epoch (penalties done s .. s+255), laid out for its chunk range: one array-slice = n_ep x OW ints
		const int32_t ep_snap_first = snap_next_blk, ep_snap_idx0 = snap_idx_blk;
		const int32_t snap_per = SEG ? n_ep * OW : 0;
		const int64_t snap_total = (int64_t)(nH + NEF) * snap_per;
		int32_t n_snap_ep = 0;
		if (SEG) {
			n_snap_ep = ep_snap_first <= s + kEpoch - 1 ? (s + kEpoch - 1 - ep_snap_first) / A.step + 1 : 0;
			if ((int64_t)(ep_snap_idx0 + n_snap_ep) * 8 > A.snap_meta_slot || snap_used + n_snap_ep * snap_total > A.snap_slot_ints || snap_total > 0x7fffffffLL) { R.status = ST_SNAP_OVERFLOW; break; }
			if (lb == 0 && tid < n_snap_ep) { // (at most 256 per epoch: step >= 1)
				int32_t *meta = M.snap_meta + (int64_t)(ep_snap_idx0 + tid) * 8;
				const int64_t base = snap_used + tid * snap_total;
				const int32_t S = ep_snap_first + tid * A.step;
				meta[0] = (int32_t)(base & 0xffffffff), meta[1] = (int32_t)(base >> 32), meta[2] = S, meta[3] = S % nH, meta[4] = gA, meta[5] = n_ep, meta[6] = OW, meta[7] = 0;
			}
		}
		if (TB) {
			if (tb_used + (int64_t)kEpoch * n_ep * kW > A.tb_slot_bytes) { R.status = ST_TB_OVERFLOW; break; }
			if (lead) {
				int64_t *const ept = A.sys_ep + (int64_t)grp * A.sys_ep_stride;
				ept[2 * ep] = ep_base, ept[2 * ep + 1] = (int64_t)(uint32_t)gA | (int64_t)n_ep << 32;
			}
			tb_used += (int64_t)kEpoch * n_ep * kW;
		}
		const int32_t gbase = gA - gA % TC;
		int32_t n_mine = 0;
#pragma unroll 1
		for (int32_t k = 0; k < kK; ++k) {
			const int32_t r = gw + NWt * k, sl = wv * kK + k;
			int32_t g = gbase + r;
			if (g < gA) g += TC;
			const bool now = g <= gB;
			const bool kept = now && uni(L.sv[sl].part) && uni(L.sv[sl].g) == g;
			n_mine += now ? 1 : 0;
			if (now && !kept) { // joins: no live history
				if (res == k) res = -1; // (whatever the registers hold is some other chunk's)
				const int32_t cb = g * OW - P;
				for (int32_t j = lane; j < nH; j += 64) L.hist[sl][j] = make_int2(cb + kW, cb - 1);
				asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
				// Every row of the slot's ring starts DEAD: a row is only ever written dead outside the window it was computed for (`act` below,
				// the reference's pads, miniwfa.c:96-99), so a read needs no window test — round 5: the three window-history reads and the masks
				// they fed made a penalty of a slot at the window's edge 2.6 x as long as one inside it, and the edge slots set every epoch's pace.
				{
					int32_t dead[C];
#pragma unroll
					for (int i = 0; i < C; ++i) dead[i] = kNegInf;
					for (int32_t j = 0; j < nH; ++j) st_cols<C>(row_ptr(r, j), dead);
				}
				if (s == 0) { // the origin (reference wf_stripe_init, miniwfa.c:103-121)
					const int32_t c0 = tl + 1;
					if (lane == 0) L.hist[sl][0] = make_int2(min(max(c0, cb), cb + kW), max(min(c0, cb + kW - 1), cb - 1));
					if ((uint32_t)(c0 - cb) < (uint32_t)kW && lane == (c0 - cb) / C) {
						ring[((int64_t)r * nH + 0) * kW + (c0 - cb)] = k0;
						if (SEG) sring[((int64_t)r * nH + 0) * kW + (c0 - cb)] = -1;
					}
				}
			}
			if (!now && res == k) res = -1;
			if (lane == 0) {
				SlotVars z;
				z.g = g, z.part = now ? 1 : 0, z.fresh = (now && !kept) ? 1 : (kept ? L.sv[sl].fresh : 1), z.wl = wf_lo, z.wh = wf_hi, z.fin_seen = 0, z.pad[0] = 0;
				z.cover_bad = kept ? L.sv[sl].cover_bad : s; // (joins: no slice so far covers anything)
				L.sv[sl] = z;
			}
		}
		asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
		int32_t curH = s % nH;
		if (n_mine == 0) { // nothing of this wave is near the window: straight to the epoch's end
			s += kEpoch;
			if (TB) while (seg_s >= 0 && seg_s < s) { ++sid; seg_s = sid < n_seg ? uni(M.seg[2 * sid]) : -1, seg_c = sid < n_seg ? uni(M.seg[2 * sid + 1]) : 0; }
		} else
		for (int blk = 0; blk < NBLK; ++blk) {
			const int32_t s0 = s;                        // penalties s0+1 .. s0+P
			const int64_t B = (int64_t)(s0 / P);         // block number since the start of the pass
			const int32_t sid0 = sid, seg_s0 = seg_s, seg_c0 = seg_c;
			// Slots are visited in the order that needs no swap at the start: the resident one first.
			const int32_t kfirst = res >= 0 ? res : 0;
#pragma unroll 1
			for (int32_t kk = 0; kk < kK; ++kk) {
				const int32_t k = kK == 1 ? 0 : (kk == 0 ? kfirst : (kfirst + kk) % kK);
				const int32_t sl = wv * kK + k;
				if (!uni(L.sv[sl].part)) continue;
				const int32_t r = gw + NWt * k, g = uni(L.sv[sl].g), cb = g * OW - P, c0 = cb + C * lane;
				const int32_t oL = cb + P, oR = cb + kW - 1 - P;
				const bool nbl = g - 1 >= gA, nbr = g + 1 <= gB;
				MWF_T(tt_a);
				make_resident(k);
				int32_t wl = uni(L.sv[sl].wl), wh = uni(L.sv[sl].wh), cover_bad = uni(L.sv[sl].cover_bad);
#ifdef MWF_SYS_TIMING
				unsigned long long tt_w = tt_a;
#endif
				// ---- hand-off: the halo becomes what the neighbours computed (nothing to fetch before the first block)
				if (s0 > 0) {
					const int32_t rl = r == 0 ? TC - 1 : r - 1, rr = r + 1 == TC ? 0 : r + 1;
					// a neighbour has something to say if it took part in the block before: within an epoch, if it takes part in this
					// epoch; at an epoch's first block, if it took part in the previous epoch — whether or not it still does (what it
					// left in its outer columns is history this slot may still read); a neighbour that has only just joined has nothing
					const bool hl = blk > 0 ? nbl : (g - 1 >= pgA && g - 1 <= pgB);
					const bool hr = blk > 0 ? nbr : (g + 1 >= pgA && g + 1 <= pgB);
					const bool me = blk > 0 || (g >= pgA && g <= pgB);
					for (unsigned spins = 0;; ++spins) { // wait for the neighbours' block B-1
						bool late = false;
						if (lane == 0 && hl) late = ld2_ag(prog + (int64_t)rl * 8) < (u64)B;
						if (lane == 1 && hr) late = ld2_ag(prog + (int64_t)rr * 8) < (u64)B;
						if (!__ballot(late)) break;
						if (spins > spin_limit || ((spins & 255u) == 255u && uni(ld_ag(&gflags[15])))) {
							if (lane == 0) L.red[0] = 1, st_ag(&gflags[15], 1);
							break;
						}
						__builtin_amdgcn_s_sleep(1);
					}
#ifdef MWF_SYS_TIMING
					tt_w = __builtin_readcyclecounter();
#endif
					const int32_t par = (int32_t)((B - 1) & 1);
					const int32_t *const bl = box + ((int64_t)rl * 2 + par) * BOX_INTS, *const br = box + ((int64_t)rr * 2 + par) * BOX_INTS;
					// halo lanes: H rows of the last P penalties and the E/F registers from the neighbour's outer lanes
					const bool hal = lane < PL, har = lane >= 64 - PL;
					if ((hal && hl) || (har && hr)) {
						// left halo lane l <- left neighbour's right outer lane l (side 1); right halo lane 64-PL+l <- right neighbour's left outer lane l (side 0)
						const int32_t l = hal ? lane : lane - (64 - PL);
						const int32_t *src = (hal ? bl : br) + ((hal ? PL : 0) + l) * LANE_INTS;
						{
							int32_t v[P][C];
#pragma unroll
							for (int t = 0; t < P; ++t) ld_box<C>(src + t * C, v[t]);
							int32_t j = (s0 - P + 1) % nH;
#pragma unroll
							for (int t = 0; t < P; ++t) {
								st_cols<C>(row_ptr(r, j), v[t]);
								j = j + 1 == nH ? 0 : j + 1;
							}
						}
						const int32_t *q = src + P * C;
#pragma unroll
						for (int a = 0; a < E1; ++a, q += C) ld_box<C>(q, e1h[a]);
#pragma unroll
						for (int a = 0; a < E1; ++a, q += C) ld_box<C>(q, f1h[a]);
#pragma unroll
						for (int a = 0; a < E2; ++a, q += C) ld_box<C>(q, e2h[a]);
#pragma unroll
						for (int a = 0; a < E2; ++a, q += C) ld_box<C>(q, f2h[a]);
						if (SEG) { // the provenance of the same values
							int32_t v[P][C];
#pragma unroll
							for (int t = 0; t < P; ++t) ld_box<C>(src + SH_OFF + t * C, v[t]);
							int32_t j = (s0 - P + 1) % nH;
#pragma unroll
							for (int t = 0; t < P; ++t) {
								st_cols<C>(srow_ptr(r, j), v[t]);
								j = j + 1 == nH ? 0 : j + 1;
							}
							q = src + SH_OFF + P * C;
#pragma unroll
							for (int a = 0; a < E1; ++a, q += C) ld_box<C>(q, se1h[a]);
#pragma unroll
							for (int a = 0; a < E1; ++a, q += C) ld_box<C>(q, sf1h[a]);
#pragma unroll
							for (int a = 0; a < E2; ++a, q += C) ld_box<C>(q, se2h[a]);
#pragma unroll
							for (int a = 0; a < E2; ++a, q += C) ld_box<C>(q, sf2h[a]);
						}
					} else if ((hal || har) && me) {
						// no neighbour on that side (it does not take part, or has just joined): nothing there was ever inside the window
						// (its H rows are masked by the window views below: the window never reached those columns)
						set_dead();
					}
					// The window edges of the last block, exactly.  wf_lo after a penalty is decided by the liveness of the cell in the
					// edge column lo (miniwfa.c:325-326), which the slot that OWNS that column computed exactly; so: start from this
					// slot's view at the start of that block (exact wherever it matters to this slot, by induction), and for every
					// penalty take the new wf_lo from whoever owned the edge column — this slot, or the neighbour on that side.
					if (me) {
						// lanes 0..P-1: own wl of penalty i, P..2P-1: own wh; the neighbours' from their boxes
						int32_t mine = 0, left = 0, right = 0;
						if (lane < P) mine = L.mywl[sl][lane];
						else if (lane < 2 * P) mine = L.mywh[sl][lane - P];
						if (lane < 2 * P) {
							if (hl) left = ld_ag(bl + WIN_OFF + lane);
							if (hr) right = ld_ag(br + WIN_OFF + lane);
						}
						int32_t j = (s0 - P + 1) % nH;
						const int2 h0 = L.hist[sl][j];
						int32_t lo_i = uni(h0.x), hi_i = uni(h0.y);
						int32_t cwl = 0, cwh = 0;
						// second pass: checkpoints not yet consumed when the last block began (the one that collapsed the window before
						// that block's first penalty is in hist already)
						int32_t cs = sid_blk;
						if (TB && cs < n_seg && uni(M.seg[2 * cs]) == s0 - P) ++cs;
#pragma unroll
						for (int i = 0; i < P; ++i) {
							// the penalty s0-P+1+i had window [lo_i, hi_i] (this slot's view; exact if inside its columns)
							if (lane == 0) L.hist[sl][j] = make_int2(min(max(lo_i, cb), cb + kW), max(min(hi_i, cb + kW - 1), cb - 1));
							if (lo_i > cb || hi_i < cb + kW - 1) cover_bad = max(cover_bad, s0 - P + 1 + i);
							j = j + 1 == nH ? 0 : j + 1;
							const int32_t mwl = __builtin_amdgcn_readlane(mine, i), mwh = __builtin_amdgcn_readlane(mine, P + i);
							const int32_t lwl = __builtin_amdgcn_readlane(left, i), lwh = __builtin_amdgcn_readlane(left, P + i);
							const int32_t rwl = __builtin_amdgcn_readlane(right, i), rwh = __builtin_amdgcn_readlane(right, P + i);
							// the owner's wf_lo is taken verbatim, so the floor (lo == 1 for wf_lo == 1 and 2) never has to be inverted
							if (lo_i >= oL && lo_i <= oR) cwl = mwl;
							else if (lo_i < oL) cwl = hl ? lwl : lo_i + 1;     // (no neighbour: the edge cell there is dead, wf_lo stays)
							else cwl = hr ? rwl : lo_i + 1;
							if (hi_i >= oL && hi_i <= oR) cwh = mwh;
							else if (hi_i > oR) cwh = hr ? rwh : hi_i - 1;
							else cwh = hl ? lwh : hi_i - 1;
							if (TB && cs < n_seg && uni(M.seg[2 * cs]) == s0 - P + 1 + i) cwl = cwh = uni(M.seg[2 * cs + 1]), ++cs; // miniwfa.c:413-416
							lo_i = cwl > 1 ? cwl - 1 : 1, hi_i = cwh < cmax ? cwh + 1 : cmax;
						}
						if (blk > 0) wl = cwl, wh = cwh; // (an epoch's first block starts from the shrunk band, known to everybody)
					}
					asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
				}

				MWF_T(tt_b);
				// ---- P penalties without talking to anybody.  A penalty has two stages: (1) the recurrence, its traceback byte, edge
				// liveness, and the REQUEST of the first eight bases behind every new offset; (2) the match extension proper — count,
				// walk long runs, end-cell test — and the store of the H row.  Nothing of penalty s+1's stage 1 needs stage 2 of
				// penalty s (E/F travel in registers; an extended H is read again no sooner than min-lag penalties later), so with
				// DEFER stage 2 runs one penalty late, behind the next stage 1: the sequence bytes travel while the wave computes.
				int32_t curHk = curH;
				sid = sid0, seg_s = seg_s0, seg_c = seg_c0; // (every slot of the wave walks the same penalties)
				int32_t snap_next = snap_next_blk, snap_idx = snap_idx_blk;
				if (!lag_one) prefetch(r, curHk + 1 == nH ? 0 : curHk + 1);
				const int32_t par = (int32_t)(B & 1);
				const bool own_fin = (uint32_t)(cfin - (cb + P)) < (uint32_t)OW; // this slot owns the end diagonal
				bool fin_seen = uni(L.sv[sl].fin_seen) != 0;
				const unsigned long long owned_lanes = (~0ull >> PL) & (~0ull << PL);
				// A block deep inside the window — every slice its penalties read covers all of the slot's columns, the window's edges
				// are outside them for good (an edge moves by at most one column per penalty), no shrink near, no checkpoint due:
				// nothing of the window bookkeeping is looked at or written per penalty (it is written once, behind the block)
				const bool deep_blk = !lag_one && (wl > 1 ? wl - 1 : 1) <= cb && (wh < cmax ? wh + 1 : cmax) >= cb + kW - 1 && cover_bad <= s0 + 1 - nH &&
				                      blk * P + P <= kEpoch - nH && !(TB && seg_s >= s0 && seg_s < s0 + P);
				// per column: the largest j = k+1 inside the matrix, min(tl, ql - d), and the query's address for j = 0
				int32_t rj[C];
				const uint8_t *const qsd = M.qs + (c0 - 1 - tl); // (column i: + i)
#pragma unroll
				for (int i = 0; i < C; ++i) rj[i] = max(min(tl, ql - (c0 + i - 1 - tl)), 0);
				// what stage 2 needs of a penalty
				int32_t x_hv[C] = {}, x_fshv = -1, x_snew = 0, x_newH = 0, x_t = 0; // (x_fshv — SEG: the provenance of the end cell's column at that penalty, one scalar)
				int32_t rec_wl = 0, rec_wh = 0;
				uint32_t rec_own = 0;
				uint64_t x_t8[C] = {}, x_q8[C] = {};
				uint32_t x_tbw = 0;
#pragma unroll 1
				for (int t = 0; t < P + (DEFER ? 1 : 0); ++t) {
					int32_t c_hv[C], c_fshv = -1, c_snew = 0, c_newH = 0;
					uint64_t c_t8[C], c_q8[C];
					uint32_t c_tbw = 0;
					// what stage 2a leaves for stage 2b
					int32_t nmat[C] = {};
					uint32_t pend = 0;
					int32_t w_cl[4] = {-1, -1, -1, -1}, w_ci[4] = {0, 0, 0, 0}, w_cj[4] = {0, 0, 0, 0}, w_cq[4] = {0, 0, 0, 0}, w_crm[4] = {0, 0, 0, 0};
					uint64_t w_t = 0, w_q = 0;
					int32_t w_left = 0;
					bool w_valid = false;
					auto rj_at = [&](int32_t ii, int32_t src) -> int32_t { return max(min(tl, ql - (cb + C * src + ii - 1 - tl)), 0); };
					auto stage2a = [&]() {
						// count the first probe; a run of >= 8 matches continues (the cells on the alignment path, a few per penalty, all
						// in one chunk — whose wave every other wave ends up waiting for).  Up to four such cells are walked at once:
						// sixteen lanes each, eight bases per lane, i.e. the next 128 bases of every run in ONE round trip to the
						// sequences, requested here and looked at in stage 2b — with DEFER a whole stage 1 later; a run that is longer
						// still goes on with the whole wave (256 bases per trip).
#pragma unroll
						for (int i = 0; i < C; ++i) {
							const int32_t room = rj[i] - (int32_t)min((uint32_t)(x_hv[i] + 1), (uint32_t)rj[i]); // bases left on the diagonal; 0 for dead and phantom offsets
							const uint64_t x = x_t8[i] ^ x_q8[i];
							nmat[i] = min(x ? (int32_t)(__builtin_ctzll(x) >> 3) : 8, room);
							pend |= ((uint32_t)(x == 0) & (uint32_t)(room > 8)) << i;
						}
						unsigned long long owners = __ballot(pend != 0);
						if (owners) { // uniform
#pragma unroll
							for (int gi = 0; gi < 4; ++gi) {
								if (!owners) continue; // uniform
								const int32_t src = (int32_t)__builtin_ctzll(owners);
								owners &= owners - 1;
								const int32_t w = (int32_t)__builtin_ctz((uint32_t)__builtin_amdgcn_readlane((int32_t)pend, src)); // (its other columns, if any: the leftovers of stage 2b)
								const int32_t hh = __builtin_amdgcn_readlane(pickc<C>(w, x_hv), src);
								w_cl[gi] = src, w_ci[gi] = w, w_cj[gi] = hh + 1, w_cq[gi] = cb + C * src + w - 1 - tl + hh + 1;
								w_crm[gi] = rj_at(w, src) - (hh + 1);
							}
							const int32_t gi = lane >> 4, off = 8 + 8 * (lane & 15);
							const int32_t mj = pick4(gi, w_cj[0], w_cj[1], w_cj[2], w_cj[3]), mq = pick4(gi, w_cq[0], w_cq[1], w_cq[2], w_cq[3]);
							const int32_t mrm = pick4(gi, w_crm[0], w_crm[1], w_crm[2], w_crm[3]), mcl = pick4(gi, w_cl[0], w_cl[1], w_cl[2], w_cl[3]);
							w_valid = mcl >= 0 && off < mrm;
							w_left = mrm - off;
							if (w_valid) w_t = ld8(M.ts + mj + off), w_q = ld8(M.qs + mq + off);
						}
					};
					MWF_T(ts_0);
					if (DEFER && t > 0) stage2a();
#ifdef MWF_SYS_TIMING
					if (DEFER && t > 0 && __ballot(nmat[0] == 0x7fffffff) == 0) {} // (forces the wait for the probe words here)
#endif
					MWF_T(ts_1);
					if (!DEFER || t < P) {
					const int32_t sc = s0 + t; // penalties done so far
					if (SEG && sc == snap_next) {
						// ---- snapshot (reference wf_snapshot1, miniwfa.c:451-474): flatten the provenance of this slot's owned columns, renumber every
						// column of the slot (halo included: a cell's index is a function of its array-slice and column) and what the slot has already
						// published of this block's rows.  Index = (slice * n_ep + owner chunk - gA) * OW + column - owner chunk * OW; slices: the H ring
						// by age (0 = the penalty just done), then E1, F1, E2, F2 by age.
						int32_t *const x = M.snap + (snap_used + (int64_t)(snap_idx - ep_snap_idx0) * snap_total);
						const int32_t gc = lane < PL ? g - 1 : lane >= 64 - PL ? g + 1 : g;
						const int32_t rel = (gc - gA) * OW + (c0 - gc * OW);
						const bool own = lane >= PL && lane < 64 - PL;
						const bool ol = lane >= PL && lane < 2 * PL, orr = lane >= 64 - 2 * PL && lane < 64 - PL;
						int32_t *const bxl = box + ((int64_t)r * 2 + par) * BOX_INTS + ((orr ? PL : 0) + (ol ? lane - PL : lane - (64 - 2 * PL))) * LANE_INTS + SH_OFF;
						for (int32_t j = 0; j < nH; ++j) {
							int32_t age = curHk - j;
							if (age < 0) age += nH;
							if (age > sc) continue; // uniform: that ring row has not been written yet
							int32_t v[C];
							ld_cols<C>(srow_ptr(r, j), v);
							const int32_t f0 = age * snap_per + rel;
							if (own) st_cols<C>(x + f0, v);
#pragma unroll
							for (int i = 0; i < C; ++i) v[i] = f0 + i;
							st_cols<C>(srow_ptr(r, j), v);
							if (age < t && (ol || orr)) st_box<C>(bxl + C * (t - 1 - age), v); // (entry t' of the box holds the penalty s0 + 1 + t')
						}
						auto flat = [&](int32_t (&reg)[C], int32_t code) {
							const int32_t f0 = code * snap_per + rel;
							if (own) st_cols<C>(x + f0, reg);
#pragma unroll
							for (int i = 0; i < C; ++i) reg[i] = f0 + i;
						};
#pragma unroll
						for (int a = 0; a < E1; ++a) flat(se1h[a], nH + a), flat(sf1h[a], nH + E1 + a);
#pragma unroll
						for (int a = 0; a < E2; ++a) flat(se2h[a], nH + 2 * E1 + a), flat(sf2h[a], nH + 2 * E1 + E2 + a);
						asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
						snap_next += A.step, ++snap_idx;
						// what was requested for this penalty before the renumbering is stale: request it again
						if (!lag_one) prefetch(r, curHk + 1 == nH ? 0 : curHk + 1);
					}
					if (TB && !deep_blk && seg_s == sc) { // checkpoint reset of the second pass (miniwfa.c:413-416): every slot knows the checkpoints
						wl = wh = seg_c;
						++sid;
						seg_s = sid < n_seg ? uni(M.seg[2 * sid]) : -1, seg_c = sid < n_seg ? uni(M.seg[2 * sid + 1]) : 0;
					}
					const int32_t s_new = sc + 1;
					const int32_t newH = curHk + 1 == nH ? 0 : curHk + 1;
					const int32_t nextH = newH + 1 == nH ? 0 : newH + 1;
					const bool track_good = !deep_blk && (((256 - (s_new & 255)) & 255) < nH);
					const int32_t lo = wl > 1 ? wl - 1 : 1;       // miniwfa.c:417-418, on this slot's view
					const int32_t hi = wh < cmax ? wh + 1 : cmax;
					bool inner = true;
					if (!deep_blk) {
						if (lag_one) prefetch(r, newH);
						if (lane == 0) L.hist[sl][newH] = make_int2(min(max(lo, cb), cb + kW), max(min(hi, cb + kW - 1), cb - 1)); // (read back at the hand-off: the chain of the block's exact windows starts from it)
						if (lo > cb || hi < cb + kW - 1) cover_bad = s_new;
						// every column of the slot inside the window: nothing to mask (sources need no masks at all: rows are dead outside their windows)
						inner = lo <= cb && hi >= cb + kW - 1;
					}
					int32_t hx[C], o1[C + 2], o2[C + 2];
					int32_t shx[C], so1[C + 2], so2[C + 2]; // SEG: provenance of the same sources
#pragma unroll
					for (int i = 0; i < C; ++i) hx[i] = phx[i], o1[i + 1] = po1[i], o2[i + 1] = po2[i];
					if (SEG) {
#pragma unroll
						for (int i = 0; i < C; ++i) shx[i] = sphx[i], so1[i + 1] = spo1[i], so2[i + 1] = spo2[i];
					}
					// the next penalty's rows: requested at once — they are at least two penalties old (every lag >= 2 here), and a whole
					// penalty's work lies between this request and their use
					if (!lag_one && t + 1 < P) prefetch(r, nextH);
					// the columns next to the slot's 256 are nobody's business: its outermost columns are never exact anyway
					o1[0] = from_left(o1[C], kNegInf), o1[C + 1] = from_right(o1[1], kNegInf);
					o2[0] = from_left(o2[C], kNegInf), o2[C + 1] = from_right(o2[1], kNegInf);
					int32_t g1m[C], g1p[C], g2m[C], g2p[C];
					g1m[0] = from_left(e1h[E1 - 1][C - 1], kNegInf);
					g2m[0] = from_left(e2h[E2 - 1][C - 1], kNegInf);
					g1p[C - 1] = from_right(f1h[E1 - 1][0], kNegInf);
					g2p[C - 1] = from_right(f2h[E2 - 1][0], kNegInf);
#pragma unroll
					for (int i = 1; i < C; ++i) g1m[i] = e1h[E1 - 1][i - 1], g2m[i] = e2h[E2 - 1][i - 1];
#pragma unroll
					for (int i = 0; i < C - 1; ++i) g1p[i] = f1h[E1 - 1][i + 1], g2p[i] = f2h[E2 - 1][i + 1];
					int32_t sg1m[C], sg1p[C], sg2m[C], sg2p[C];
					if (SEG) {
						so1[0] = from_left(so1[C], -1), so1[C + 1] = from_right(so1[1], -1);
						so2[0] = from_left(so2[C], -1), so2[C + 1] = from_right(so2[1], -1);
						sg1m[0] = from_left(se1h[E1 - 1][C - 1], -1);
						sg2m[0] = from_left(se2h[E2 - 1][C - 1], -1);
						sg1p[C - 1] = from_right(sf1h[E1 - 1][0], -1);
						sg2p[C - 1] = from_right(sf2h[E2 - 1][0], -1);
#pragma unroll
						for (int i = 1; i < C; ++i) sg1m[i] = se1h[E1 - 1][i - 1], sg2m[i] = se2h[E2 - 1][i - 1];
#pragma unroll
						for (int i = 0; i < C - 1; ++i) sg1p[i] = sf1h[E1 - 1][i + 1], sg2p[i] = sf2h[E2 - 1][i + 1];
					}
					int32_t sne1[C], snf1[C], sne2[C], snf2[C], shv[C];

					int32_t ne1[C], nf1[C], ne2[C], nf2[C];
					uint32_t tbw = 0, live = 0, gbits = 0;
					if (inner && !track_good) {
						// The common case — a chunk well inside the window, no shrink in sight: every column is computed, no source is
						// masked, no edge column is among the slot's exact columns.  Validity of an offset folds into the probe address:
						// j = k+1 clamped to Rj = min(tl, ql-d) leaves room Rj - j = 0 for dead (NEG_INF + drift: huge as unsigned) and
						// phantom (beyond the matrix) offsets, and both addresses stay inside the sequences' slack.
#pragma unroll
						for (int i = 0; i < C; ++i) {
							const Cell v = sys_cell<WTB>(hx[i], o1[i], g1m[i], o2[i], g2m[i], o1[i + 2], g1p[i], o2[i + 2], g2p[i]);
							ne1[i] = v.e1, nf1[i] = v.f1, ne2[i] = v.e2, nf2[i] = v.f2;
							if (SEG) { // provenance follows the choices the traceback byte records (miniwfa.c:504-523)
								const Cell u = shadow_cell(v.tb, shx[i], so1[i], sg1m[i], so2[i], sg2m[i], so1[i + 2], sg1p[i], so2[i + 2], sg2p[i]);
								sne1[i] = u.e1, snf1[i] = u.f1, sne2[i] = u.e2, snf2[i] = u.f2, shv[i] = u.h;
							}
							const int32_t jc = (int32_t)min((uint32_t)(v.h + 1), (uint32_t)rj[i]);
							c_t8[i] = ld8(M.ts + jc), c_q8[i] = ld8((qsd + jc) + i);
							c_hv[i] = v.h;
							tbw |= v.tb << (8 * i);
						}
					} else
#pragma unroll
					for (int i = 0; i < C; ++i) {
						const int32_t c = c0 + i, d = c - 1 - tl;
						const uint32_t act = inner ? 1u : (uint32_t)((c >= lo) & (c <= hi));
						const Cell v = sys_cell<WTB>(hx[i], o1[i], g1m[i], o2[i], g2m[i], o1[i + 2], g1p[i], o2[i + 2], g2p[i]);
						ne1[i] = act ? v.e1 : kNegInf, nf1[i] = act ? v.f1 : kNegInf;
						ne2[i] = act ? v.e2 : kNegInf, nf2[i] = act ? v.f2 : kNegInf;
						if (SEG) { // (the provenance of a dead cell is never followed: no masks)
							const Cell u = shadow_cell(v.tb, shx[i], so1[i], sg1m[i], so2[i], sg2m[i], so1[i + 2], sg1p[i], so2[i + 2], sg2p[i]);
							sne1[i] = u.e1, snf1[i] = u.f1, sne2[i] = u.e2, snf2[i] = u.f2, shv[i] = u.h;
						}
						const uint32_t inm = act & inm_bit(d, v.h, tl, ql);
						if (track_good)
							gbits |= (act & (inm | inm_bit(d, v.e1, tl, ql) | inm_bit(d, v.f1, tl, ql) | inm_bit(d, v.e2, tl, ql) | inm_bit(d, v.f2, tl, ql))) << i;
						// edge rule (miniwfa.c:325-326): H is the max of the five, so "any live" == "H live"
						const uint32_t lv = act & (uint32_t)(v.h >= -1);
						live |= (lv & (uint32_t)(c == lo)) | ((lv & (uint32_t)(c == hi)) << 1);
						// first probe: eight bases (a random 4-mer matches in one cell of 256, i.e. once per chunk and penalty — and every
						// such cell would send the whole wave on a walk); addresses clamped as above (a cell outside the window holds NEG_INF)
						c_hv[i] = act ? v.h : kNegInf;
						const int32_t jc = (int32_t)min((uint32_t)(c_hv[i] + 1), (uint32_t)rj[i]);
						c_t8[i] = ld8(M.ts + jc), c_q8[i] = ld8((qsd + jc) + i);
						tbw |= v.tb << (8 * i);
					}
					c_tbw = tbw, c_snew = s_new, c_newH = newH;
					if (SEG) { // the provenance of the new H row is final here (the match extension moves offsets, not predecessors)
#pragma unroll
						for (int i = 0; i < C; ++i) (void)shv[i];
						if (own_fin) c_fshv = __builtin_amdgcn_readlane(pickc<C>((cfin - cb) % C, shv), (cfin - cb) / C); // uniform: only the slot that owns the end cell's column keeps it
						st_cols<C>(srow_ptr(r, newH), shv);
						const bool ol = lane >= PL && lane < 2 * PL, orr = lane >= 64 - 2 * PL && lane < 64 - PL;
						if (ol || orr) st_box<C>(box + ((int64_t)r * 2 + par) * BOX_INTS + ((orr ? PL : 0) + (ol ? lane - PL : lane - (64 - 2 * PL))) * LANE_INTS + SH_OFF + C * t, shv);
					}
					if (TB) {
						uint8_t *const tp = M.tb + ep_base + ((int64_t)(s_new - 1 - (ep << 8)) * n_ep + (g - gA)) * kW + C * lane;
						if constexpr (C == 4) *(uint32_t*)tp = tbw;
						else if constexpr (C == 2) *(uint16_t*)tp = (uint16_t)tbw;
						else *tp = (uint8_t)tbw;
					}
					if (track_good) {
						unsigned long long *gword = M.good + ((int64_t)newH * TC + r) * C;
#pragma unroll
						for (int i = 0; i < C; ++i) {
							const unsigned long long m = __ballot((gbits >> i) & 1u) & owned_lanes;
							if (lane == 0) gword[i] = m;
						}
					}
					// the slot's view of the window after this penalty: liveness of an edge cell counts where the cell is exact
					if (!deep_blk) {
						const int32_t vl = cb + 1 + t, vr = cb + kW - 2 - t;
						if (lo >= vl && lo <= vr && __ballot(live & 1u)) wl = lo;
						if (hi >= vl && hi <= vr && __ballot(live & 2u)) wh = hi;
						// lane t keeps the view after penalty t (and whether this slot owns the edge columns, i.e. keeps the log): written behind the block
						const bool mine = lane == t;
						rec_wl = mine ? wl : rec_wl, rec_wh = mine ? wh : rec_wh;
						rec_own = mine ? ((uint32_t)((uint32_t)(lo - (cb + P)) < (uint32_t)OW) | (uint32_t)((uint32_t)(hi - (cb + P)) < (uint32_t)OW) << 1) : rec_own;
					}
#pragma unroll
					for (int i = 0; i < C; ++i) {
#pragma unroll
						for (int a = E1 - 1; a > 0; --a) e1h[a][i] = e1h[a - 1][i], f1h[a][i] = f1h[a - 1][i];
#pragma unroll
						for (int a = E2 - 1; a > 0; --a) e2h[a][i] = e2h[a - 1][i], f2h[a][i] = f2h[a - 1][i];
						e1h[0][i] = ne1[i], f1h[0][i] = nf1[i], e2h[0][i] = ne2[i], f2h[0][i] = nf2[i];
						if (SEG) {
#pragma unroll
							for (int a = E1 - 1; a > 0; --a) se1h[a][i] = se1h[a - 1][i], sf1h[a][i] = sf1h[a - 1][i];
#pragma unroll
							for (int a = E2 - 1; a > 0; --a) se2h[a][i] = se2h[a - 1][i], sf2h[a][i] = sf2h[a - 1][i];
							se1h[0][i] = sne1[i], sf1h[0][i] = snf1[i], se2h[0][i] = sne2[i], sf2h[0][i] = snf2[i];
						}
					}
					curHk = newH;
					}
					MWF_T(ts_2);
					if (!DEFER) {
#pragma unroll
						for (int i = 0; i < C; ++i) x_hv[i] = c_hv[i], x_t8[i] = c_t8[i], x_q8[i] = c_q8[i];
						x_fshv = c_fshv;
						x_tbw = c_tbw, x_snew = c_snew, x_newH = c_newH, x_t = t;
						stage2a();
					}
					// ---- stage 2b, of this penalty or (DEFER) of the one before: resolve the walks, finish the H row
					if (!DEFER || t > 0) {
						uint32_t fin = 0;
						if (w_cl[0] >= 0) { // uniform
							int32_t m8 = 0; // matching bases among this lane's eight (0 beyond the room: stops the scan there)
							if (w_valid) {
								const uint64_t x = w_t ^ w_q;
								m8 = min(x ? (int32_t)(__builtin_ctzll(x) >> 3) : 8, w_left);
							}
							const unsigned long long stop = __ballot(m8 < 8);
#pragma unroll
							for (int g4 = 0; g4 < 4; ++g4) {
								if (w_cl[g4] < 0) continue; // uniform
								const uint32_t sb = (uint32_t)((stop >> (16 * g4)) & 0xffffu);
								int32_t n;
								if (sb) {
									const int32_t first = (int32_t)__builtin_ctz(sb);
									n = min(8 + 8 * first + __builtin_amdgcn_readlane(m8, 16 * g4 + first), w_crm[g4]);
								} else n = lcp_wave(M, w_cj[g4], w_cq[g4], w_crm[g4], 136);
								if (lane == w_cl[g4]) {
#pragma unroll
									for (int i = 0; i < C; ++i) nmat[i] = w_ci[g4] == i ? n : nmat[i];
									pend &= ~(1u << w_ci[g4]);
								}
							}
							// what did not fit the four groups (rare): one owning lane and column at a time, the whole wave on each
							unsigned long long owners = __ballot(pend != 0);
							while (owners) {
								const int32_t src = (int32_t)__builtin_ctzll(owners);
								owners &= owners - 1;
								uint32_t bits = (uint32_t)__builtin_amdgcn_readlane((int32_t)pend, src);
								while (bits) {
									const int32_t ii = (int32_t)__builtin_ctz(bits);
									bits &= bits - 1;
									const int32_t hh = __builtin_amdgcn_readlane(pickc<C>(ii, x_hv), src);
									const int32_t rm = rj_at(ii, src) - (hh + 1);
									const int32_t n = lcp_wave(M, hh + 1, cb + C * src + ii - 1 - tl + hh + 1, rm, 8);
#pragma unroll
									for (int i = 0; i < C; ++i) nmat[i] = (ii == i && lane == src) ? n : nmat[i];
								}
							}
						}
						int32_t done_info = 0, hv[C];
#pragma unroll
						for (int i = 0; i < C; ++i) {
							const int32_t d = c0 + i - 1 - tl;
							const uint32_t in = inm_bit(d, x_hv[i], tl, ql); // (cells outside the window hold NEG_INF)
							const int32_t kk2 = x_hv[i] + nmat[i];
							if (own_fin) {
								const uint32_t f = in & (uint32_t)(c0 + i == cfin) & (uint32_t)(kk2 == tl - 1) & (uint32_t)(d + kk2 == ql - 1);
								fin |= f;
								done_info = f ? (SEG ? x_fshv : (nmat[i] == 0 ? (int32_t)((x_tbw >> (8 * i)) & 7u) : 0)) : done_info; // (SEG: where the chain through the snapshots starts, miniwfa.c:577)
							}
							hv[i] = kk2;
						}
						st_cols<C>(row_ptr(r, x_newH), hv);
						// the outer owned columns, for the neighbours' halos
						{
							const bool ol = lane >= PL && lane < 2 * PL, orr = lane >= 64 - 2 * PL && lane < 64 - PL;
							if (ol || orr) {
								st_box<C>(box + ((int64_t)r * 2 + par) * BOX_INTS + ((orr ? PL : 0) + (ol ? lane - PL : lane - (64 - 2 * PL))) * LANE_INTS + C * x_t, hv);
							}
						}
						if (own_fin && !fin_seen) {
							const unsigned long long fm = __ballot(fin);
							if (fm) {
								fin_seen = true;
								const int32_t info = __builtin_amdgcn_readlane(done_info, (int32_t)__builtin_ctzll(fm));
								if (lane == 0) st_ag(&gflags[21], info), st_ag(&gflags[20], x_snew), L.sv[sl].fin_seen = 1; // read after the epoch's barrier
							}
						}
						if (lag_one) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
					}
					if (DEFER) {
#pragma unroll
						for (int i = 0; i < C; ++i) x_hv[i] = c_hv[i], x_t8[i] = c_t8[i], x_q8[i] = c_q8[i];
						x_fshv = c_fshv;
						x_tbw = c_tbw, x_snew = c_snew, x_newH = c_newH, x_t = t;
					}
#ifdef MWF_SYS_TIMING
					if (deep_blk) {
						const unsigned long long ts_3 = __builtin_readcyclecounter();
						t_st[0] += ts_1 - ts_0, t_st[1] += ts_2 - ts_1, t_st[2] += ts_3 - ts_2, t_st[3] += 1;
					}
#endif
				}

				if (deep_blk && lane < P) { // what the penalties of a deep block did not write one by one
					int32_t j = curH + 1 + lane;
					if (j >= nH) j -= nH;
					if (j >= nH) j %= nH;
					L.hist[sl][j] = make_int2(cb, cb + kW - 1);
					L.mywl[sl][lane] = wl, L.mywh[sl][lane] = wh;
				}
				if (!deep_blk && lane < P) { // the views of the block's penalties, and the edge log where this slot owned the edge column
					L.mywl[sl][lane] = rec_wl, L.mywh[sl][lane] = rec_wh;
					if (rec_own & 1u) st_ag(&logL[s0 + 1 + lane], rec_wl);
					if (rec_own & 2u) st_ag(&logH[s0 + 1 + lane], rec_wh);
				}
				if (lane == 0) L.sv[sl].cover_bad = cover_bad;
				MWF_T(tt_c);
				// ---- publish: E/F of the outer owned columns, the window views of the block, then the progress word
				{
					int32_t *const bx = box + ((int64_t)r * 2 + par) * BOX_INTS;
					const bool ol = lane >= PL && lane < 2 * PL, orr = lane >= 64 - 2 * PL && lane < 64 - PL;
					if (ol || orr) {
						int32_t *dst = bx + ((orr ? PL : 0) + (ol ? lane - PL : lane - (64 - 2 * PL))) * LANE_INTS + C * P;
#pragma unroll
						for (int a = 0; a < E1; ++a, dst += C) st_box<C>(dst, e1h[a]);
#pragma unroll
						for (int a = 0; a < E1; ++a, dst += C) st_box<C>(dst, f1h[a]);
#pragma unroll
						for (int a = 0; a < E2; ++a, dst += C) st_box<C>(dst, e2h[a]);
#pragma unroll
						for (int a = 0; a < E2; ++a, dst += C) st_box<C>(dst, f2h[a]);
						if (SEG) {
							dst = bx + ((orr ? PL : 0) + (ol ? lane - PL : lane - (64 - 2 * PL))) * LANE_INTS + SH_OFF + C * P;
#pragma unroll
							for (int a = 0; a < E1; ++a, dst += C) st_box<C>(dst, se1h[a]);
#pragma unroll
							for (int a = 0; a < E1; ++a, dst += C) st_box<C>(dst, sf1h[a]);
#pragma unroll
							for (int a = 0; a < E2; ++a, dst += C) st_box<C>(dst, se2h[a]);
#pragma unroll
							for (int a = 0; a < E2; ++a, dst += C) st_box<C>(dst, sf2h[a]);
						}
					}
					if (lane < 2 * P) st_ag(bx + WIN_OFF + lane, lane < P ? L.mywl[sl][lane] : L.mywh[sl][lane - P]);
					if (lane == 0) L.sv[sl].wl = wl, L.sv[sl].wh = wh;
					asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the payload has left before the progress word does
					if (lane == 0) __hip_atomic_store(prog + (int64_t)r * 8, (u64)(B + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				}
#ifdef MWF_SYS_TIMING
				{
					const unsigned long long tt_d = __builtin_readcyclecounter();
					t_acc[0] += tt_w - tt_a, t_acc[1] += tt_b - tt_w, t_acc[2] += tt_c - tt_b, t_acc[3] += tt_d - tt_c, t_blocks += 1;
				}
#endif
			}
			// (a wave with two slots inside the window: the second one's hand-off waits for neighbours that may be this wave's
			// other slot's neighbours' neighbours — every slot publishes before any slot of the NEXT block waits, so nothing cycles)
			s = s0 + P;
			for (int t = 0; t < P; ++t) curH = curH + 1 == nH ? 0 : curH + 1;
			sid_blk = sid0;
			if (SEG) while (snap_next_blk < s0 + P) snap_next_blk += A.step, ++snap_idx_blk;
		}
		pgA = gA, pgB = gB;
		if (SEG) snap_next_blk = ep_snap_first + n_snap_ep * A.step, snap_idx_blk = ep_snap_idx0 + n_snap_ep, snap_used += n_snap_ep * snap_total;
		MWF_T(tt_e1);

		// ---- end of the epoch: everybody meets; edges from the log, n_iter, stop rules, end cell, shrink
		if (!sys_grid_sync(spin_limit, sync, &gflags[15], (unsigned)lb, L, epoch, G) || uni(L.red[0]) != 0) { R.status = ST_INTERNAL; break; }
		MWF_T(tt_e2);
		{
			// widths of the epoch's 256 slices from the log: lane l looks at penalties s-255+4l .. s-252+4l
			const int32_t sb = s - kEpoch; // penalties sb+1 .. s
			int32_t w4[4];
			int64_t sum = 0;
#pragma unroll
			for (int i = 0; i < 4; ++i) {
				const int32_t sp = sb + 4 * lane + i; // the slice sp+1 grows from the edges after penalty sp
				int32_t a = logL[sp], b = logH[sp];
				if (TB && n_seg > 0) // a checkpoint at penalty sp collapses the window before slice sp+1 (checkpoints are few)
					for (int32_t j = 0; j < n_seg; ++j)
						if (M.seg[2 * j] == sp) a = b = M.seg[2 * j + 1];
				const int32_t lo = a > 1 ? a - 1 : 1, hi = b < cmax ? b + 1 : cmax;
				w4[i] = hi - lo + 1;
				sum += w4[i];
			}
			// inclusive prefix over the lanes
			int64_t pre = sum;
#pragma unroll
			for (int d = 1; d < 64; d <<= 1) {
				const int64_t o = __shfl_up(pre, d, 64);
				if (lane >= d) pre += o;
			}
			const int64_t before = cells + pre - sum; // cells up to and including penalty sb+4*lane
			const bool rules = A.coop_pass != 1 && A.coop_pass != 3; // the low-memory first pass has no stop rules (miniwfa.c:569-589)
			const int64_t max_iter = A.max_iter;
			const int32_t max_s = A.max_s;
			int32_t first_stop = 0x7fffffff;
			int64_t stop_cells = 0, run = before;
#pragma unroll
			for (int i = 0; i < 4; ++i) {
				run += w4[i];
				const int32_t sp1 = sb + 4 * lane + i + 1;
				if (rules && first_stop == 0x7fffffff && ((max_iter > 0 && run > max_iter) || (max_s > 0 && sp1 > max_s))) first_stop = sp1, stop_cells = run; // miniwfa.c:422-425
			}
			const int32_t s_done = uni(ld_ag(&gflags[20])), done_info = uni(ld_ag(&gflags[21]));
			const unsigned long long sm = __ballot(first_stop != 0x7fffffff);
			if (sm) {
				const int32_t sl0 = (int32_t)__builtin_ctzll(sm);
				const int32_t fs = __builtin_amdgcn_readlane(first_stop, sl0);
				if (fs <= s_done) { // the rules are looked at before the end cell of the same slice
					R.status = ST_STOPPED, s = fs;
					cells = ((int64_t)__builtin_amdgcn_readlane((int32_t)(stop_cells >> 32), sl0) << 32) | (uint32_t)__builtin_amdgcn_readlane((int32_t)(stop_cells & 0xffffffff), sl0);
					break;
				}
			}
			if (s_done <= s) {
				// cells up to and including penalty s_done
				const int32_t at = s_done - sb - 1, ln = at >> 2, ii = at & 3;
				int64_t c = before;
#pragma unroll
				for (int i = 0; i < 4; ++i) c += i <= ii ? w4[i] : 0;
				cells = ((int64_t)__builtin_amdgcn_readlane((int32_t)(c >> 32), ln) << 32) | (uint32_t)__builtin_amdgcn_readlane((int32_t)(c & 0xffffffff), ln);
				s = s_done, R.info = done_info;
				break;
			}
			const int64_t tot = before + sum;
			cells = ((int64_t)__builtin_amdgcn_readlane((int32_t)(tot >> 32), 63) << 32) | (uint32_t)__builtin_amdgcn_readlane((int32_t)(tot & 0xffffffff), 63);
		}
		wf_lo = uni(logL[s]), wf_hi = uni(logH[s]);
		MWF_T(tt_e3);
		{ // shrink (reference wf_stripe_shrink, miniwfa.c:144-171) on the good bits of the last nH slices
			const int32_t rd = (ep & 1) ? 17 : 13;            // this epoch's reduction words; the other pair is reset for the next one
			if (lead) st_ag(&gflags[(ep & 1) ? 13 : 17], 0x7fffffff), st_ag(&gflags[(ep & 1) ? 14 : 18], -1);
			const int32_t gfirst = max(gA, wf_lo / OW), glast = min(gB, wf_hi / OW), n_words = (glast - gfirst + 1) * C;
			int32_t mylo = 0x7fffffff, myhi = -1;
			for (int32_t q = lb * kT + tid; q < n_words; q += G * kT) {
				const int32_t gg = gfirst + q / C, kq = q % C, base = gg * OW - P, rr = gg % TC;
				unsigned long long m = 0;
				for (int32_t j = 0; j < nH; ++j) m |= M.good[((int64_t)j * TC + rr) * C + kq];
				// bit l of word kq is column base + C l + kq
				for (; m; m &= m - 1) {
					const int32_t c = base + C * (int32_t)__builtin_ctzll(m) + kq;
					if (c >= wf_lo && c <= wf_hi) { mylo = min(mylo, c); break; }
				}
				for (; m; ) {
					const int32_t hb = 63 - (int32_t)__builtin_clzll(m);
					const int32_t c = base + C * hb + kq;
					if (c >= wf_lo && c <= wf_hi) { myhi = max(myhi, c); break; }
					m &= ~(1ull << hb);
				}
			}
			if (mylo != 0x7fffffff) __hip_atomic_fetch_min(&gflags[rd], mylo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			if (myhi >= 0) __hip_atomic_fetch_max(&gflags[rd + 1], myhi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			if (!sys_grid_sync(spin_limit, sync, &gflags[15], (unsigned)lb, L, epoch, G)) { R.status = ST_INTERNAL; break; }
			const int32_t glo = uni(ld_ag(&gflags[rd])), ghi = uni(ld_ag(&gflags[rd + 1]));
			if (ghi < 0 || glo == 0x7fffffff) { R.status = ST_INTERNAL; break; }
			wf_lo = glo, wf_hi = ghi;
			if (lead) st_ag(&logL[s], glo), st_ag(&logH[s], ghi); // what the next slice grows from
		}
#ifdef MWF_SYS_TIMING
		t_acc[4] += tt_e1 - tt_e0, t_acc[5] += __builtin_readcyclecounter() - tt_e1, t_runs += 1;
		t_ee[0] += tt_e2 - tt_e1, t_ee[1] += tt_e3 - tt_e2, t_ee[2] += __builtin_readcyclecounter() - tt_e3;
#endif
	}
#ifdef MWF_SYS_TIMING
	if (lane == 0 && t_blocks > 0 && (((wv == 0 || wv == 5) && (lb % 37) == 0) || t_acc[5] * 6 < t_acc[4]))
		printf("wg %3d wave %d: deep-block iterations %llu: stage 2a (incl. wait for the probe words) %.0f  stage 1 %.0f  stage 2b %.0f cycles | %llu slot-blocks in %llu epochs | per slot-block: wait %.0f  refresh %.0f  steps %.0f  publish %.0f cycles | per epoch: blocks %.0f  end (barriers, scan, shrink) %.0f = first barrier %.0f + scan %.0f + shrink and second barrier %.0f\n", lb, wv,
		       t_st[3], (double)t_st[0] / (t_st[3] ? t_st[3] : 1), (double)t_st[1] / (t_st[3] ? t_st[3] : 1), (double)t_st[2] / (t_st[3] ? t_st[3] : 1), t_blocks, t_runs, (double)t_acc[0] / t_blocks, (double)t_acc[1] / t_blocks, (double)t_acc[2] / t_blocks, (double)t_acc[3] / t_blocks, (double)t_acc[4] / t_runs, (double)t_acc[5] / t_runs, (double)t_ee[0] / t_runs, (double)t_ee[1] / t_runs, (double)t_ee[2] / t_runs);
#endif
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	R.s = s, R.cells = cells;
	if (SEG) R.n_snap = A.step > 0 ? s / A.step : 0; // snapshots of the penalties that exist (miniwfa.c:585: none is taken once the end cell is found)
	return R;
}

// which pair a group of workgroups works on, and where its pass state lives
__device__ __forceinline__ int32_t group_pair(const BatchArgs &A, int32_t grp) { return A.coop_pair_ids ? A.coop_pair_ids[grp] : A.coop_pair; }
__device__ __forceinline__ int32_t *group_state(const BatchArgs &A, int32_t grp) { return (int32_t*)((char*)A.coop_flags + (int64_t)grp * A.coop_misc_stride + 2048); }

__device__ __forceinline__ void sys_pair_mem(const BatchArgs &A, int32_t grp, int32_t pair, PairMem &M)
{
	pair_mem(A, grp, pair, M); // ring / good / tb slots are per group
	M.good = A.good + (int64_t)grp * A.pen.nH * A.GW;
	if (A.sys_ep) M.ep = A.sys_ep + (int64_t)grp * A.sys_ep_stride, M.ep_ow = 64 * A.sys_c - 2 * A.sys_p, M.ep_p = A.sys_p, M.ep_kw = 64 * A.sys_c;
}

// DEFER: the match extension of a penalty runs behind the recurrence of the next one (every H lag >= 3)
template <int E1, int E2, int P, bool DEFER, bool TB, int C>
__global__ __launch_bounds__(kT) void wfa_sys_kernel(const BatchArgs A)
{
	__shared__ SysLds L;
	const int32_t G = A.coop_group_size, grp = (int32_t)blockIdx.x / G, lb = (int32_t)blockIdx.x % G;
	const int32_t pair = group_pair(A, grp);
	int32_t *const state = group_state(A, grp);
	PairMem M;
	sys_pair_mem(A, grp, pair, M);
	if (threadIdx.x == 0) L.red[0] = 0;
	__syncthreads();
	int32_t n_seg = 0;
	if (A.coop_pass == 2) { // second pass of the low-memory mode: checkpoints left by the walk / the provenance trace
		if (state[0] != ST_OK) return; // first pass failed: nothing to do, the finish kernel reports it
		n_seg = state[3];
	}
	const PassResult R = sys_pass<E1, E2, TB, P, DEFER, C>(A, M, L, TB ? n_seg : 0, grp, lb, G);
	if (lb == 0 && threadIdx.x == 0) {
		int32_t *st = state + (A.coop_pass == 2 ? 8 : 0);
		st[0] = R.status, st[1] = R.s, st[2] = R.info;
		st[4] = (int32_t)(R.cells & 0xffffffff), st[5] = (int32_t)(R.cells >> 32);
		st[6] = 0;
	}
}

// The provenance pass of the true low-memory mode (coop_pass == 3): no traceback, snapshots every `step` penalties.
template <int E1, int E2, int P, bool DEFER, int C>
__global__ __launch_bounds__(kT) void wfa_sys_seg_kernel(const BatchArgs A)
{
	__shared__ SysLds L;
	const int32_t G = A.coop_group_size, grp = (int32_t)blockIdx.x / G, lb = (int32_t)blockIdx.x % G;
	const int32_t pair = group_pair(A, grp);
	int32_t *const state = group_state(A, grp);
	PairMem M;
	sys_pair_mem(A, grp, pair, M);
	if (threadIdx.x == 0) L.red[0] = 0;
	__syncthreads();
	const PassResult R = sys_pass<E1, E2, false, P, DEFER, C, true>(A, M, L, 0, grp, lb, G);
	if (lb == 0 && threadIdx.x == 0) {
		state[0] = R.status, state[1] = R.s, state[2] = R.info; // info: the end cell's provenance
		state[4] = (int32_t)(R.cells & 0xffffffff), state[5] = (int32_t)(R.cells >> 32);
		state[6] = R.n_snap;
	}
}

// Checkpoints of the true low-memory mode: chase the provenance of the end cell back through the snapshots (reference wf_traceback_seg,
// miniwfa.c:528-549).  An index decodes to (slice, owner chunk, column) with the snapshot's chunk range; the slice gives the penalty: H ring
// slices and E/F registers are numbered by age (0 = the penalty the snapshot was taken behind).
__global__ void sys_trace_kernel(const BatchArgs A)
{
	if (threadIdx.x != 0) return;
	const int32_t grp = (int32_t)blockIdx.x; // one block per pair
	int32_t *st = group_state(A, grp);
	st[3] = 0;
	if (st[0] != ST_OK) return;
	const Penalty &P = A.pen;
	PairMem M;
	sys_pair_mem(A, grp, group_pair(A, grp), M);
	const int32_t n_snap = st[6];
	if (n_snap > A.seg_slot) { st[0] = ST_SNAP_OVERFLOW; return; }
	int32_t last = st[2];
	for (int32_t j = n_snap - 1; j >= 0; --j) {
		const int32_t *meta = M.snap_meta + (int64_t)j * 8;
		const int64_t base = (int64_t)(uint32_t)meta[0] | (int64_t)meta[1] << 32;
		const int32_t S = meta[2], gA = meta[4], n_ep = meta[5], OW = meta[6], per = n_ep * OW;
		if (last < 0 || per <= 0) { st[0] = ST_INTERNAL; return; }
		const int32_t id = last / per, rem = last - id * per, col = (gA + rem / OW) * OW + rem % OW;
		int32_t age;
		if (id < P.nH) age = id;
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

// Checkpoints of the low-memory mode from the full traceback matrix of a first pass (
// reference miniwfa.c:495-549): the same walk over this kernel's traceback layout.
__global__ void sys_walk_kernel(const BatchArgs A)
{
	// One wave, every lane walking the same chain (their loads are one broadcast): a step is one dependent byte of a matrix written long ago — a round trip
	// to HBM.  Every 40 rows the 64 lanes fetch the byte in the walk's column of the next 64 rows together, so that the steps through them hit L2
	// (dev::traceback_wave does the same).
	const int32_t lane = (int32_t)threadIdx.x;
	const int32_t grp = (int32_t)blockIdx.x; // one block per pair
	int32_t *st = group_state(A, grp);
	if (lane == 0) st[3] = 0;
	if (st[0] != ST_OK) return;
	const Penalty &P = A.pen;
	PairMem M;
	sys_pair_mem(A, grp, group_pair(A, grp), M);
	const int32_t s_final = st[1], step = A.step;
	const int32_t n_seg = s_final / step;
	if (n_seg > A.seg_slot) { if (lane == 0) st[0] = ST_SNAP_OVERFLOW; return; }
	int32_t arr = 0, s = s_final, col = M.ql + 1; // array 0=H 1=E1 2=F1 3=E2 4=F2; the end cell is on diagonal ql-tl
	int32_t j = n_seg - 1, pf_at = s_final;
	while (j >= 0) {
		const int32_t Sj = (j + 1) * step - 1;
		if (s <= Sj) { // first cell of the chain that already existed at snapshot j
			if (lane == 0) M.seg[2 * j] = s, M.seg[2 * j + 1] = col;
			--j;
			continue;
		}
		if (s <= 0) { if (lane == 0) st[0] = ST_INTERNAL; return; }
		if (s <= pf_at && M.ep && A.tb_slot_bytes > 0) {
			const int32_t r = s - 2 - lane;
			uint32_t pf = 0;
			if (r >= 0) {
				const int64_t base = M.ep[2 * (r >> 8)], gn = M.ep[2 * (r >> 8) + 1];
				const int32_t g = col / M.ep_ow;
				int64_t at = base + ((int64_t)(r & 255) * (int32_t)(gn >> 32) + (g - (int32_t)(gn & 0xffffffff))) * M.ep_kw + (col - (g * M.ep_ow - M.ep_p));
				at = min(max(at, (int64_t)0), (int64_t)A.tb_slot_bytes - 1); // (an older row may not reach this column: any byte of the arena will do)
				pf = M.tb[at];
			}
			asm volatile("" :: "v"(pf));
			pf_at = s - 40;
		}
		const uint32_t x = tb_byte(M, s - 1, col);
		if (arr == 0) {
			const uint32_t z = x & 7u;
			if (z == 0) s -= P.x;
			else arr = (int32_t)z == 1 ? 1 : (int32_t)z == 2 ? 2 : (int32_t)z == 3 ? 3 : 4;
		} else if (arr == 1) { if (x & 0x08u) s -= P.e1; else s -= P.oe1, arr = 0; col -= 1; }
		else if (arr == 2) { if (x & 0x10u) s -= P.e1; else s -= P.oe1, arr = 0; col += 1; }
		else if (arr == 3) { if (x & 0x20u) s -= P.e2; else s -= P.oe2, arr = 0; col -= 1; }
		else { if (x & 0x40u) s -= P.e2; else s -= P.oe2, arr = 0; col += 1; }
	}
	if (lane == 0) st[3] = n_seg;
}

__global__ __launch_bounds__(64) void sys_finish_kernel(const BatchArgs A)
{
	const int32_t grp = (int32_t)blockIdx.x; // one block per pair
	const int32_t pair = group_pair(A, grp);
	PairMem M;
	sys_pair_mem(A, grp, pair, M);
	const int32_t *st1 = group_state(A, grp), *st = A.step > 0 && A.want_cigar ? st1 + 8 : st1;
	PassResult R;
	int32_t status = st1[0] != ST_OK ? st1[0] : st[0];
	R.status = status, R.s = st[1], R.info = st[2], R.n_snap = 0;
	R.cells = (int64_t)(uint32_t)st[4] | (int64_t)st[5] << 32;
	const int64_t cells1 = A.step > 0 && A.want_cigar ? ((int64_t)(uint32_t)st1[4] | (int64_t)st1[5] << 32) : 0;
	if (A.dbg && (status == ST_OK || status == ST_STOPPED)) { // band trace (diagnostics): lo,hi of every slice, from the edge log
		const int32_t *logL = A.sys_log + (int64_t)grp * A.sys_log_stride, *logH = logL + A.sys_log_stride / 2;
		const int32_t cmax = M.tl + M.ql + 1, n_seg = A.step > 0 && A.want_cigar ? seg_effective(M.seg, st1[3]) : 0;
		for (int32_t sp = threadIdx.x; sp < R.s && sp < A.dbg_cap; sp += 64) {
			int32_t a = logL[sp], b = logH[sp];
			for (int32_t j = 0; j < n_seg; ++j)
				if (M.seg[2 * j] == sp) a = b = M.seg[2 * j + 1];
			M.dbg[2 * sp] = a > 1 ? a - 1 : 1, M.dbg[2 * sp + 1] = b < cmax ? b + 1 : cmax;
		}
	}
	finish_pair(A, M, grp, pair, R, status, cells1);
}

// One pass with P penalties per hand-off block: the provenance pass of the true low-memory mode (coop_pass == 3: never with traceback) or the plain / second pass.
// The waits between workgroups rely on every workgroup being resident.  The grid is sized for that (one per CU, sys_max_grid) and the engine keeps this
// library's other kernels off the device meanwhile; a cooperative launch makes the runtime refuse a grid that could not be resident whatever else the
// process runs.  (A plain launch if the runtime refuses: the waits are bounded.)
template <int E1, int E2, int P, bool DEFER, bool TB, int C>
int launch_pass_pc(const BatchArgs &a, int grid, hipStream_t st)
{
	const void *fn;
	if (a.coop_pass == 3) {
		if constexpr (TB) return -1;
		else fn = reinterpret_cast<const void*>(&wfa_sys_seg_kernel<E1, E2, P, DEFER, C>);
	} else fn = reinterpret_cast<const void*>(&wfa_sys_kernel<E1, E2, P, DEFER, TB, C>);
	BatchArgs arg = a;
	void *args[] = {(void*)&arg};
	if (a.sys_coop_launch) {
		if (hipLaunchCooperativeKernel(fn, dim3(grid), dim3(kT), args, 0, st) == hipSuccess) return 0;
		(void)hipGetLastError();
	}
	if (hipLaunchKernel(fn, dim3(grid), dim3(kT), args, 0, st) != hipSuccess) { (void)hipGetLastError(); return -2; }
	return 0;
}

template <int E1, int E2, bool DEFER, bool TB, int C>
int launch_pass_c(const BatchArgs &a, int grid, hipStream_t st)
{
	switch (a.sys_p) {
#ifdef MWF_SYS_ALL_P // (experiments: profiles/coop_quick.py with MWF_SYS_P, profiles/mhc_lowmem_p.py)
	case 4:  return launch_pass_pc<E1, E2, 4, DEFER, TB, C>(a, grid, st);
	case 16: return launch_pass_pc<E1, E2, 16, DEFER, TB, C>(a, grid, st);
#endif
	case 8:  return launch_pass_pc<E1, E2, 8, DEFER, TB, C>(a, grid, st);
	default: return -1; // no kernel for this block length: the host's layout (boxes, traceback rows) would not be the kernel's
	}
}

template <int E1, int E2, bool DEFER, bool TB>
int launch_pass_p(const BatchArgs &a, int grid, hipStream_t st)
{
	if (a.sys_c == 1) return launch_pass_c<E1, E2, DEFER, TB, 1>(a, grid, st);
#ifdef MWF_SYS_C2 // (experiment: 128-column slots that own 112)
	if (a.sys_c == 2) return launch_pass_c<E1, E2, DEFER, TB, 2>(a, grid, st);
#endif
	return launch_pass_c<E1, E2, DEFER, TB, 4>(a, grid, st);
}

// DEFER needs every H lag >= 3 (the edit-distance preset, every lag 1, only ever takes the plain form)
template <int E1, int E2, bool CAN_DEFER>
int launch_pass_d(const BatchArgs &a, int grid, hipStream_t st)
{
	const bool defer = CAN_DEFER && a.pen.x >= 3 && a.pen.oe1 >= 3 && a.pen.oe2 >= 3;
	if (a.coop_pass == 3) { // the provenance pass stores no traceback
		if constexpr (CAN_DEFER) {
			// (measured, round 5: the provenance pass carries twice the wavefront state — with four columns per lane the deferred form needs ~316 VGPRs
			// and spills 60 of them; without the deferral 25, and the 5 Mb pair's first pass falls from 1.06 to 0.97 s; one column per lane: 125 -> 122 ms
			// on the 150 kb pair: the deferred provenance kernel is no longer built.)
		}
		return launch_pass_p<E1, E2, false, false>(a, grid, st);
	}
	if constexpr (CAN_DEFER) {
		// (... and so does the traceback pass on four columns per lane, 59 spilled VGPRs: the 5 Mb pair in high-memory CIGAR mode 851 -> 808 ms undeferred;
		// on one column per lane the deferral wins, 56.6 against 59.4 ms on the 150 kb pair)
		if (defer && a.want_cigar && a.sys_c == 4) return launch_pass_p<E1, E2, false, true>(a, grid, st);
		if (defer) return a.want_cigar ? launch_pass_p<E1, E2, true, true>(a, grid, st) : launch_pass_p<E1, E2, true, false>(a, grid, st);
	}
	return a.want_cigar ? launch_pass_p<E1, E2, false, true>(a, grid, st) : launch_pass_p<E1, E2, false, false>(a, grid, st);
}

} // namespace

int64_t sys_chunk_slots(int grid) { return (int64_t)grid * kNW * kK; }
int64_t coop_chunk_slots(int grid) { return sys_chunk_slots(grid); }
// penalties the whole-device kernel is instantiated for; the per-slot window history lives in LDS
bool coop_supported(const Penalty &p)
{
	return ((p.e1 == 2 && p.e2 == 1) || (p.e1 == 2 && p.e2 == 2) || (p.e1 == 1 && p.e2 == 1)) && p.nH <= kMaxRing;
}
int sys_owned_cols(int p, int c) { return 64 * c - 2 * p; }
bool sys_p_supported(int p)
{
#ifdef MWF_SYS_ALL_P
	return p == 4 || p == 8 || p == 16;
#else
	return p == 8; // the product build instantiates one block length (P = 4 and 16 measured slower)
#endif
}
bool sys_c_supported(int c)
{
#ifdef MWF_SYS_C2
	if (c == 2) return true; // (experiment: 128-column slots)
#endif
	return c == 1 || c == 4;
}
int64_t sys_box_ints(int p, bool seg) { return ((seg ? 4 : 2) * p * (p + 8) + 2 * p + 31) / 32 * 32; } // (whatever the columns per lane: 2 (p/c) lanes x (p + 8) c ints; twice that with provenance)

int sys_max_grid()
{
	int dev = 0, n_cu = 0, per = 0;
	if (hipGetDevice(&dev) != hipSuccess) return 0;
	if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
	if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, wfa_sys_kernel<2, 1, 8, true, true, 4>, kT, 0) != hipSuccess || per < 1) return 0;
	return n_cu; // one workgroup per CU: every one of them is resident, which the waits between them rely on
}

int launch_sys_pass(const BatchArgs &a, int grid, void *stream)
{
	if (a.pen.e1 == 2 && a.pen.e2 == 1) return launch_pass_d<2, 1, true>(a, grid, (hipStream_t)stream);
	if (a.pen.e1 == 2 && a.pen.e2 == 2) return launch_pass_d<2, 2, true>(a, grid, (hipStream_t)stream);
	if (a.pen.e1 == 1 && a.pen.e2 == 1) return launch_pass_d<1, 1, false>(a, grid, (hipStream_t)stream);
	return -1;
}

int launch_sys_trace(const BatchArgs &a, void *stream)
{
	hipLaunchKernelGGL(sys_trace_kernel, dim3(a.coop_groups > 0 ? a.coop_groups : 1), dim3(64), 0, (hipStream_t)stream, a);
	return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_sys_walk(const BatchArgs &a, void *stream)
{
	hipLaunchKernelGGL(sys_walk_kernel, dim3(a.coop_groups > 0 ? a.coop_groups : 1), dim3(64), 0, (hipStream_t)stream, a);
	return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_sys_finish(const BatchArgs &a, void *stream)
{
	hipLaunchKernelGGL(sys_finish_kernel, dim3(a.coop_groups > 0 ? a.coop_groups : 1), dim3(64), 0, (hipStream_t)stream, a);
	return hipGetLastError() == hipSuccess ? 0 : -2;
}

} // namespace mwf
