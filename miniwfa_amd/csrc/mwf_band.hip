// mwf_band.hip — register-resident band kernel (the fast path for batches of short/medium pairs).
//
// Same arithmetic and bookkeeping as the generic kernel (mwf_kernels.hip), different data placement,
// chosen from the rocprof profile of the generic kernel (83 % of wave-cycles waiting on memory, every
// one of the 5 stores per cell going out to HBM):
//
//   * The E1/F1/E2/F2 wavefronts are only ever read e1 resp. e2 penalties later (reference
//     miniwfa.c:255-257), so they never leave the chip: every thread OWNS fixed columns and keeps their
//     last e1 (e2) values in VGPRs.  A wave owns 256 consecutive columns per slot (4 per lane, 2 slots);
//     the d-1 / d+1 neighbours the recurrence needs (:269-273) come from the adjacent lane through a DPP
//     wave shift, and across wave boundaries through a 2 KB LDS edge table written once per penalty.
//     Columns outside the current window push NEG_INF, which is exactly what the reference's pads yield.
//   * Only H goes to HBM (it is read again x, o1+e1 and o2+e2 penalties later): one 16-byte store and
//     three 16-byte loads per lane per 4 cells — the wide coalesced pattern — instead of 9 dword loads
//     and 5 dword stores per cell.  16 algorithmic bytes per cell instead of 48.
//   * The three H rows of penalty s+1 were final at least one penalty earlier (every H lag >= 2 here), so
//     their loads are issued BEFORE the barrier that ends penalty s and stay in flight across it
//     (raw s_barrier behind a counted s_waitcnt vmcnt): HBM/L2 latency is off the critical path.
//   * Both sequences are copied to LDS once per pair; match extension reads two aligned dwords and
//     v_alignbyte's them, so the per-cell probe never leaves the CU.
//
// Column -> owner mapping.  Columns are cut in 256-column chunks; chunk g belongs to (wave, slot) with
// g mod (waves*2) == wave + waves*slot.  A workgroup therefore holds any window narrower than
// (waves*2 - 1) chunks; one chunk is always spare, which is what makes the mapping change safe (a chunk
// that becomes mapped has been outside every window for hundreds of penalties, so its registers already
// hold NEG_INF).  A pair whose window outgrows that span is reported ST_BAND_OVERFLOW and re-run by the
// host on the generic kernel.
#include <type_traits>
#include "mwf_device.h"

namespace mwf {

using namespace dev;

namespace {

extern __shared__ __attribute__((aligned(16))) uint8_t lds_dyn[];

constexpr int kChunk = 256;

// lane i <- lane i-1 (lane 0 keeps `fill`); lane i <- lane i+1 (lane 63 keeps `fill`).  All 64 lanes must be active.
__device__ __forceinline__ int32_t from_left(int32_t v, int32_t fill) { return __builtin_amdgcn_update_dpp(fill, v, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ int32_t from_right(int32_t v, int32_t fill) { return __builtin_amdgcn_update_dpp(fill, v, 0x130, 0xf, 0xf, false); }

// four bytes at an arbitrary byte offset of an LDS array.  (gfx950 does read LDS unaligned — a plain ds_read_b32 at the
// byte address works — but measures far slower: 37 ms against 28.5 ms for the benchmark batch.)
__device__ __forceinline__ uint32_t lds_ld4(const uint8_t *base, int32_t off)
{
	const uint32_t *p = (const uint32_t*)(base + (off & ~3));
	return __builtin_amdgcn_alignbyte(p[1], p[0], (uint32_t)off); // v_alignbyte_b32 shifts by the low two bits of its third operand
}

template <bool LSEQ>
__device__ __forceinline__ int32_t extend_cell(const PairMem &M, const uint8_t *lt, const uint8_t *lq, int32_t k, int32_t d)
{
	if (!LSEQ) return extend_run(M.ts, M.qs, M.tl, M.ql, k, d);
	const int32_t j = k + 1, i = d + j;
	const int32_t room = min(M.tl - j, M.ql - i);
	int32_t n = 0;
	while (n < room) {
		const uint32_t x = lds_ld4(lt, j + n) ^ lds_ld4(lq, i + n);
		if (x) { n += (int32_t)(__builtin_ctz(x) >> 3); break; }
		n += 4;
	}
	return k + min(n, room);
}

// four sequence bytes at t[j..] and q[i..] XORed (0 bits = equal bytes)
template <bool LSEQ>
__device__ __forceinline__ uint32_t probe4(const PairMem &M, const uint8_t *lt, const uint8_t *lq, int32_t j, int32_t i)
{
	if (LSEQ) return lds_ld4(lt, j) ^ lds_ld4(lq, i);
	uint32_t a, b;
	__builtin_memcpy(&a, M.ts + j, 4);
	__builtin_memcpy(&b, M.qs + i, 4);
	return a ^ b;
}

// Length of the exact-match run t[j..] == q[i..] (at most `room`, the first n0 bytes already known equal), walked by all
// 64 lanes together: 256 bytes per trip.  Every argument is wave-uniform.
template <bool LSEQ>
__device__ __forceinline__ int32_t run_wave(const PairMem &M, const uint8_t *lt, const uint8_t *lq, int32_t j, int32_t i, int32_t room, int32_t n0)
{
	const int32_t lane = threadIdx.x & 63;
	int32_t n = n0;
	while (n < room) {
		const int32_t off = n + 4 * lane;
		int32_t m = 0;
		if (off < room) {
			const uint32_t x = probe4<LSEQ>(M, lt, lq, j + off, i + off);
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

// lanes l of interleaved word k (column = base + 4*l + k) whose column lies in [a,b]
__device__ __forceinline__ unsigned long long lane_mask(int32_t base, int32_t k, int32_t a, int32_t b)
{
	int32_t lmin = a - base - k, lmax = b - base - k;
	if (lmax < 0) return 0ull;
	lmin = lmin <= 0 ? 0 : (lmin + 3) >> 2;
	lmax = min(lmax >> 2, 63);
	if (lmin > lmax) return 0ull;
	return (~0ull >> (63 - lmax)) & (~0ull << lmin);
}

__device__ __forceinline__ int32_t pick4(int32_t i, int32_t a0, int32_t a1, int32_t a2, int32_t a3)
{
	return i == 0 ? a0 : i == 1 ? a1 : i == 2 ? a2 : a3;
}

// in_matrix() without short-circuit control flow
__device__ __forceinline__ uint32_t inm_bit(int32_t d, int32_t k, int32_t tl, int32_t ql)
{
	return (uint32_t)((uint32_t)(k + 1) < (uint32_t)(tl + 1)) & (uint32_t)((uint32_t)(d + k + 1) < (uint32_t)(ql + 1));
}

template <int T, int K, int E1, int E2, bool TB, bool LSEQ, bool PACK>
__device__ PassResult band_pass(const BatchArgs &A, const PairMem &M, Shared &sh,
                                int32_t (*edge)[(T / 64) * K][4], const uint8_t *lt, const uint8_t *lq,
                                int32_t n_seg, bool trace_band)
{
	constexpr int NW = T / 64, NWK = NW * K, D = (E1 > E2 ? E1 : E2) + 1;
	const int32_t tl = M.tl, ql = M.ql, cmax = tl + ql + 1;
	const int32_t tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6);
	const int32_t W = A.W, nH = A.pen.nH, lagx = A.pen.x, lag1 = A.pen.oe1, lag2 = A.pen.oe2;
	int32_t *const H = M.H;
	// every H access below is row * W + column with an unsigned 32-bit byte offset from the workgroup's (uniform) base: band
	// kernel rows are short, and this is the SGPR-base + VGPR-offset addressing form (no 64-bit VALU address arithmetic)
	auto at = [&](int32_t row, int32_t col) -> int32_t* { return (int32_t*)((char*)H + (size_t)((uint32_t)(row * W + col) << 2)); };
	// H rows written at penalty s are loaded again, at the earliest, by the prefetch issued at the start of penalty
	// s + lag - 1.  With every lag >= 3 the stores of a penalty may therefore stay in flight across its barrier.
	const bool relaxed_stores = min(lagx, min(lag1, lag2)) >= 3;
	PassResult R;
	R.status = ST_OK, R.s = 0, R.info = 0, R.n_snap = 0, R.cells = 0;

	// per-thread wavefront state: [age][slot][column-in-lane]; age 0 is the previous penalty.
	// PACK: two columns per register as int16 (the host only selects it when every offset that can occur, phantom ones
	// included, is below 32767 and the penalty count is too, so a dead value — stored as max(v, -32768) — can drift up by
	// one per penalty without ever reaching -1).  Halves the state registers: what lets two workgroups share a CU.
	constexpr int NS = PACK ? 2 : 4;
	int32_t e1h[E1][K][NS], f1h[E1][K][NS], e2h[E2][K][NS], f2h[E2][K][NS];
	constexpr int32_t kDeadState = PACK ? (int32_t)0x80008000u : kNegInf;
#pragma unroll
	for (int k = 0; k < K; ++k)
#pragma unroll
		for (int i = 0; i < NS; ++i) {
#pragma unroll
			for (int a = 0; a < E1; ++a) e1h[a][k][i] = f1h[a][k][i] = kDeadState;
#pragma unroll
			for (int a = 0; a < E2; ++a) e2h[a][k][i] = f2h[a][k][i] = kDeadState;
		}
	auto col_of = [](const int32_t (&v)[NS], int i) -> int32_t { // column i (0..3) of a state vector
		if (!PACK) return v[i];
		return (i & 1) ? (v[i >> 1] >> 16) : (int32_t)(int16_t)(v[i >> 1] & 0xffff);
	};
	auto pack2 = [](int32_t a, int32_t b) -> int32_t { // two offsets -> one register; dead values clamp to -32768
		typedef short short2_t __attribute__((ext_vector_type(2)));
		const short2_t v = __builtin_amdgcn_cvt_pk_i16(a, b); // saturating: live values are < 32767 by the host's choice of this variant
		return __builtin_bit_cast(int32_t, v);
	};
	int4 phx[K], po1[K], po2[K];   // prefetched H rows of the next penalty: lags x, o1+e1, o2+e2
	int32_t pe1[K], pe2[K];        // lane 0: column to the left of the chunk, lane 63: column to its right

	// ---- penalty 0 (reference wf_stripe_init, miniwfa.c:103-121) and its extension
	for (int32_t j = tid; j < D * NWK * 4; j += T) (&edge[0][0][0])[j] = kNegInf;
	if (tid == 0) {
		for (int32_t j = 0; j < nH; ++j) sh.rng_lo[j] = 1, sh.rng_hi[j] = 0;
		for (int32_t j = 0; j < 12; ++j) (&sh.flags[0][0])[j] = 0;
		const int32_t c0 = tl + 1;
		const int32_t k0 = extend_cell<LSEQ>(M, lt, lq, -1, 0);
		H[c0] = k0;
		sh.rng_lo[0] = sh.rng_hi[0] = c0;
		sh.word[1] = k0;
	}
	__syncthreads();
	{
		const int32_t k0 = uni(sh.word[1]);
		if (k0 == tl - 1 && k0 == ql - 1) return R;
	}

	int32_t s = 0, wf_lo = tl + 1, wf_hi = tl + 1;
	int32_t curH = 0, par = 0, sid = 0, dcur = 0;
	int64_t cells = 0, tb_used = 0;

	// Loads of the three H rows penalty `slotH` reads, for this wave's chunk of slot k under the mapping that starts at
	// chunk g_lo.  ALWAYS exactly five loads: a chunk beyond column phi loads columns 0..3 instead, so that the
	// s_waitcnt before the barrier can use a fixed count.
	auto prefetch = [&](int k, int32_t slotH, int32_t phi, int32_t g_lo) {
		int32_t jx = slotH - lagx; if (jx < 0) jx += nH;
		int32_t j1 = slotH - lag1; if (j1 < 0) j1 += nH;
		int32_t j2 = slotH - lag2; if (j2 < 0) j2 += nH;
		int32_t g = g_lo - g_lo % NWK + wave + NW * k;
		if (g < g_lo) g += NWK;
		const int32_t c0 = g * kChunk <= phi ? g * kChunk + 4 * lane : 0;
		phx[k] = *(const int4*)at(jx, c0);
		po1[k] = *(const int4*)at(j1, c0);
		po2[k] = *(const int4*)at(j2, c0);
		const int32_t ce = lane == 0 ? max(c0 - 1, 0) : c0 + 4; // only lanes 0 and 63 use it; column 0 is a pad
		pe1[k] = *at(j1, ce), pe2[k] = *at(j2, ce);
	};
	int32_t gl; // lowest chunk of the mapping the prefetched registers were loaded under
	{
		const int32_t lo1 = wf_lo > 1 ? wf_lo - 1 : 1, hi1 = wf_hi < cmax ? wf_hi + 1 : cmax;
		gl = lo1 >> 8;
		if (!PACK)
#pragma unroll
			for (int k = 0; k < K; ++k) prefetch(k, 1, hi1, gl);
	}

#ifdef MWF_BAND_TIMING
	unsigned long long t_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, t_steps = 0, t_active = 0, t_pend = 0, t_last = 0;
#define MWF_TICK(i) do { const unsigned long long t_now = __builtin_readcyclecounter(); t_acc[i] += t_now - t_last; t_last = t_now; } while (0)
#else
#define MWF_TICK(i) do {} while (0)
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
		const int32_t lo = wf_lo > 1 ? wf_lo - 1 : 1;       // miniwfa.c:417-418
		const int32_t hi = wf_hi < cmax ? wf_hi + 1 : cmax;
		const int32_t s_new = s + 1;
		const int32_t newH = curH + 1 == nH ? 0 : curH + 1;
		const int32_t npar = par + 1 == 3 ? 0 : par + 1;
		const int32_t dnew = dcur + 1 == D ? 0 : dcur + 1;
		const int32_t origin = lo & ~3;
		const int32_t row_bytes = (hi | 3) - origin + 1;
		if (TB) {
			if (s_new - 1 >= A.rows_slot) { R.status = ST_ROWS_OVERFLOW; break; }
			if (tb_used + row_bytes > A.tb_slot_bytes) { R.status = ST_TB_OVERFLOW; break; }
		}
		// the window of penalty s_new+1 lies inside [lo-1, hi+1] whatever the flags say; it must fit the register span
		const int32_t plo = lo > 1 ? lo - 1 : 1, phi = hi < cmax ? hi + 1 : cmax;
		const int32_t gl_next = plo >> 8;
		if ((phi >> 8) - gl_next + 1 > NWK - 1) { R.status = ST_BAND_OVERFLOW; break; }
		const int32_t nextH = newH + 1 == nH ? 0 : newH + 1;
		// windows of the three H source slices (reference wf_next_prep, miniwfa.c:252-254)
		int32_t jx = newH - lagx; if (jx < 0) jx += nH;
		int32_t j1 = newH - lag1; if (j1 < 0) j1 += nH;
		int32_t j2 = newH - lag2; if (j2 < 0) j2 += nH;
		const int32_t xlo = uni(sh.rng_lo[jx]), xhi = uni(sh.rng_hi[jx]);
		const int32_t alo = uni(sh.rng_lo[j1]), ahi = uni(sh.rng_hi[j1]);
		const int32_t blo = uni(sh.rng_lo[j2]), bhi = uni(sh.rng_hi[j2]);
		// columns whose every H read (c and c+-1) falls inside its source window and that are inside [lo,hi]
		const int32_t ilo = max(max(lo, xlo), max(alo, blo) + 1), ihi = min(min(hi, xhi), min(ahi, bhi) - 1);
		const bool track_good = (((256 - (s_new & 255)) & 255) < nH);
		// ages of the LDS edge table to read: penalty s_new-E1 and s_new-E2
		int32_t d1 = dnew - E1; if (d1 < 0) d1 += D;
		int32_t d2 = dnew - E2; if (d2 < 0) d2 += D;

		if (tid == 0) {
			sh.rng_lo[newH] = lo, sh.rng_hi[newH] = hi;
			const int32_t nn = npar + 1 == 3 ? 0 : npar + 1; // flags of the NEXT penalty (see mwf_kernels.hip)
			sh.flags[nn][0] = sh.flags[nn][1] = sh.flags[nn][2] = sh.flags[nn][3] = 0;
			if (TB) M.row_off[s_new - 1] = tb_used, M.row_lo[s_new - 1] = origin;
			if (trace_band && s_new - 1 < A.dbg_cap) M.dbg[2 * (s_new - 1)] = lo, M.dbg[2 * (s_new - 1) + 1] = hi;
		}

#ifdef MWF_BAND_TIMING
		const unsigned long long t_b = __builtin_readcyclecounter();
		int32_t t_nact = 0;
#endif
		const int32_t gbase = gl - gl % NWK;
		if (PACK) prefetch(0, newH, hi, gl);
		bool act_k[K];
#pragma unroll
		for (int k = 0; k < K; ++k) {
			int32_t g = gbase + wave + NW * k;
			if (g < gl) g += NWK;
			act_k[k] = g * kChunk <= hi && g * kChunk + kChunk - 1 >= lo;
		}
		{ // the waves with the most chunks to do set the pace of the penalty: let them issue first
			int n_act = 0;
#pragma unroll
			for (int k = 0; k < K; ++k) n_act += act_k[k] ? 1 : 0;
			if (n_act >= 2) __builtin_amdgcn_s_setprio(3);
			else __builtin_amdgcn_s_setprio(0);
		}
		// PACK (slot-pipelined loads): the rows of chunk k+1 are loaded while chunk k computes; only the first chunk's rows
		// of the next penalty cross the barrier.  Otherwise every chunk's rows of the next penalty are prefetched.
		auto refill = [&](int k) {
			if (!PACK) prefetch(k, nextH, phi, gl_next);
			else if (k + 1 < K) prefetch(k + 1, newH, hi, gl); // unconditional (an idle chunk loads columns 0..3): the registers
			                                                   // must be dead until here for the allocator to share them
			// (the packed variant keeps nothing in flight across the barrier: the co-resident workgroup hides chunk 0's load latency)
		};
#pragma unroll
		for (int k = 0; k < K; ++k) {
			const int32_t r = wave + NW * k;
			int32_t g = gbase + r;
			if (g < gl) g += NWK;
			const int32_t cb = g * kChunk;
			const bool active = act_k[k]; // this wave's chunk meets the window (uniform)
			int32_t ne1[4], nf1[4], ne2[4], nf2[4];
			// publish this chunk's outer columns for the neighbouring waves, then age the registers
			auto retire = [&]() {
				if (lane == 63) edge[dnew][r][0] = ne1[3], edge[dnew][r][1] = ne2[3];
				if (lane == 0) edge[dnew][r][2] = nf1[0], edge[dnew][r][3] = nf2[0];
#pragma unroll
				for (int i = 0; i < NS; ++i) {
#pragma unroll
					for (int a = E1 - 1; a > 0; --a) e1h[a][k][i] = e1h[a - 1][k][i], f1h[a][k][i] = f1h[a - 1][k][i];
#pragma unroll
					for (int a = E2 - 1; a > 0; --a) e2h[a][k][i] = e2h[a - 1][k][i], f2h[a][k][i] = f2h[a - 1][k][i];
					if (PACK) {
						e1h[0][k][i] = pack2(ne1[2 * i], ne1[2 * i + 1]), f1h[0][k][i] = pack2(nf1[2 * i], nf1[2 * i + 1]);
						e2h[0][k][i] = pack2(ne2[2 * i], ne2[2 * i + 1]), f2h[0][k][i] = pack2(nf2[2 * i], nf2[2 * i + 1]);
					} else e1h[0][k][i] = ne1[i], f1h[0][k][i] = nf1[i], e2h[0][k][i] = ne2[i], f2h[0][k][i] = nf2[i];
				}
			};
			if (active) {
#ifdef MWF_BAND_TIMING
				++t_nact;
				{
					const unsigned long long t0 = __builtin_readcyclecounter();
					asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
					t_acc[3] += (t_last = __builtin_readcyclecounter()) - t0;
				}
#endif
				const int32_t c0 = cb + 4 * lane;
				const bool inner = cb >= ilo && cb + kChunk - 1 <= ihi; // uniform: no window test needed anywhere
				int32_t hx[4] = {phx[k].x, phx[k].y, phx[k].z, phx[k].w};
				int32_t o1[6], o2[6]; // o1[i+1] is column c0+i; o1[0], o1[5] the neighbours
				o1[1] = po1[k].x, o1[2] = po1[k].y, o1[3] = po1[k].z, o1[4] = po1[k].w;
				o2[1] = po2[k].x, o2[2] = po2[k].y, o2[3] = po2[k].z, o2[4] = po2[k].w;
				int32_t v1 = pe1[k], v2 = pe2[k];
				if (!PACK) refill(k); // the registers just read are free: start the next loads now
				if (!inner) {
#pragma unroll
					for (int i = 0; i < 4; ++i) {
						const int32_t c = c0 + i;
						hx[i] = ((c >= xlo) & (c <= xhi)) ? hx[i] : kNegInf;
						o1[i + 1] = ((c >= alo) & (c <= ahi)) ? o1[i + 1] : kNegInf;
						o2[i + 1] = ((c >= blo) & (c <= bhi)) ? o2[i + 1] : kNegInf;
					}
					const int32_t ce = lane == 0 ? c0 - 1 : c0 + 4;
					v1 = ((ce >= alo) & (ce <= ahi)) ? v1 : kNegInf;
					v2 = ((ce >= blo) & (ce <= bhi)) ? v2 : kNegInf;
				}
				o1[0] = from_left(o1[4], v1), o1[5] = from_right(o1[1], v1);
				o2[0] = from_left(o2[4], v2), o2[5] = from_right(o2[1], v2);
				// gap-extension sources: E of column c-1, F of column c+1, e1 (e2) penalties ago
				int32_t g1m[4], g1p[4], g2m[4], g2p[4];
				{
					const int32_t rl = r == 0 ? NWK - 1 : r - 1, rr = r + 1 == NWK ? 0 : r + 1;
					const int32_t le1 = edge[d1][rl][0], le2 = edge[d2][rl][1], rf1 = edge[d1][rr][2], rf2 = edge[d2][rr][3];
					g1m[0] = from_left(col_of(e1h[E1 - 1][k], 3), le1);
					g2m[0] = from_left(col_of(e2h[E2 - 1][k], 3), le2);
					g1p[3] = from_right(col_of(f1h[E1 - 1][k], 0), rf1);
					g2p[3] = from_right(col_of(f2h[E2 - 1][k], 0), rf2);
#pragma unroll
					for (int i = 1; i < 4; ++i) g1m[i] = col_of(e1h[E1 - 1][k], i - 1), g2m[i] = col_of(e2h[E2 - 1][k], i - 1);
#pragma unroll
					for (int i = 0; i < 3; ++i) g1p[i] = col_of(f1h[E1 - 1][k], i + 1), g2p[i] = col_of(f2h[E2 - 1][k], i + 1);
				}
				MWF_TICK(5);
				// ---- the recurrence, then the first 4-byte probe of the match extension, branch-free for all 4 columns
				int32_t hv[4], nmat[4];
				uint32_t tbw = 0, pend = 0, live = 0, fin = 0, gbits = 0;
				// Two copies of the column code: chunks strictly inside every window (nearly all of them) carry no window tests.
				// (Costs the two-workgroups-per-CU variant 60 B/lane of scratch and still measures 3 % faster: 28.5 against 29.3 ms.)
				constexpr bool kTwoCopies = true;
				auto columns = [&](auto inner_c) {
					constexpr bool INNER = decltype(inner_c)::value;
#pragma unroll
					for (int i = 0; i < 4; ++i) {
						const int32_t c = c0 + i, d = c - 1 - tl;
						const uint32_t act = (INNER || inner) ? 1u : (uint32_t)((c >= lo) & (c <= hi));
						const Cell v = wf_cell<TB>(hx[i], o1[i], g1m[i], o2[i], g2m[i], o1[i + 2], g1p[i], o2[i + 2], g2p[i]);
						ne1[i] = act ? v.e1 : kNegInf, nf1[i] = act ? v.f1 : kNegInf;
						ne2[i] = act ? v.e2 : kNegInf, nf2[i] = act ? v.f2 : kNegInf;
						const uint32_t inm = act & inm_bit(d, v.h, tl, ql);
						if (track_good) // uniform
							gbits |= (act & (inm | inm_bit(d, v.e1, tl, ql) | inm_bit(d, v.f1, tl, ql) | inm_bit(d, v.e2, tl, ql) | inm_bit(d, v.f2, tl, ql))) << i;
						const int32_t j = inm ? v.h + 1 : 0, q = inm ? d + v.h + 1 : 0;
						const int32_t room = inm ? min(tl - j, ql - q) : 0;
						const uint32_t x = probe4<LSEQ>(M, lt, lq, j, q);
						// equal leading bytes: ffs() - 1 is v_ffbl_b32, which yields ~0 for x == 0
						nmat[i] = min(min((int32_t)((uint32_t)(__builtin_ffs((int)x) - 1) >> 3), 4), room);
						pend |= ((uint32_t)(x == 0) & (uint32_t)(room > 4)) << i;
						hv[i] = v.h;
						tbw |= v.tb << (8 * i);
					}
				};
				if (kTwoCopies && inner) columns(std::true_type{});
				else columns(std::false_type{});
				retire(); // the new E/F are final and nothing below reads the old ages: frees their registers for the tail
				if ((uint32_t)(lo - cb) < (uint32_t)kChunk || (uint32_t)(hi - cb) < (uint32_t)kChunk) // uniform: this chunk holds an edge column.
					// Edge rule (miniwfa.c:325-326): H is the max of the five, so "any live" == "H live"
#pragma unroll
					for (int i = 0; i < 4; ++i) {
						const uint32_t lv = (uint32_t)(hv[i] >= -1);
						live |= (lv & (uint32_t)(c0 + i == lo)) | ((lv & (uint32_t)(c0 + i == hi)) << 1);
					}
				if (PACK) refill(k); // packed variant: after the register-hungry recurrence; the loads still overlap the probes and the tail
				MWF_TICK(6);
#ifdef MWF_BAND_TIMING
				if (__ballot(pend != 0)) ++t_pend;
#endif
				// A run of >= 4 matches continues (one cell in 256 by chance, plus the cells on the alignment path, whose runs are
				// 1/divergence long).  Each lane first walks its own runs 8 bytes per trip, four trips at most; what is still open
				// after that (long runs: low divergence) the whole wave walks together, 256 bytes per trip.
				if (__ballot(pend != 0)) {
					uint32_t open = 0;
					while (pend) {
						const int32_t ii = __builtin_ctz(pend);
						const int32_t hh = pick4(ii, hv[0], hv[1], hv[2], hv[3]);
						int32_t n = 4; // pend is only set for a full first probe of an in-matrix cell with room left
						const int32_t j = hh + 1, q = c0 + ii - 1 - tl + j, rm = min(tl - j, ql - q);
						for (int trip = 0; n < rm; ++trip) {
							if (trip == 4) { open |= 1u << ii; break; }
							const uint32_t xa = probe4<LSEQ>(M, lt, lq, j + n, q + n), xb = probe4<LSEQ>(M, lt, lq, j + n + 4, q + n + 4);
							if (xa | xb) { n += xa ? (int32_t)(__builtin_ctz(xa) >> 3) : 4 + (int32_t)(__builtin_ctz(xb) >> 3); break; }
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
							const int32_t n = run_wave<LSEQ>(M, lt, lq, j, q, rm, 36);
#pragma unroll
							for (int i = 0; i < 4; ++i) nmat[i] = (ii == i && lane == src) ? n : nmat[i];
						}
					}
				}
				MWF_TICK(4);
				// termination test of the extension sweep (miniwfa.c:405-409): the end cell (tl-1, ql-1) lies on diagonal ql-tl,
				// i.e. column ql+1, and nowhere else
				int32_t done_info = 0;
				const int32_t cfin = ql + 1;
#pragma unroll
				for (int i = 0; i < 4; ++i) hv[i] += nmat[i];
				if (cfin >= cb && cfin < cb + kChunk && cfin >= lo && cfin <= hi) { // uniform
#pragma unroll
					for (int i = 0; i < 4; ++i) {
						const uint32_t f = (uint32_t)(c0 + i == cfin) & (uint32_t)(hv[i] == tl - 1) & inm_bit(ql - tl, hv[i] - nmat[i], tl, ql);
						fin |= f;
						done_info = f ? (nmat[i] == 0 ? (int32_t)((tbw >> (8 * i)) & 7u) : 0) : done_info;
					}
				}
				*(int4*)at(newH, c0) = make_int4(hv[0], hv[1], hv[2], hv[3]);
				if (TB && c0 >= origin && c0 <= hi) *(uint32_t*)(M.tb + tb_used - origin + c0) = tbw;
				if (track_good) {
					unsigned long long *gword = M.good + (int64_t)newH * A.GW + g * 4;
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
				MWF_TICK(7);
			} else {
				refill(k);
#pragma unroll
				for (int i = 0; i < 4; ++i) ne1[i] = nf1[i] = ne2[i] = nf2[i] = kNegInf;
				retire();
			}
		}

		// Everything issued before this penalty must be complete before another wave may load it (vmcnt retires in issue
		// order).  This penalty issued at least 5*K memory operations (its loads are unconditional); letting the youngest 5*K
		// stay in flight therefore never leaves an older penalty's store pending, and keeps this penalty's own stores (and, in
		// the variant that prefetches across the barrier, the next penalty's loads) in flight when every H lag is >= 3.
		// Otherwise wait for everything.
#ifdef MWF_BAND_TIMING
		const unsigned long long t_c = __builtin_readcyclecounter();
#endif
		if (relaxed_stores) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(5 * K) : "memory");
		else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
		__builtin_amdgcn_s_barrier();
		asm volatile("" ::: "memory");
#ifdef MWF_BAND_TIMING
		{
			const unsigned long long t_d = __builtin_readcyclecounter();
			t_acc[0] += t_b - t_a, t_acc[1] += t_c - t_b, t_acc[2] += t_d - t_c;
			t_steps += 1, t_active += (unsigned long long)t_nact;
		}
#endif

		// ---- bookkeeping, identical on every thread
		if (uni(sh.flags[npar][0])) wf_lo = lo;
		if (uni(sh.flags[npar][1])) wf_hi = hi;
		const int32_t done = uni(sh.flags[npar][2]), payload = uni(sh.flags[npar][3]);
		s = s_new, curH = newH, par = npar, dcur = dnew, gl = gl_next;
		if (TB) tb_used += row_bytes;
		if ((s & 0xff) == 0) { // shrink (reference wf_stripe_shrink, miniwfa.c:144-171) on the interleaved good bits
			if (tid == 0) sh.red[0] = 0x7fffffff, sh.red[1] = -1;
			__syncthreads();
			const int32_t gfirst = wf_lo >> 8, n_words = ((wf_hi >> 8) - gfirst + 1) * 4;
			for (int32_t q = tid; q < n_words; q += T) {
				const int32_t gg = gfirst + (q >> 2), kq = q & 3, base = gg * kChunk;
				unsigned long long m = 0;
				for (int32_t j = 0; j < nH; ++j)
					if (sh.rng_lo[j] <= sh.rng_hi[j] && sh.rng_lo[j] <= base + kChunk - 1 && sh.rng_hi[j] >= base) m |= M.good[(int64_t)j * A.GW + gg * 4 + kq];
				m &= lane_mask(base, kq, wf_lo, wf_hi);
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
		if ((A.max_iter > 0 && cells > A.max_iter) || (A.max_s > 0 && s > A.max_s)) { // miniwfa.c:422-425
			R.status = ST_STOPPED;
			break;
		}
		if (done) {
			R.info = payload;
			break;
		}
	}
#ifdef MWF_BAND_TIMING
	if (lane == 0 && blockIdx.x == 0)
		printf("wave %2d steps %llu active-slots %llu | header %llu  slots %llu  wait+barrier %llu cycles (per step %.0f / %.0f / %.0f) | per active slot: load wait %.0f, long-run loop %.0f (entered in %.0f %% of slots), setup %.0f, recurrence+probe %.0f, tail %.0f\n", wave, t_steps, t_active,
		       t_acc[0], t_acc[1], t_acc[2], (double)t_acc[0] / t_steps, (double)t_acc[1] / t_steps, (double)t_acc[2] / t_steps,
		       (double)t_acc[3] / t_active, (double)t_acc[4] / t_active, 100.0 * t_pend / t_active, (double)t_acc[5] / t_active, (double)t_acc[6] / t_active, (double)t_acc[7] / t_active);
#endif
	R.s = s, R.cells = cells;
	return R;
}

// The packed kernels up to 512 threads are meant to share a CU — 2 x 512, 4 x 256, 8 x 128 or 16 x 64 threads: 4 waves per SIMD, i.e.
// at most 128 VGPRs; with traceback the two smaller ones get 168 (3 per SIMD), which they nearly fit.
template <int T, int K, int E1, int E2, bool TB, bool LSEQ, bool PACK>
__global__ __launch_bounds__(T, (PACK && T <= 512) ? ((TB && T < 512) ? 3 : 4) : 1) void wfa_band_kernel(const BatchArgs A)
{
	constexpr int NWK = (T / 64) * K, D = (E1 > E2 ? E1 : E2) + 1;
	__shared__ Shared sh;
	__shared__ int32_t edge[D][NWK][4];
	for (;;) {
		if (threadIdx.x == 0) sh.item = (int32_t)atomicAdd(A.queue, 1);
		__syncthreads();
		const int32_t item = uni(sh.item);
		__syncthreads();
		if (item >= A.n_pairs) break;
		const int32_t pair = A.order ? A.order[item] : item;
		PairMem M;
		pair_mem(A, (int32_t)blockIdx.x, pair, M);
		const uint8_t *lt = lds_dyn, *lq = lds_dyn;
		if (LSEQ) { // both sequences into LDS, each starting on a dword
			const int32_t qoff = ((M.tl + 3) & ~3) + 8;
			lq = lds_dyn + qoff;
			for (int32_t j = threadIdx.x; j < M.tl; j += T) lds_dyn[j] = M.ts[j];
			for (int32_t j = threadIdx.x; j < M.ql; j += T) lds_dyn[qoff + j] = M.qs[j];
			__syncthreads();
		}
		const bool trace = A.dbg && pair == A.debug_pair;
		const PassResult R = band_pass<T, K, E1, E2, TB, LSEQ, PACK>(A, M, sh, edge, lt, lq, 0, trace);
		finish_pair(A, M, (int32_t)blockIdx.x, pair, R, R.status, 0);
	}
}

template <int T, int K, int E1, int E2, bool PACK>
int launch_one(const BatchArgs &a, int grid, int lds, hipStream_t st)
{
	const bool tb = a.want_cigar != 0;
	if (lds > 48 * 1024) { // per device, possibly from several host threads: on every launch that needs it
		if (tb) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wfa_band_kernel<T, K, E1, E2, true, true, PACK>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
		else (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wfa_band_kernel<T, K, E1, E2, false, true, PACK>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
		(void)hipGetLastError();
	}
	if (lds > 0 && tb) hipLaunchKernelGGL((wfa_band_kernel<T, K, E1, E2, true, true, PACK>), dim3(grid), dim3(T), lds, st, a);
	else if (lds > 0) hipLaunchKernelGGL((wfa_band_kernel<T, K, E1, E2, false, true, PACK>), dim3(grid), dim3(T), lds, st, a);
	else if (tb) hipLaunchKernelGGL((wfa_band_kernel<T, K, E1, E2, true, false, PACK>), dim3(grid), dim3(T), 0, st, a);
	else hipLaunchKernelGGL((wfa_band_kernel<T, K, E1, E2, false, false, PACK>), dim3(grid), dim3(T), 0, st, a);
	return hipGetLastError() == hipSuccess ? 0 : -2;
}

template <int T, int K, int E1, int E2, bool PACK>
int occ_one(int lds, bool tb)
{
	int n = 0;
	hipError_t e;
	if (lds > 0 && tb) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, wfa_band_kernel<T, K, E1, E2, true, true, PACK>, T, lds);
	else if (lds > 0) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, wfa_band_kernel<T, K, E1, E2, false, true, PACK>, T, lds);
	else if (tb) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, wfa_band_kernel<T, K, E1, E2, true, false, PACK>, T, 0);
	else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, wfa_band_kernel<T, K, E1, E2, false, false, PACK>, T, 0);
	return e == hipSuccess ? n : 0;
}

} // namespace

bool band_supported(const Penalty &p)
{
	const bool inst = (p.e1 == 2 && p.e2 == 1) || (p.e1 == 2 && p.e2 == 2);
	// the prefetch reads H rows that are at least two penalties old
	return inst && p.x >= 2 && p.oe1 >= 2 && p.oe2 >= 2 && p.nH <= kMaxRing; // (the window table in LDS holds kMaxRing slices)
}

// (the int16-packed geometries live in mwf_band2.hip; here: long targets, whose offsets need 32 bits)
#define MWF_BAND_DISPATCH(FN, ...)                                                    \
	do {                                                                              \
		if (g.block == 768) {                                                         \
			if (a_e1 == 2 && a_e2 == 1) return FN<768, 2, 2, 1, false>(__VA_ARGS__);  \
			if (a_e1 == 2 && a_e2 == 2) return FN<768, 2, 2, 2, false>(__VA_ARGS__);  \
		} else if (g.block == 256) {                                                  \
			if (a_e1 == 2 && a_e2 == 1) return FN<256, 2, 2, 1, false>(__VA_ARGS__);  \
			if (a_e1 == 2 && a_e2 == 2) return FN<256, 2, 2, 2, false>(__VA_ARGS__);  \
		}                                                                             \
	} while (0)

int launch_band(const BatchArgs &a, int grid, const BandGeom &g, void *stream)
{
	const int a_e1 = a.pen.e1, a_e2 = a.pen.e2;
	MWF_BAND_DISPATCH(launch_one, a, grid, g.lds_bytes, (hipStream_t)stream);
	return -1;
}

int band_kernel_occupancy(const Penalty &p, const BandGeom &g, bool cigar)
{
	const int a_e1 = p.e1, a_e2 = p.e2;
	MWF_BAND_DISPATCH(occ_one, g.lds_bytes, cigar);
	return 0;
}

} // namespace mwf
