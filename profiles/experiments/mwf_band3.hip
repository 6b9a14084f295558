// mwf_band3.hip — the balanced band kernel: one workgroup per pair, ONE column per lane and pass, the E/F wavefronts in LDS.
//
// Same algorithm and bookkeeping as the other band kernels (reference loops miniwfa.c:261-308 and :212-226, driver
// :397-426, shrink :144-171); what changed is where the state lives and how a penalty's columns are dealt to the waves.
// The packed band kernel (mwf_band2.hip) keeps E/F in registers, which pins every 256-column chunk to one wave: the
// workgroup moves at the pace of the wave that holds two or three chunks while the others hold one, every wave drags a
// ~200-instruction per-penalty header plus the pack/unpack/shift/ageing code of register-resident 16-bit state along, and
// a wave issues only one instruction every ~5 cycles whatever it is (profiles/r02).  Here
//   * E1/F1/E2/F2 of the last e1 / e2 penalties live in LDS as int16 arrays indexed by column (a ring of `cap` columns,
//     updated in place: the slice of penalty s replaces the slice of penalty s-e, which is exactly the one it reads);
//   * the window is cut into 64-column chunks dealt round-robin to the waves (chunk g -> wave g mod waves), one column per
//     lane: every wave runs ceil or floor of (window / 64 / waves) passes of the SAME small loop body — no per-slot
//     register state, no unrolled copies, a header of a few dozen scalar instructions;
//   * the neighbours c-1 / c+1 are plain LDS / global reads at +-1 element; only a chunk's outer columns, which the
//     neighbouring WAVE may already have overwritten in place, go through a small table of edge values kept per penalty;
//   * sequences are held 2 bits per base (the pair must be plain ACGT — anything else is reported as ST_ALPHABET and the
//     host re-runs the pair on the byte-wise packed band kernel), so the first probe of the match extension compares
//     SIXTEEN bases with two ds_read2_b32, two v_alignbit and one xor.
// H rows stay in HBM/L2 as int16 (one 2-byte load per lane and source: hx, o1-, o1+, o2-, o2+).
// Results are bit-identical to the other kernels (tests/test_gpu_parity.py).
#include "mwf_device.h"

namespace mwf {

using namespace dev;

namespace {

extern __shared__ __attribute__((aligned(16))) uint8_t lds3[];


constexpr int32_t kDead16 = -32768;

// bookkeeping words behind the sequence copy
struct Band3Tab {
	Shared sh;
	int32_t bad;       // some byte of the pair is not A/C/G/T
	int32_t pad[3];
};

__device__ __forceinline__ uint32_t inm_bit(int32_t d, int32_t k, int32_t tl, int32_t ql)
{
	return (uint32_t)((uint32_t)(k + 1) < (uint32_t)(tl + 1)) & (uint32_t)((uint32_t)(d + k + 1) < (uint32_t)(ql + 1));
}

// LDS by byte offset (the dynamic allocation starts at LDS address 0: no base to add)
typedef __attribute__((address_space(3))) int16_t lds_i16_t;
typedef __attribute__((address_space(3))) uint32_t lds_u32_t;
__device__ __forceinline__ int32_t lds_i16(int32_t off) { return *(const lds_i16_t*)(uintptr_t)(uint32_t)off; }
__device__ __forceinline__ void lds_w16(int32_t off, int32_t v) { *(lds_i16_t*)(uintptr_t)(uint32_t)off = (int16_t)v; }
__device__ __forceinline__ void lds_w32(int32_t off, uint32_t v) { *(lds_u32_t*)(uintptr_t)(uint32_t)off = v; }

// sixteen bases starting at base j of the 2-bit copy that begins at LDS byte offset `base` (a multiple of 4)
__device__ __forceinline__ uint32_t seq16(int32_t base, int32_t j)
{
	const lds_u32_t *p = (const lds_u32_t*)(uintptr_t)(uint32_t)(base + ((j >> 4) << 2));
	return __builtin_amdgcn_alignbit(p[1], p[0], (uint32_t)j << 1);
}

// leading equal bases of a 16-base probe result (0 bits = equal), 16 when all are equal
__device__ __forceinline__ int32_t lead_eq2(uint32_t x)
{
	int32_t fb;
	asm("v_ffbl_b32 %0, %1" : "=v"(fb) : "v"(x)); // -1 for x == 0
	return (int32_t)min((uint32_t)fb >> 1, 16u);
}

// exact-match run t[j..] == q[i..], at most `room`, the first n0 known equal, walked by all 64 lanes: 1024 bases per trip.
// Arguments wave-uniform.
__device__ __forceinline__ int32_t run_wave3(int32_t qbase, int32_t j, int32_t i, int32_t room, int32_t n0)
{
	const int32_t lane = threadIdx.x & 63;
	int32_t n = n0;
	while (n < room) {
		const int32_t off = n + 16 * lane;
		int32_t m = 0;
		if (off < room) m = min(lead_eq2(seq16(0, j + off) ^ seq16(qbase, i + off)), room - off);
		const unsigned long long stop = __ballot(m < 16);
		if (stop == 0) { n += 1024; continue; }
		const int32_t first = (int32_t)__builtin_ctzll(stop);
		n += 16 * first + __builtin_amdgcn_readlane(m, first);
		break;
	}
	return min(n, room);
}

// bits of the 64-column word starting at column w0 that fall inside [lo,hi]
__device__ __forceinline__ unsigned long long word_mask(int32_t w0, int32_t lo, int32_t hi)
{
	if (hi < w0 || lo > w0 + 63 || lo > hi) return 0ull;
	unsigned long long m = ~0ull;
	if (lo > w0) m &= ~0ull << (lo - w0);
	if (hi < w0 + 63) m &= ~0ull >> (w0 + 63 - hi);
	return m;
}

// LDS byte offsets of one workgroup's regions
struct Lay3 {
	int32_t qbase;     // 2-bit copy of the query (the target's starts at 0)
	int32_t edge_e;    // [D][nch] dwords: {E1, E2} (int16 each) of every chunk's LAST column, per penalty mod D
	int32_t edge_f;    // [D][nch] dwords: {F1, F2} of every chunk's FIRST column
	int32_t st;        // [2*(E1+E2)][cap] int16: E1 slices, F1 slices, E2 slices, F2 slices
	int32_t cap;       // columns of the state ring (a multiple of 64)
	int32_t nch;       // cap / 64
};

template <int T, int E1, int E2, bool TB>
__device__ PassResult band3_pass(const BatchArgs &A, const PairMem &M, Shared &sh, const Lay3 &Y, bool trace_band)
{
	constexpr int NW = T / 64, D = (E1 > E2 ? E1 : E2) + 1;
	const int32_t tl = M.tl, ql = M.ql, cmax = tl + ql + 1;
	const int32_t tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6);
	const int32_t W = A.W, nH = A.pen.nH, lagx = A.pen.x, lag1 = A.pen.oe1, lag2 = A.pen.oe2;
	char *const Hb = (char*)M.H; // rows of W int16: (row, column) at byte (row * W + column) * 2
	const int32_t cap = Y.cap, nch = Y.nch, cap2 = cap * 2;
	const bool relaxed_stores = min(lagx, min(lag1, lag2)) >= 3;
	PassResult R;
	R.status = ST_OK, R.s = 0, R.info = 0, R.n_snap = 0, R.cells = 0;

	// ---- penalty 0 (reference wf_stripe_init, miniwfa.c:103-121) and its extension; every E/F slice starts dead
	{
		const int32_t words = (Y.st + 2 * (E1 + E2) * cap2 - Y.edge_e) >> 2; // edge tables and state arrays are contiguous
		uint32_t *p = (uint32_t*)(lds3 + Y.edge_e);
		for (int32_t j = tid; j < words; j += T) p[j] = 0x80008000u;
	}
	if (tid == 0) {
		for (int32_t j = 0; j < nH; ++j) sh.rng_lo[j] = 1, sh.rng_hi[j] = 0;
		for (int32_t j = 0; j < 12; ++j) (&sh.flags[0][0])[j] = 0;
		sh.rng_lo[0] = sh.rng_hi[0] = tl + 1;
	}
	if (tid < 64) { // the origin's run, walked by the first wave
		const int32_t k0 = run_wave3(Y.qbase, 0, 0, min(tl, ql), 0) - 1;
		if (tid == 0) {
			*(int16_t*)(Hb + ((size_t)(uint32_t)(tl + 1) << 1)) = (int16_t)k0;
			sh.word[1] = k0;
		}
	}
	__syncthreads();
	{
		const int32_t k0 = uni(sh.word[1]);
		if (k0 == tl - 1 && k0 == ql - 1) return R;
	}

	int32_t s = 0, wf_lo = tl + 1, wf_hi = tl + 1;
	int32_t curH = 0, par = 0, dcur = 0;
	int32_t p1 = 0, p2 = 0;                  // s mod E1, s mod E2 of the penalty being computed: the state slices it reads and replaces
	int64_t cells = 0, tb_used = 0;
	// ring position (in chunks) of chunk ga = lo >> 6, kept incrementally (a real modulo only after a shrink)
	int32_t ga_prev = tl >> 6, ra = (tl >> 6) % nch;

	const int32_t cfin = ql + 1; // the end cell (tl-1, ql-1) lies on diagonal ql-tl, i.e. in this column
	for (;;) {
#ifdef MWF_B3_TIMING
		const uint64_t tm0 = __builtin_readcyclecounter();
#endif
		const int32_t lo = wf_lo > 1 ? wf_lo - 1 : 1;       // miniwfa.c:417-418
		const int32_t hi = wf_hi < cmax ? wf_hi + 1 : cmax;
		const int32_t s_new = s + 1;
		const int32_t newH = curH + 1 == nH ? 0 : curH + 1;
		const int32_t npar = par + 1 == 3 ? 0 : par + 1;
		const int32_t dnew = dcur + 1 == D ? 0 : dcur + 1;
		p1 = p1 + 1 == E1 ? 0 : p1 + 1;
		p2 = p2 + 1 == E2 ? 0 : p2 + 1;
		const int32_t origin = lo & ~3;
		const int32_t row_bytes = (hi | 3) - origin + 1;
		if (TB) {
			if (s_new - 1 >= A.rows_slot) { R.status = ST_ROWS_OVERFLOW; break; }
			if (tb_used + row_bytes > A.tb_slot_bytes) { R.status = ST_TB_OVERFLOW; break; }
		}
		// every column the slices still alive can hold must have a ring position of its own
		if (((hi < cmax ? hi + 1 : cmax) >> 6) - ((lo > 1 ? lo - 1 : 1) >> 6) + 1 > nch - 1) { R.status = ST_BAND_OVERFLOW; break; }
		int32_t jx = newH - lagx; if (jx < 0) jx += nH;
		int32_t j1 = newH - lag1; if (j1 < 0) j1 += nH;
		int32_t j2 = newH - lag2; if (j2 < 0) j2 += nH;
		int32_t k1 = newH - E1; if (k1 < 0) k1 += nH; // H slots of the penalties whose E/F slices are read (for their windows)
		int32_t k2 = newH - E2; if (k2 < 0) k2 += nH;
		const bool track_good = (((256 - (s_new & 255)) & 255) < nH); // a shrink can still see this slice
		int32_t d1 = dnew - E1; if (d1 < 0) d1 += D; // ages of the edge table to read: penalties s_new-E1 and s_new-E2
		int32_t d2 = dnew - E2; if (d2 < 0) d2 += D;
		const int32_t ga = lo >> 6, gb = hi >> 6;
		if (ga != ga_prev) { // the window's first chunk moved: by one to the left, or anywhere after a shrink
			if (ga == ga_prev - 1) ra = ra == 0 ? nch - 1 : ra - 1;
			else ra = ga % nch;
			ga_prev = ga;
		}
		// columns nH+1 inside the window can read nothing outside any source window (an edge moves outwards by at most one
		// column per penalty; after a shrink the older slices are the wider ones)
		const int32_t dlo = lo + nH + 1, dhi = hi - nH - 1;

		if (tid == 0) {
			sh.rng_lo[newH] = lo, sh.rng_hi[newH] = hi;
			sh.flags[npar + 1 == 3 ? 0 : npar + 1][0] = 0; // the flag word of the NEXT penalty (its last readers passed the previous barrier)
			if (TB) M.row_off[s_new - 1] = tb_used, M.row_lo[s_new - 1] = origin;
#ifndef MWF_B3_TIMING // (the timing build keeps per-phase cycle counts in the trace buffer instead)
			if (trace_band && s_new - 1 < A.dbg_cap) M.dbg[2 * (s_new - 1)] = lo, M.dbg[2 * (s_new - 1) + 1] = hi;
#endif
		}

		const char *const rowx = Hb + ((size_t)(uint32_t)(jx * W) << 1);
		const char *const row1 = Hb + ((size_t)(uint32_t)(j1 * W) << 1);
		const char *const row2 = Hb + ((size_t)(uint32_t)(j2 * W) << 1);
		char *const rown = Hb + ((size_t)(uint32_t)(newH * W) << 1);
		// state slices of this penalty (read at c-1 / c+1, replaced at c) and edge rows
		const int32_t sE1 = Y.st + p1 * cap2, sF1 = Y.st + (E1 + p1) * cap2, sE2 = Y.st + (2 * E1 + p2) * cap2, sF2 = Y.st + (2 * E1 + E2 + p2) * cap2;
		const int32_t eE1 = Y.edge_e + d1 * nch * 4, eE2 = Y.edge_e + d2 * nch * 4 + 2, eF1 = Y.edge_f + d1 * nch * 4, eF2 = Y.edge_f + d2 * nch * 4 + 2;
		const int32_t eEn = Y.edge_e + dnew * nch * 4, eFn = Y.edge_f + dnew * nch * 4;

		// window history: only a chunk near a window edge needs it
		int32_t xlo = 1, xhi = 0, alo = 1, ahi = 0, blo = 1, bhi = 0, ulo = 1, uhi = 0, vlo = 1, vhi = 0;
		bool hist = false;
		// lane-constant parts of the state addresses: E is read at column c-1 and replaced at c, F read at c+1 and replaced at c
		const int32_t vE1 = sE1 + 2 * lane - 2, vE2 = sE2 + 2 * lane - 2, vF1 = sF1 + 2 * lane, vF2 = sF2 + 2 * lane;

		// One chunk's inputs
		struct In { int32_t hx, o1m, o1p, o2m, o2p, g1m, g2m, g1p, g2p, g, rg; };
		auto fetch = [&](In &x, int32_t g, int32_t rg) {
			const uint32_t co = (uint32_t)((g << 6) + lane) << 1;
			const uint32_t cm = max(co, 2u); // (column 0 is never inside a window: keep its left neighbour's address inside the row)
			x.hx = *(const int16_t*)(rowx + co);
			x.o1m = *(const int16_t*)(row1 + cm - 2), x.o1p = *(const int16_t*)(row1 + co + 2);
			x.o2m = *(const int16_t*)(row2 + cm - 2), x.o2p = *(const int16_t*)(row2 + co + 2);
			// E of column c-1 and F of column c+1, e1 (e2) penalties ago: lane 0 / lane 63 take them from the edge table
			const int32_t rl = rg == 0 ? nch - 1 : rg - 1, rr = rg + 1 == nch ? 0 : rg + 1;
			const int32_t so = rg << 7;
			x.g1m = lds_i16(lane == 0 ? eE1 + 4 * rl : vE1 + so), x.g2m = lds_i16(lane == 0 ? eE2 + 4 * rl : vE2 + so);
			x.g1p = lds_i16(lane == 63 ? eF1 + 4 * rr : vF1 + so + 2), x.g2p = lds_i16(lane == 63 ? eF2 + 4 * rr : vF2 + so + 2);
			x.g = g, x.rg = rg;
		};
		auto process = [&](In &x) {
			const int32_t g = x.g, cb = g << 6, c = cb + lane;
			const bool deep = cb >= dlo && cb + 63 <= dhi && !track_good; // uniform
			if (!deep) {
				if (!hist) {
					xlo = uni(sh.rng_lo[jx]), xhi = uni(sh.rng_hi[jx]);
					alo = uni(sh.rng_lo[j1]), ahi = uni(sh.rng_hi[j1]);
					blo = uni(sh.rng_lo[j2]), bhi = uni(sh.rng_hi[j2]);
					ulo = uni(sh.rng_lo[k1]), uhi = uni(sh.rng_hi[k1]);
					vlo = uni(sh.rng_lo[k2]), vhi = uni(sh.rng_hi[k2]);
					hist = true;
				}
				// reads outside a source window yield "dead" (what the reference's pads supply, miniwfa.c:96-99)
				x.hx = (uint32_t)(c - xlo) <= (uint32_t)(xhi - xlo) && xlo <= xhi ? x.hx : kDead16;
				x.o1m = (uint32_t)(c - 1 - alo) <= (uint32_t)(ahi - alo) && alo <= ahi ? x.o1m : kDead16;
				x.o1p = (uint32_t)(c + 1 - alo) <= (uint32_t)(ahi - alo) && alo <= ahi ? x.o1p : kDead16;
				x.o2m = (uint32_t)(c - 1 - blo) <= (uint32_t)(bhi - blo) && blo <= bhi ? x.o2m : kDead16;
				x.o2p = (uint32_t)(c + 1 - blo) <= (uint32_t)(bhi - blo) && blo <= bhi ? x.o2p : kDead16;
				x.g1m = (uint32_t)(c - 1 - ulo) <= (uint32_t)(uhi - ulo) && ulo <= uhi ? x.g1m : kDead16;
				x.g1p = (uint32_t)(c + 1 - ulo) <= (uint32_t)(uhi - ulo) && ulo <= uhi ? x.g1p : kDead16;
				x.g2m = (uint32_t)(c - 1 - vlo) <= (uint32_t)(vhi - vlo) && vlo <= vhi ? x.g2m : kDead16;
				x.g2p = (uint32_t)(c + 1 - vlo) <= (uint32_t)(vhi - vlo) && vlo <= vhi ? x.g2p : kDead16;
			}
			const Cell v = wf_cell<TB>(x.hx, x.o1m, x.g1m, x.o2m, x.g2m, x.o1p, x.g1p, x.o2p, x.g2p);
			int32_t ne1 = v.e1, nf1 = v.f1, ne2 = v.e2, nf2 = v.f2, hq = v.h;
			uint32_t live = 0, gbit = 0;
			if (!deep) { // uniform: mask the columns outside the window, note edge liveness and the good bits
				const uint32_t a = (uint32_t)((c >= lo) & (c <= hi));
				ne1 = a ? v.e1 : kDead16, nf1 = a ? v.f1 : kDead16, ne2 = a ? v.e2 : kDead16, nf2 = a ? v.f2 : kDead16;
				hq = a ? v.h : kDead16;
				if (track_good) { // uniform
					const int32_t d = c - 1 - tl;
					gbit = a & (inm_bit(d, v.h, tl, ql) | inm_bit(d, v.e1, tl, ql) | inm_bit(d, v.f1, tl, ql) | inm_bit(d, v.e2, tl, ql) | inm_bit(d, v.f2, tl, ql));
				}
				// edge rule (miniwfa.c:325-326): H is the max of the five, so "any live" == "H live"
				const uint32_t lv = (uint32_t)(v.h >= -1);
				live = (lv & (uint32_t)(c == lo)) | ((lv & (uint32_t)(c == hi)) << 1);
			}
			// ---- match extension (miniwfa.c:212-226): j = k+1 clamped to the largest j inside the matrix, so that dead and
			// phantom offsets have no room; the first probe compares sixteen bases
			const int32_t rj = min(tl, cmax - c);                                  // min(tl, ql - d)
			const int32_t j = (int32_t)min((uint32_t)(hq + 1), (uint32_t)rj);
			const int32_t i = j + c - 1 - tl;                                      // query index d + j
			const int32_t room = rj - j;
			const uint32_t pt = seq16(0, j), pq = seq16(Y.qbase, i);
			// the new E/F are final: replace the slices in place, publish the chunk's outer columns for the neighbouring waves
			const int32_t so = x.rg << 7;
			lds_w16(vE1 + so + 2, ne1), lds_w16(vF1 + so, nf1), lds_w16(vE2 + so + 2, ne2), lds_w16(vF2 + so, nf2);
			if (lane == 0 || lane == 63) {
				const uint32_t pe = (uint32_t)(ne1 & 0xffff) | (uint32_t)ne2 << 16, pf = (uint32_t)(nf1 & 0xffff) | (uint32_t)nf2 << 16;
				lds_w32((lane == 0 ? eFn : eEn) + 4 * x.rg, lane == 0 ? pf : pe);
			}
			int32_t n = min(lead_eq2(pt ^ pq), room);
			// a run of sixteen matches continues (the cells near the alignment path): each lane walks its own run, four trips at
			// most; what is still open then the whole wave walks, 1024 bases per trip
			if (__ballot(n == 16 && room > 16)) {
				uint32_t open = 0;
				if (n == 16 && room > 16) {
					for (int trip = 0; n < room; ++trip) {
						if (trip == 4) { open = 1; break; }
						const int32_t m = lead_eq2(seq16(0, j + n) ^ seq16(Y.qbase, i + n));
						n += m;
						if (m < 16) break;
					}
					n = min(n, room);
				}
				for (unsigned long long owners = __ballot(open != 0); owners; owners &= owners - 1) {
					const int32_t src = (int32_t)__builtin_ctzll(owners);
					const int32_t js = __builtin_amdgcn_readlane(j, src), is = __builtin_amdgcn_readlane(i, src), rs = __builtin_amdgcn_readlane(room, src);
					const int32_t nn = run_wave3(Y.qbase, js, is, rs, 80);
					n = lane == src ? nn : n;
				}
			}
			const int32_t hv = v.h + n;
			// ---- termination test of the extension sweep (miniwfa.c:405-409): only column ql+1 can hold the end cell
			if ((uint32_t)(cfin - cb) < 64u && cfin >= lo && cfin <= hi) { // uniform
				const uint32_t f = (uint32_t)(c == cfin) & (uint32_t)(hv == tl - 1) & inm_bit(ql - tl, v.h, tl, ql);
				const unsigned long long fm = __ballot(f != 0);
				if (fm) {
					const int32_t info = f ? (n == 0 ? (int32_t)(v.tb & 7u) : 0) : 0;
					const uint32_t bits = 4u | (uint32_t)__builtin_amdgcn_readlane(info, (int32_t)__builtin_ctzll(fm)) << 4;
					if (lane == 0) atomicOr((unsigned int*)&sh.flags[npar][0], bits);
				}
			}
			*(int16_t*)(rown + ((uint32_t)c << 1)) = (int16_t)hv;
			if (TB && c >= lo && c <= hi) M.tb[tb_used - origin + c] = (uint8_t)v.tb;
			if (track_good) {
				const unsigned long long m = __ballot(gbit != 0);
				if (lane == 0) M.good[(int64_t)newH * A.GW + g] = m;
			}
			if (!deep) { // uniform
				const uint32_t bits = (__ballot(live & 1u) ? 1u : 0u) | (__ballot(live & 2u) ? 2u : 0u);
				if (bits && lane == 0) atomicOr((unsigned int*)&sh.flags[npar][0], bits);
			}
		};

		// this wave's chunks: the first at or after ga, then every NW-th.  (Measured and dropped: two chunks per iteration with the
		// next two requested a whole iteration ahead — 113 VGPRs, no faster: a wave's time goes into issuing its ~16 memory
		// instructions per chunk)
#ifdef MWF_B3_TIMING
		const uint64_t tm1 = __builtin_readcyclecounter();
#endif
		const int32_t skip = (NW & (NW - 1)) == 0 ? ((wave - ga) & (NW - 1)) : ((wave - ga) % NW + NW) % NW;
		int32_t g = ga + skip, rg = ra + skip;
		if (rg >= nch) rg -= nch;
		for (; g <= gb; g += NW) {
			In x;
			fetch(x, g, rg);
			process(x);
			rg += NW;
			if (rg >= nch) rg -= nch;
		}

		// everything this penalty wrote must be complete before another wave may load it
#ifdef MWF_B3_TIMING
		const uint64_t tm2 = __builtin_readcyclecounter();
#endif
		// With every lag >= 3 the rows written now are first loaded two penalties from now: the youngest store may stay in
		// flight across the barrier (vmcnt retires in issue order; the next penalty's wait covers it).
		if (relaxed_stores && !TB && !track_good) asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
		else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#ifdef MWF_B3_TIMING
		const uint64_t tm3 = __builtin_readcyclecounter();
#endif
		__builtin_amdgcn_s_barrier();
		asm volatile("" ::: "memory");

		// ---- bookkeeping, identical on every thread
		const uint32_t fl = (uint32_t)uni(sh.flags[npar][0]);
#ifdef MWF_B3_TIMING
		if (trace_band && tid == (A.max_s < 0 ? -A.max_s : 0) && s_new - 1 < A.dbg_cap) { // cycles: header+first requests | chunks, drain | barrier+flags
			const uint64_t tm4 = __builtin_readcyclecounter();
			M.dbg[2 * (s_new - 1)] = (int32_t)(min((uint32_t)(tm1 - tm0), 65535u) | min((uint32_t)(tm2 - tm1), 65535u) << 16);
			M.dbg[2 * (s_new - 1) + 1] = (int32_t)(min((uint32_t)(tm3 - tm2), 65535u) | min((uint32_t)(tm4 - tm3), 65535u) << 16);

		}
#endif
		if (fl & 1u) wf_lo = lo;
		if (fl & 2u) wf_hi = hi;
		const int32_t done = (int32_t)((fl >> 2) & 1u), payload = (int32_t)((fl >> 4) & 7u);
		s = s_new, curH = newH, par = npar, dcur = dnew;
		if (TB) tb_used += row_bytes;
		if ((s & 0xff) == 0) { // shrink (reference wf_stripe_shrink, miniwfa.c:144-171) on the good bits
			if (tid == 0) sh.red[0] = 0x7fffffff, sh.red[1] = -1;
			__syncthreads();
			const int32_t gfirst = wf_lo >> 6, n_words = (wf_hi >> 6) - gfirst + 1;
			for (int32_t q = tid; q < n_words; q += T) {
				const int32_t gg = gfirst + q, base = gg << 6;
				unsigned long long m = 0;
				for (int32_t jj = 0; jj < nH; ++jj)
					if (sh.rng_lo[jj] <= sh.rng_hi[jj] && sh.rng_lo[jj] <= base + 63 && sh.rng_hi[jj] >= base) m |= M.good[(int64_t)jj * A.GW + gg];
				m &= word_mask(base, wf_lo, wf_hi);
				if (m) {
					atomicMin(&sh.red[0], base + (int32_t)__builtin_ctzll(m));
					atomicMax(&sh.red[1], base + 63 - (int32_t)__builtin_clzll(m));
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
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	R.s = s, R.cells = cells;
	return R;
}

// Bytes -> 2 bits per base into LDS at `base` (dwords of sixteen bases, two dwords of slack behind the last base).
// Returns nonzero when a byte is not one of A, C, G, T.  code = (byte >> 1) & 3: A 0, C 1, T 2, G 3.
template <int T>
__device__ __forceinline__ uint32_t pack2bit(const uint8_t *src, int32_t len, int32_t base)
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
				// the byte each code stands for: 0 'A' 0x41, 1 'C' 0x43, 2 'T' 0x54, 3 'G' 0x47
				const uint32_t lo1 = code & 0x01010101u, hi1 = (code >> 1) & 0x01010101u;
				const uint32_t expect = 0x41414141u + (lo1 & ~hi1) * 0x02u + (hi1 & ~lo1) * 0x13u + (hi1 & lo1) * 0x06u;
				bad |= x ^ expect;
				out |= ((code | code >> 6 | code >> 12 | code >> 18) & 0xffu) << (8 * k);
			}
		} else {
			for (int32_t k = 0; k < 16 && b0 + k < len; ++k) {
				const uint32_t x = src[b0 + k], code = (x >> 1) & 3u;
				bad |= x ^ ((0x47544341u >> (8 * code)) & 0xffu);
				out |= code << (2 * k);
			}
		}
		*(uint32_t*)(lds3 + base + 4 * w) = out;
	}
	return bad;
}

// Two 512-thread workgroups share a CU (LDS: ~80 KB each): 4 waves per SIMD.
template <int T, int E1, int E2, bool TB>
__global__ __launch_bounds__(T, T / 128) void wfa_band3_kernel(const BatchArgs A)
{
	constexpr int D = (E1 > E2 ? E1 : E2) + 1;
	Band3Tab *const L = (Band3Tab*)(lds3 + A.band_lds_seq);
	Shared &sh = L->sh;
	Lay3 Y;
	Y.cap = A.band3_cap, Y.nch = Y.cap >> 6;
	Y.edge_e = A.band_lds_seq + (int32_t)sizeof(Band3Tab);
	Y.edge_f = Y.edge_e + D * Y.nch * 4;
	Y.st = Y.edge_f + D * Y.nch * 4;
	for (;;) {
		if (threadIdx.x == 0) sh.item = (int32_t)atomicAdd(A.queue, 1), L->bad = 0;
		__syncthreads();
		const int32_t item = uni(sh.item);
		__syncthreads();
		if (item >= A.n_pairs) break;
		const int32_t pair = A.order ? A.order[item] : item;
		PairMem M;
		pair_mem(A, (int32_t)blockIdx.x, pair, M);
		Y.qbase = ((M.tl >> 4) + 2) * 4;
		uint32_t bad = pack2bit<T>(M.ts, M.tl, 0);
		bad |= pack2bit<T>(M.qs, M.ql, Y.qbase);
		if (__ballot(bad != 0) && (threadIdx.x & 63) == 0) L->bad = 1;
		__syncthreads();
		PassResult R;
		if (uni(L->bad)) R.status = ST_ALPHABET, R.s = 0, R.info = 0, R.n_snap = 0, R.cells = 0;
		else R = band3_pass<T, E1, E2, TB>(A, M, sh, Y, A.dbg && pair == A.debug_pair);
		finish_pair(A, M, (int32_t)blockIdx.x, pair, R, R.status, 0);
	}
}

template <int E1, int E2>
int lds_total(int lds_seq, int cap) { return lds_seq + (int)sizeof(Band3Tab) + 2 * ((E1 > E2 ? E1 : E2) + 1) * (cap / 64) * 4 + 2 * (E1 + E2) * cap * 2; }

template <int T, int E1, int E2>
int launch_one(const BatchArgs &a, int grid, int lds, hipStream_t st)
{
	if (a.want_cigar) {
		static int max_set = 0;
		if (lds > 48 * 1024 && lds > max_set) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wfa_band3_kernel<T, E1, E2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds), max_set = lds;
		hipLaunchKernelGGL((wfa_band3_kernel<T, E1, E2, true>), dim3(grid), dim3(T), lds, st, a);
	} else {
		static int max_set = 0;
		if (lds > 48 * 1024 && lds > max_set) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wfa_band3_kernel<T, E1, E2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds), max_set = lds;
		hipLaunchKernelGGL((wfa_band3_kernel<T, E1, E2, false>), dim3(grid), dim3(T), lds, st, a);
	}
	return hipGetLastError() == hipSuccess ? 0 : -2;
}

template <int T, int E1, int E2>
int occ_one(int lds, bool tb)
{
	int n = 0;
	hipError_t e;
	if (tb) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, wfa_band3_kernel<T, E1, E2, true>, T, lds);
	else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, wfa_band3_kernel<T, E1, E2, false>, T, lds);
	return e == hipSuccess ? n : 0;
}

} // namespace

bool band3_supported(const Penalty &p)
{
	return (p.e1 == 2 && p.e2 == 1) || (p.e1 == 2 && p.e2 == 2) || (p.e1 == 1 && p.e2 == 1);
}

// LDS bytes of a workgroup that holds `cap` columns of state and `lds_seq` bytes of 2-bit sequences
int band3_lds_bytes(const Penalty &p, int lds_seq, int cap)
{
	if (p.e1 == 2 && p.e2 == 1) return lds_total<2, 1>(lds_seq, cap);
	if (p.e1 == 2 && p.e2 == 2) return lds_total<2, 2>(lds_seq, cap);
	return lds_total<1, 1>(lds_seq, cap);
}

#define MWF_BAND3_PEN(FN, T, ...)                                               \
	{                                                                           \
		if (a_e1 == 2 && a_e2 == 1) return FN<T, 2, 1>(__VA_ARGS__);            \
		if (a_e1 == 2 && a_e2 == 2) return FN<T, 2, 2>(__VA_ARGS__);            \
		if (a_e1 == 1 && a_e2 == 1) return FN<T, 1, 1>(__VA_ARGS__);            \
	}
#define MWF_BAND3_DISPATCH(FN, ...)                                             \
	do {                                                                        \
		if (g.block == 512) MWF_BAND3_PEN(FN, 512, __VA_ARGS__)                 \
		if (g.block == 768) MWF_BAND3_PEN(FN, 768, __VA_ARGS__)                 \
		if (g.block == 1024) MWF_BAND3_PEN(FN, 1024, __VA_ARGS__)               \
	} while (0)

int launch_band3(const BatchArgs &a0, int grid, const BandGeom &g, void *stream)
{
	const int a_e1 = a0.pen.e1, a_e2 = a0.pen.e2;
	BatchArgs a = a0;
	a.band_lds_seq = g.lds_bytes;
	a.band3_cap = g.span;
	const int lds = band3_lds_bytes(a0.pen, g.lds_bytes, g.span);
	MWF_BAND3_DISPATCH(launch_one, a, grid, lds, (hipStream_t)stream);
	return -1;
}

int band3_kernel_occupancy(const Penalty &p, const BandGeom &g, bool cigar)
{
	const int a_e1 = p.e1, a_e2 = p.e2;
	const int lds = band3_lds_bytes(p, g.lds_bytes, g.span);
	MWF_BAND3_DISPATCH(occ_one, lds, cigar);
	return 0;
}

} // namespace mwf
