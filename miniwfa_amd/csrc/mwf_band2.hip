// mwf_band2.hip — the packed band kernel: the fast path for batches of pairs whose offsets fit 16 bits (targets below
// ~32 kb: BASELINE configs[2], read-length batches, chain-mode gap fills).
//
// One workgroup per pair, a wave owns 256-column chunks (four columns per lane, K chunk slots per wave), E/F wavefronts in
// registers, one barrier per penalty; reference loops miniwfa.c:261-308 and :212-226, driver :397-426.  What makes it fast
// (DESIGN.md section 4.2 has the measurements):
//   * two columns per instruction: a lane's columns c0..c3 live in two registers (c0,c2) and (c1,c3) — the 16-bit H rows in HBM
//     hold a quad in that order — and the recurrence, the validity / room arithmetic of the match extension, the good bits and the
//     traceback byte are v_pk_* instructions on those pairs; neighbouring columns come from one DPP shift + one v_alignbit;
//   * rows read as DEAD beyond the window they were computed for (columns outside the window are stored dead, the chunk on either
//     side of the window is stored dead every penalty), so no read is ever masked and no window history is consulted;
//   * the sequence copy (2 bits per base, or bytes) starts at LDS address 0; the first probe of the match extension looks at 16
//     bases (8 bytes), its eight LDS reads are issued together; longer runs are walked per lane, then by the whole wave;
//   * kernel arguments are read from the kernarg segment where they are used (no SGPR spills from holding them for ever), the edge
//     table between neighbouring waves is addressed by immediates and written without exec masks, register histories are aged by
//     v_swap, the three per-penalty flags travel as one LDS word, rows are loaded only for chunks that are active.
// Results are bit-identical to the other kernels (tests/test_gpu_parity.py).
#include <cstddef>
#include <type_traits>
#include "mwf_device.h"

namespace mwf {

using namespace dev;

namespace {

extern __shared__ __attribute__((aligned(16))) uint8_t lds2[];

// Geometry constants (round 6: plain constants — rounds 3-5 overrode them with -D for experiments whose results are in profiles/HISTORY.md and the comments here;
// the experiment switches MWF_B2_XPREF / MWF_B2_MERGE_OFF / MWF_B2_NOPRIO and their dead branches are gone, see profiles/experiments/).
constexpr int kWideT = 512, kWideK = 3; // threads x chunk slots per wave of the widest two-per-CU geometry (24 chunks): 512 x 3; measured: 256 x 6 equal, 384 x 4 slower
constexpr int kWaves768 = 1;            // waves per SIMD the 768-thread geometry is compiled for (one workgroup per CU, up to 168 VGPRs; two per CU at 80 VGPRs: slower)
// chunk slots per wave of the 1024-thread geometry (16 waves, one workgroup per CU): 80 chunks = windows of up to 20 224 columns.  Measured on
// 1250 x 50 kb @ 3 % (windows of up to 16 900 columns): 4 slots 172 ms + 24 pairs re-run on the generic kernel = 240 ms, 5 slots (4 spilled VGPRs)
// 174.6 ms and no re-run, 6 slots (12 spilled) 177.0; 256 x 50 kb: 36.2 (+ 9 re-runs: 104) / 38.7 / 39.6 ms; generic kernel 279 and 78 ms.
constexpr int kSpanK = 5;
// Chunk slots per wave of the 512-thread geometry's copies on biased offsets (class 14 of mwf_plan.cpp: pairs of ~11-21 kb, two per CU instead of the span
// geometry's one).  Measured (ms per align; span geometry | 4 | 5 | 6 slots): 1024 x 12 kb @ 5 % 34.9 | 24.6 | 24.8 | 27.7, 1024 x 15 kb @ 4 % 35.6 | - | 25.1 | 25.7,
// 1024 x 17 kb @ 3 % 29.5 | - | 20.2 | 20.5, 512 x 18 kb @ 5 % 32.1 | - | - | 25.3, 1024 x 20 kb @ 3 % 37.3 | - | - | 26.9: five slots (40 chunks, 4 spilled VGPRs)
// while target + query stay below 3.5 of their span, six (48 chunks) up to 3.5 of theirs.
constexpr int kW4K = 5;
constexpr int kSpanT = 1024;  // threads of the span geometry (measured: 768 x 7 slots — twelve waves, 168 VGPRs, no spills — 187.6 against 178.7 ms on 1250 x 50 kb, 768 x 6 189.6)
constexpr bool is_span(int T, int K) { return T == kSpanT && K == kSpanK; }
constexpr int kWideWaves = 4; // waves per SIMD the widest geometry is compiled for (4: 128 VGPRs, two 512-thread workgroups per CU)
constexpr int kChunk = 256;
constexpr int kFoldMaxLag = 8; // largest o1 + e1 the folded form of the packed kernel is launched for
constexpr int32_t kDeadPair = (int32_t)0x80008000u;

__device__ __forceinline__ int32_t from_left(int32_t v, int32_t fill) { return __builtin_amdgcn_update_dpp(fill, v, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ int32_t from_right(int32_t v, int32_t fill) { return __builtin_amdgcn_update_dpp(fill, v, 0x130, 0xf, 0xf, false); }
__device__ __forceinline__ int32_t lo16(int32_t v) { return (int32_t)(int16_t)(v & 0xffff); }
__device__ __forceinline__ int32_t hi16(int32_t v) { return v >> 16; }
__device__ __forceinline__ int32_t pack2(int32_t a, int32_t b)
{
	typedef short short2_t __attribute__((ext_vector_type(2)));
	const short2_t v = __builtin_amdgcn_cvt_pk_i16(a, b); // saturating
	return __builtin_bit_cast(int32_t, v);
}
__device__ __forceinline__ int32_t both16(int32_t v) { return (int32_t)(((uint32_t)v << 16) | ((uint32_t)v & 0xffffu)); }

// four bytes at byte offset `off` of the sequence copy (which starts at LDS offset 0)
__device__ __forceinline__ uint32_t seq4(int32_t off)
{
	const uint32_t *p = (const uint32_t*)(lds2 + (off & ~3));
	return __builtin_amdgcn_alignbyte(p[1], p[0], (uint32_t)off);
}

// leading equal bytes of a 4-byte probe result (0 bits = equal bytes): v_ffbl_b32 yields -1 for x == 0
__device__ __forceinline__ int32_t lead_eq(uint32_t x)
{
	int32_t fb;
	asm("v_ffbl_b32 %0, %1" : "=v"(fb) : "v"(x));
	return (int32_t)((uint32_t)fb >> 3);
}

// The recurrence with its traceback byte (dev::wf_cell, miniwfa.c:267-278, :289-306) written for few live registers.
template <bool WANT_TB>
__device__ __forceinline__ Cell cell16(int32_t hx, int32_t o1m, int32_t g1m, int32_t o2m, int32_t g2m,
                                       int32_t o1p, int32_t g1p, int32_t o2p, int32_t g2p)
{
	if (!WANT_TB) return wf_cell<false>(hx, o1m, g1m, o2m, g2m, o1p, g1p, o2p, g2p);
	Cell c;
	c.e1 = max(o1m, g1m);
	c.e2 = max(o2m, g2m);
	c.f1 = max(o1p, g1p) + 1;
	c.f2 = max(o2p, g2p) + 1;
	const int32_t e = max(c.e1, c.e2), f = max(c.f1, c.f2), g = max(e, f), m = hx + 1;
	c.h = max(m, g);
	// The byte from the RESULTS (so that the gap sources are read once and can stay 16-bit operands): H is the maximum of
	// m, e1, e2, f1, f2 and the reference's tie-breaking (mismatch, then E1, E2, F1, F2) is the first of them that equals it;
	// a gap state was extended iff it differs from what opening it would have given.
	const uint32_t z = c.h == m ? 0u : c.h == c.e1 ? 1u : c.h == c.e2 ? 3u : c.h == c.f1 ? 2u : 4u;
	c.tb = z | ((uint32_t)(c.e1 != o1m) << 3) | ((uint32_t)(c.f1 != o1p + 1) << 4) | ((uint32_t)(c.e2 != o2m) << 5) | ((uint32_t)(c.f2 != o2p + 1) << 6);
	(void)e, (void)f, (void)g;
	return c;
}

// Leading equal bytes of t[j..] and q[..] (aq: byte offset of the query base), at most eight looked at: three dwords of
// each sequence, two v_alignbyte's each.  Returns min(equal bytes, 9 if all eight are equal).
struct Probe8 { uint32_t t0, t1, t2, q0, q1, q2; };
__device__ __forceinline__ void probe8_issue(Probe8 &p, int32_t j, int32_t aq)
{
	const uint32_t *pt = (const uint32_t*)(lds2 + (j & ~3)), *pq = (const uint32_t*)(lds2 + (aq & ~3));
	p.t0 = pt[0], p.t1 = pt[1], p.t2 = pt[2], p.q0 = pq[0], p.q1 = pq[1], p.q2 = pq[2];
}
__device__ __forceinline__ int32_t probe8_count(const Probe8 &p, int32_t j, int32_t aq)
{
	const uint32_t x0 = __builtin_amdgcn_alignbyte(p.t1, p.t0, (uint32_t)j) ^ __builtin_amdgcn_alignbyte(p.q1, p.q0, (uint32_t)aq);
	const uint32_t x1 = __builtin_amdgcn_alignbyte(p.t2, p.t1, (uint32_t)j) ^ __builtin_amdgcn_alignbyte(p.q2, p.q1, (uint32_t)aq);
	return min(min(lead_eq(x0), lead_eq(x1) + 4), 9); // lead_eq is huge for "no difference"
}

// ---- 2-bit sequence copy (pairs of plain A/C/G/T): sixteen bases per dword, base j at bits 2*(j & 15) of dword j >> 4.
// One ds_read2_b32 and one v_alignbit per sequence give SIXTEEN bases from any position: the first probe of the match
// extension needs two LDS instructions instead of four (a wave issues one LDS instruction every 15-40 cycles,
// profiles/r02/lds_issue_rates_microbench.txt) and a run must be twice as long before the per-lane loop is entered.
__device__ __forceinline__ uint32_t seq16(int32_t base, int32_t j)
{
	const uint32_t *p = (const uint32_t*)(lds2 + base + ((j >> 4) << 2));
	return __builtin_amdgcn_alignbit(p[1], p[0], (uint32_t)j << 1);
}
// leading equal bases of a 16-base probe result (0 bits = equal); huge for "no difference"
__device__ __forceinline__ int32_t lead_eq2(uint32_t x)
{
	int32_t fb;
	asm("v_ffbl_b32 %0, %1" : "=v"(fb) : "v"(x));
	return (int32_t)((uint32_t)fb >> 1);
}
struct Probe16 { uint32_t t0, t1, q0, q1; };
__device__ __forceinline__ void probe16_issue(Probe16 &p, int32_t qbase, int32_t j, int32_t iq)
{
	const uint32_t *pt = (const uint32_t*)(lds2 + ((j >> 4) << 2)), *pq = (const uint32_t*)(lds2 + qbase + ((iq >> 4) << 2));
	p.t0 = pt[0], p.t1 = pt[1], p.q0 = pq[0], p.q1 = pq[1];
}
// min(equal bases, 17 if all sixteen are equal)
__device__ __forceinline__ int32_t probe16_count(const Probe16 &p, int32_t j, int32_t iq)
{
	return min(lead_eq2(__builtin_amdgcn_alignbit(p.t1, p.t0, (uint32_t)j << 1) ^ __builtin_amdgcn_alignbit(p.q1, p.q0, (uint32_t)iq << 1)), 17);
}
// exact-match run t[j..] == q[iq..] on the 2-bit copies, at most `room`, the first n0 known equal, walked by all 64 lanes:
// 1024 bases per trip.  Arguments wave-uniform.
__device__ __forceinline__ int32_t run_wave16(int32_t qbase, int32_t j, int32_t iq, int32_t room, int32_t n0)
{
	const int32_t lane = threadIdx.x & 63;
	int32_t n = n0;
	while (n < room) {
		const int32_t off = n + 16 * lane;
		int32_t m = 0;
		if (off < room) m = min(min(lead_eq2(seq16(0, j + off) ^ seq16(qbase, iq + off)), 16), room - off);
		const unsigned long long stop = __ballot(m < 16);
		if (stop == 0) { n += 1024; continue; }
		const int32_t first = (int32_t)__builtin_ctzll(stop);
		n += 16 * first + __builtin_amdgcn_readlane(m, first);
		break;
	}
	return min(n, room);
}
// Bytes -> 2 bits per base into LDS at `base` (two dwords of slack behind the last base).  Returns nonzero when a byte is not
// one of A, C, G, T.  code = (byte >> 1) & 3: A 0, C 1, T 2, G 3.
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
#pragma unroll 1
			for (int32_t k = 0; k < 16 && b0 + k < len; ++k) {
				const uint32_t x = src[b0 + k], code = (x >> 1) & 3u;
				bad |= x ^ ((0x47544341u >> (8 * code)) & 0xffu);
				out |= code << (2 * k);
			}
		}
		*(uint32_t*)(lds2 + base + 4 * w) = out;
	}
	return bad;
}

__device__ __forceinline__ uint32_t inm_bit(int32_t d, int32_t k, int32_t tl, int32_t ql)
{
	return (uint32_t)((uint32_t)(k + 1) < (uint32_t)(tl + 1)) & (uint32_t)((uint32_t)(d + k + 1) < (uint32_t)(ql + 1));
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

// exact-match run t[j..] == q[..] (aq = byte offset of the query base in LDS), at most `room`, the first n0 known equal,
// walked by all 64 lanes: 256 bytes per trip.  Arguments wave-uniform.
__device__ __forceinline__ int32_t run_wave2(int32_t j, int32_t aq, int32_t room, int32_t n0)
{
	const int32_t lane = threadIdx.x & 63;
	int32_t n = n0;
	while (n < room) {
		const int32_t off = n + 4 * lane;
		int32_t m = 0;
		if (off < room) {
			const uint32_t x = seq4(j + off) ^ seq4(aq + off);
			m = min(min(lead_eq(x), 4), room - off);
		}
		const unsigned long long stop = __ballot(m < 4);
		if (stop == 0) { n += 256; continue; }
		const int32_t first = (int32_t)__builtin_ctzll(stop);
		n += 4 * first + __builtin_amdgcn_readlane(m, first);
		break;
	}
	return min(n, room);
}

// ---- biased offsets (the 1024-thread geometry: targets of up to ~60 kb).  A 16-bit half holds  offset - B  with B = wide_bias(tl):
// live offsets are >= -1, i.e. halves >= -1 - B, and everything below is dead.  Every place that reads an offset as a number adds B
// back (the "+ 1" of j = k + 1 becomes "+ 1 + B": no instruction more); as unsigned 16-bit numbers the true values (up to 65 535)
// fit where signed ones would not.  Two things move towards the ends of the 16 bits by at most one per penalty and are CHECKED at
// every penalty that is a multiple of 256 (in the good-bit code of that penalty), with a margin of kBiasMargin > 2 x 256 + nH (what was
// computed since the last check but one is still being read from the H ring):
//   * dead values start at -32768 and a chain of them — F2 of the columns next to the lower window edge, which moves with the chain — gains
//     one per penalty: no value may lie in [-1 - B - kBiasMargin, -1 - B);
//   * offsets that ran past the end of the target (F of cells beyond the matrix keeps counting): no H above 32767 - kBiasMargin, i.e. more
//     than kBiasOver - kBiasMargin beyond the target's end.
// A pair that fails a check is handed back (ST_BAND_OVERFLOW: generic kernel).  With tl = 50 000 the dead side holds ~12 900 penalties.
constexpr int32_t kBiasOver = 1500, kBiasMargin = 600;
__device__ __forceinline__ int32_t wide_bias(int32_t tl) { return max(tl + kBiasOver - 32767, 0); }

template <int D, int NWK>
struct alignas(16) Band2Lds {
	Shared sh;
	// per age and chunk slot r (entry r + 1): {E1, E2 of columns (c1,c3) of lane 63 | F1, F2 of columns (c0,c2) of lane 0}; entry 0 mirrors
	// slot NWK-1 and entry NWK+1 slot 0, so that a slot finds its neighbours at fixed distances from its own entry
	int32_t edge[D][NWK + 2][4];
	int32_t dump[64 * 2 + (2 * NWK + 4) * 4]; // where the lanes that do not hold an outer column put theirs (no exec-mask games around the stores)
};

// ---- packed 16-bit arithmetic: two columns per register, every operation one VOP3P instruction
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
#define MWF_BC(T, v) __builtin_bit_cast(T, v)
__device__ __forceinline__ int32_t pk_max(int32_t a, int32_t b) { return MWF_BC(int32_t, __builtin_elementwise_max(MWF_BC(s16x2, a), MWF_BC(s16x2, b))); }
__device__ __forceinline__ int32_t pk_add(int32_t a, int32_t b) { return MWF_BC(int32_t, (s16x2)(MWF_BC(s16x2, a) + MWF_BC(s16x2, b))); }
__device__ __forceinline__ int32_t pk_sub(int32_t a, int32_t b) { return MWF_BC(int32_t, (s16x2)(MWF_BC(s16x2, a) - MWF_BC(s16x2, b))); }
__device__ __forceinline__ int32_t pk_minu(int32_t a, int32_t b) { return MWF_BC(int32_t, __builtin_elementwise_min(MWF_BC(u16x2, a), MWF_BC(u16x2, b))); }
// max(a - b, 0) on unsigned halves (v_pk_sub_u16 clamp): zero iff a <= b
__device__ __forceinline__ int32_t pk_subsat(int32_t a, int32_t b) { return MWF_BC(int32_t, __builtin_elementwise_sub_sat(MWF_BC(u16x2, a), MWF_BC(u16x2, b))); }
// 0xffff in every half of x that is not zero (asm: the compiler turns the min into compares, selects and a v_perm)
__device__ __forceinline__ int32_t pk_nonzero_mask(int32_t x)
{
	int32_t m;
	asm("v_pk_min_u16 %0, %1, 1 op_sel_hi:[1,0]\n\tv_pk_sub_i16 %0, 0, %0 op_sel_hi:[0,1]" : "=&v"(m) : "v"(x));
	return m;
}
// 1 in every half where a != b
__device__ __forceinline__ int32_t pk_ne1(int32_t a, int32_t b)
{
	int32_t m;
	asm("v_xor_b32 %0, %1, %2\n\tv_pk_min_u16 %0, %0, 1 op_sel_hi:[1,0]" : "=&v"(m) : "v"(a), "v"(b));
	return m;
}
// a * b + c on unsigned halves
__device__ __forceinline__ int32_t pk_mad(int32_t a, int32_t b, int32_t c)
{
	int32_t m;
	asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(m) : "v"(a), "v"(b), "v"(c));
	return m;
}
__device__ __forceinline__ int32_t bfi(int32_t mask, int32_t a, int32_t b) { return (a & mask) | (b & ~mask); } // mask ? a : b, bitwise
__device__ __forceinline__ int32_t half_of(int32_t v, int hi) { return hi ? v >> 16 : (int32_t)(int16_t)(v & 0xffff); }
__device__ __forceinline__ int32_t pair_of(int32_t lo, int32_t hi) { return (int32_t)(((uint32_t)hi << 16) | ((uint32_t)lo & 0xffffu)); }

// A lane's four columns c0..c3 of a chunk live in two registers, A = (c0, c2) and B = (c1, c3): the columns to the left of A's
// are (c-1, c1) — B shifted in from the lane to the left — and those to the left of B's are A itself; to the right of A's: B, to
// the right of B's: (c2, c4).  `fill` supplies what lane 0 (lane 63) takes from beyond the chunk.
__device__ __forceinline__ int32_t left_of_A(int32_t B, int32_t fill) { return __builtin_amdgcn_alignbit(B, from_left(B, fill), 16); }
__device__ __forceinline__ int32_t right_of_B(int32_t A, int32_t fill) { return __builtin_amdgcn_alignbit(from_right(A, fill), A, 16); }

template <int T, int K, int E1, int E2, bool TB, bool S2, bool BI4, bool FOLD, typename ArgsT>
__device__ PassResult band2_pass(const ArgsT &A, const PairMem &M, Shared &sh, const int32_t edge_base, const int32_t qoff, bool trace_band)
{
	constexpr int NW = T / 64, NWK = NW * K, D = (E1 > E2 ? E1 : E2) + 1;
	constexpr bool BI = is_span(T, K) || BI4; // biased offsets with range checks (wide_bias): the span geometry and the four-slot 512-thread one's copy for long pairs
	constexpr int FULL = S2 ? 16 : 8; // bases the first probe of the match extension looks at
	// FOLD (score-only, gap-open lag - mismatch lag == e1, i.e. o1 == x as in the default penalties): the row a penalty reads for its
	// mismatch term, H[s-x], IS the row the first gap piece opens from e1 penalties later (miniwfa.c:267-278: both E1 and F1 take
	// max(H[s-o1-e1], E1/F1[s-e1]) of a neighbouring column).  The registers that hold E1/F1 of the last e1 penalties therefore hold
	// max(E1, H[s-x]) (max(F1, H[s-x])) instead — the maximum is what the reference computes e1 penalties later — and the kernel does
	// not load the row of lag o1+e1 at all.  A chunk that left the window keeps running until the rows it computed have been folded
	// and aged out: kAgeOut penalties (the host asks for this form only when o1 + e1 <= kFoldMaxLag).
	// With traceback the byte must tell an opened first gap piece from an extended one (miniwfa.c:289-306) — the comparison the fold no
	// longer makes at the cell itself.  It makes it e1 penalties EARLIER, in the neighbouring column, where both terms are at hand:
	// E1[s] > H[s-x] there means "the E1 of penalty s+e1 one column on is an extension".  Bits 3 and 4 of a folded byte say that
	// (PairMem::tb_fwd), and the walk reads them from the cell an extension would come from (traceback_wave: one more byte per gap run).
	constexpr int kAgeOut = FOLD ? kFoldMaxLag : D;
	constexpr int kAge = (NWK + 2) * 16; // bytes of one age of the edge table
	static_assert(D == 2 || D == 3, "edge-table ages");
	const int32_t tl = M.tl, ql = M.ql, cmax = tl + ql + 1;
	const int32_t tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6);
	const int32_t W = A.W, nH = A.pen.nH, lagx = A.pen.x, lag1 = A.pen.oe1, lag2 = A.pen.oe2;
	const int32_t B = BI ? wide_bias(tl) : 0;                   // a half holds offset - B
	const int32_t ONEB = BI ? both16(1 + B) : 0x00010001;       // k -> j = k + 1 as a true (unsigned) number
	// H rows: W int16 per row, a quad of columns 4q..4q+3 stored as (c0, c2, c1, c3) — the two registers of a lane, one 8-byte load;
	// 8 bytes of slack in front (lane 0 of chunk 0 looks one quad to the left; offsets are unsigned: the slack is part of `lane8`)
	char *const Hb = (char*)M.H;
	const uint32_t RS = (uint32_t)W << 1; // bytes per row
	const int32_t min_lag = min(lagx, min(lag1, lag2));
	const bool relaxed_stores = min_lag >= 3; // rows written now are first loaded two penalties from now: stores may cross the barrier
	PassResult R;
	R.status = ST_OK, R.s = 0, R.info = 0, R.n_snap = 0, R.cells = 0;

	// per-thread wavefront state, two columns per register: [age - 1][slot][A / B]
	static_assert((E1 == 1 || E1 == 2) && (E2 == 1 || E2 == 2), "history depth");
	int32_t e1h[E1][K][2], f1h[E1][K][2], e2h[E2][K][2], f2h[E2][K][2];
#pragma unroll
	for (int k = 0; k < K; ++k)
#pragma unroll
		for (int i = 0; i < 2; ++i) {
#pragma unroll
			for (int a = 0; a < E1; ++a) e1h[a][k][i] = f1h[a][k][i] = kDeadPair;
#pragma unroll
			for (int a = 0; a < E2; ++a) e2h[a][k][i] = f2h[a][k][i] = kDeadPair;
		}
	// lane constants: local columns of A and B, byte offset of the lane's quad, of the neighbouring word it fetches
	const int32_t RA = pair_of(4 * lane, 4 * lane + 2); // (RB = RA + 1, RB1 = RA + 2 per half are recomputed where needed: registers are scarcer than adds)
	const uint32_t lane8 = ((uint32_t)lane << 3) + 8u;
	const int32_t nd = lane == 0 ? -4 : 8; // lane 0: B of the quad to the left; lane 63: A of the quad to the right
	const int32_t T0 = both16(cmax), TLp = both16(tl), TL1 = both16(tl + 1);
	// edge-table addresses: a slot's entry sits (k NW + 1) entries behind the wave's base.  Every lane stores its outer columns —
	// lane 63 (E) and lane 0 (F) into the table, the others into a dump — so that no store needs an exec mask.
	const int32_t ebase = edge_base + wave * 16, dump_base = edge_base + D * kAge + NW * 16; // (the mirror store reaches NW-1 entries back)
	const int32_t dump_lane = dump_base + lane * 8;
	int32_t epos[D]; // byte offset of the table of age a + 1 (penalty s_new - a - 1); the oldest is overwritten and becomes age 1
#pragma unroll
	for (int a = 0; a < D; ++a) epos[a] = a * kAge;

	// ---- every row read before it is written must read as dead around the origin: a slot that no penalty has written yet is only read
	// during the first nH - 1 penalties, whose windows (and the columns next to them) stay within nH + 1 columns of the origin
	{
		const int32_t reach = nH + 1 + 8, g_a = max(tl + 1 - reach, 0) >> 8, n_g = ((tl + 1 + reach) >> 8) - g_a + 1, per_row = n_g * 64;
		for (int32_t q = tid; q < nH * per_row; q += T) {
			const int32_t row = q / per_row, rem = q - row * per_row;
			*(int2*)(Hb + (size_t)((uint32_t)row * RS + (uint32_t)(g_a * 512 + rem * 8 + 8))) = make_int2(kDeadPair, kDeadPair);
		}
	}
	// ---- penalty 0 (reference wf_stripe_init, miniwfa.c:103-121) and its extension
	for (int32_t j = tid; j < D * (NWK + 2) * 4; j += T) *(int32_t*)(lds2 + edge_base + 4 * j) = kDeadPair;
	if (tid == 0) {
		for (int32_t j = 0; j < nH; ++j) sh.rng_lo[j] = 1, sh.rng_hi[j] = 0;
		for (int32_t j = 0; j < 12; ++j) (&sh.flags[0][0])[j] = 0;
		sh.rng_lo[0] = sh.rng_hi[0] = tl + 1;
		sh.word[3] = -1; // furthest offset seen at a forecast penalty (dev::window_forecast)
	}
	__syncthreads(); // (orders the dead rows before the origin's store)
	if (tid < 64) { // the origin's run, walked by the first wave
		const int32_t k0 = (S2 ? run_wave16(qoff, 0, 0, min(tl, ql), 0) : run_wave2(0, qoff, min(tl, ql), 0)) - 1;
		if (tid == 0) {
			const int32_t c = tl + 1, e = c & 3;
			*(int16_t*)(Hb + 8 + (size_t)(uint32_t)(((c & ~3) + ((e & 1) << 1) + (e >> 1)) << 1)) = (int16_t)(k0 - B);
			sh.word[1] = k0;
		}
	}
	__syncthreads();
	{
		const int32_t k0 = uni(sh.word[1]);
		if (k0 == tl - 1 && k0 == ql - 1) return R;
	}

	int32_t s = 0, wf_lo = tl + 1, wf_hi = tl + 1;
	int32_t curH = 0, par = 0;
	const uint32_t ring_bytes = (uint32_t)nH * RS;
	// rows of the coming penalty and of its three lags, as byte offsets that advance by one row per penalty (penalty 1 first)
	uint32_t bn = (uint32_t)(1 % nH) * RS, bx = (uint32_t)((nH - lagx + 1) % nH) * RS, b1 = (uint32_t)((nH - lag1 + 1) % nH) * RS, b2 = (uint32_t)((nH - lag2 + 1) % nH) * RS;
	// The rows of ONE chunk: H at the three lags (one 8-byte load each) and the word next to the chunk for the two gap-open rows.
	// (Requesting the coming penalty's rows of the wave's first chunk before or straight behind the barrier was measured in rounds 3-5: no gain, -3 % on the
	// folded form — the row loads are not what the critical wave waits for; the variants are in profiles/experiments/.)
	struct Rows { int2 HX, O1, O2; int32_t N1, N2; } pre;
	pre.HX = pre.O1 = pre.O2 = make_int2(0, 0), pre.N1 = pre.N2 = 0;
	auto load_rows = [&](Rows &r, const char *rx, const char *r1, const char *r2, uint32_t off) {
		const uint32_t noff = off + (uint32_t)nd;
		r.HX = *(const int2*)(rx + off), r.O2 = *(const int2*)(r2 + off);
		r.N2 = *(const int32_t*)(r2 + noff);
		if (!FOLD) r.O1 = *(const int2*)(r1 + off), r.N1 = *(const int32_t*)(r1 + noff);
	};
	int64_t cells = 0, tb_used = 0;
	int32_t est_window = 0;
	const int64_t iter_limit = A.max_iter > 0 ? A.max_iter : INT64_MAX;
	const int32_t s_limit = A.max_s > 0 ? A.max_s : INT32_MAX;
	const int64_t rows_slot = TB ? A.rows_slot : 0, tb_slot_bytes = TB ? A.tb_slot_bytes : 0;

	// chunk of every slot of this wave under the mapping that starts at chunk gl (changes only when gl does)
	int32_t gl = (wf_lo > 1 ? wf_lo - 1 : 1) >> 8, gk[K];
	auto remap = [&](int32_t g_lo) {
		const int32_t base = g_lo - g_lo % NWK;
#pragma unroll
		for (int k = 0; k < K; ++k) {
			int32_t g = base + wave + NW * k;
			if (g < g_lo) g += NWK;
			gk[k] = g;
		}
	};
	remap(gl);
	int32_t idle[K]; // penalties since the slot last held an active chunk (registers and edge-table entries start dead)
#pragma unroll
	for (int k = 0; k < K; ++k) idle[k] = kAgeOut;
	int32_t up_wait = 0, up_min = 0; // penalties for which the window has started above chunk gl, the lowest start among them

	// One penalty; returns true when the pass ends.  A depth-2 history is two registers, [0] the newer: the penalty reads [1] for the last
	// time, overwrites it, and the two trade places (v_swap_b32) — no copies, and a slot that is skipped leaves its registers alone.
	auto step = [&]() __attribute__((always_inline)) -> bool {
		constexpr int P1 = E1 - 1, P2 = E2 - 1;
#ifdef MWF_B2_TIMING
		const uint64_t tm0 = __builtin_readcyclecounter();
#endif
		const int32_t lo = wf_lo > 1 ? wf_lo - 1 : 1;       // miniwfa.c:417-418
		const int32_t hi = wf_hi < cmax ? wf_hi + 1 : cmax;
		const int32_t lo_p = lo, hi_p = hi;
		const int32_t s_new = s + 1;
		const int32_t newH = curH + 1 == nH ? 0 : curH + 1;
		const int32_t npar = par + 1 == 3 ? 0 : par + 1;
		const int32_t origin = lo & ~3;
		const int32_t row_bytes = (hi | 3) - origin + 1;
		if (TB) {
			if (s_new - 1 >= rows_slot) { R.status = ST_ROWS_OVERFLOW; return true; }
			if (tb_used + row_bytes > tb_slot_bytes) { R.status = ST_TB_OVERFLOW; return true; }
		}
		// the window of penalty s_new+1 lies inside [lo-1, hi+1] whatever the flags say; it must fit the register span
		const int32_t gl_next = (lo > 1 ? lo - 1 : 1) >> 8;
		// (the mapping follows a window that moves UP kAgeOut penalties late — the chunks it leaves behind still run, see below)
		if ((hi >> 8) - min(lo >> 8, gl + 1) + 3 > NWK - 1) // (only then can the exact test fail)
			if (((hi < cmax ? hi + 1 : cmax) >> 8) - min(gl_next, gl) + 1 > NWK - 1) { R.status = ST_BAND_OVERFLOW; return true; }
		// (the four-slot form of the 512-thread geometry notes whether the three-slot form would have held the pair: the host's choice for the next align)
		if (T == 512 && K == 4 && (hi >> 8) - (lo >> 8) + 3 > 23)
			if (((hi < cmax ? hi + 1 : cmax) >> 8) - gl_next + 1 > 23) R.n_snap = 1;
		// (biased offsets, sequences beyond 32 kb: the room arithmetic holds ql - d in 16 unsigned bits)
		if (BI && cmax - lo > 65535) { R.status = ST_BAND_OVERFLOW; return true; }
		const char *const rowx = Hb + bx, *const row1 = Hb + b1, *const row2 = Hb + b2;
		char *const rown = Hb + bn;
		const bool track_good = (((256 - (s_new & 255)) & 255) < nH); // a shrink can still see this slice
		// edge table: the ages to read (penalties s_new-E1 and s_new-E2) and the one to overwrite, as LDS addresses
		const int32_t rE1 = ebase + epos[E1 - 1], rE2 = ebase + epos[E2 - 1];
		const int32_t wE = lane == 63 ? ebase + epos[D - 1] : dump_lane, wF = lane == 0 ? ebase + epos[D - 1] : dump_lane;
		auto put_edge = [&](int k, int32_t e1b, int32_t e2b, int32_t f1a, int32_t f2a) {
			*(int2*)(lds2 + wE + (k * NW + 1) * 16) = make_int2(e1b, e2b);
			*(int2*)(lds2 + wF + (k * NW + 1) * 16 + 8) = make_int2(f1a, f2a);
			if (k == K - 1 && wave == NW - 1) *(int2*)(lds2 + wE - (NW - 1) * 16) = make_int2(e1b, e2b);  // slot NWK-1 is slot 0's left neighbour
			if (k == 0 && wave == 0) *(int2*)(lds2 + wF + (NWK + 1) * 16 + 8) = make_int2(f1a, f2a);       // slot 0 is slot NWK-1's right neighbour
		};
		const int32_t ga = lo >> 8, gb = hi >> 8, gspan = gb - ga; // chunks [ga, gb] meet the window
		// (not in the widest geometry: what it hands back goes to the generic kernel, several times slower — it would only do so at penalty 1024
		// and beyond 1.5 x its span, and the bookkeeping costs the headline kernel 21 more spilled SGPRs)
		// (the span geometry forecasts later: what it hands back goes to the generic kernel, twice as slow — no more)
		// (none in the four-slot 512-thread geometry either — measured: a tight one at penalty 1024 costs its batches 6 % in spilled SGPRs and hands back pairs that
		// are then re-run alone, 1024 x 12 kb @ 5 % 34.2 against 24.7 ms)
		const bool forecast = NWK < 24 ? (s_new == 64 || s_new == 256 || s_new == 1024) : NWK >= 64 ? (s_new == 1024 || s_new == 4096) : false; // uniform: look at how far the pair has come (dev::window_forecast)
		int32_t far = kDeadPair;
		const int32_t cfin = ql + 1; // the end cell (tl-1, ql-1) lies on diagonal ql-tl, i.e. in this column

		if (wave == 0) { // (the whole wave stores the same words: no exec mask to set up)
			sh.rng_lo[newH] = lo, sh.rng_hi[newH] = hi;
			sh.flags[npar + 1 == 3 ? 0 : npar + 1][0] = 0; // the flag word of the NEXT penalty (its last readers passed the previous barrier)
			if (TB) M.row_off[s_new - 1] = tb_used, M.row_lo[s_new - 1] = origin;
#ifndef MWF_B2_TIMING // (the timing build keeps per-phase cycle counts in the trace buffer instead)
			if (trace_band && s_new - 1 < fresh(A).dbg_cap) M.dbg[2 * (s_new - 1)] = lo, M.dbg[2 * (s_new - 1) + 1] = hi;
#endif
		}

		bool act[K];
#pragma unroll
		for (int k = 0; k < K; ++k) act[k] = (uint32_t)(gk[k] - ga) <= (uint32_t)gspan;
		// the waves with the most chunks to do set the pace of the penalty: let them issue first.  Chunks are dealt round-robin from
		// chunk ga on: the wave at distance p from it holds ceil((n - p) / NW) of the n active chunks.
		bool busy;
		if ((NW & (NW - 1)) == 0) busy = ((wave - ga) & (NW - 1)) < gspan + 1 - NW;
		else busy = (int)act[0] + (int)act[1] + (K > 2 ? (int)act[K - 1] : 0) >= 2;
		// One priority level per active chunk of the wave (0 ... 3), kept through the barrier and the next header: 18.4 -> 17.7 ms on
		// 1024 x 10 kb against "two or more chunks: 3, else 0"; stepping it down as chunks complete 19.1, other maps 17.8 ... 18.3, one more
		// level for a wave that walked a long run at the last penalty: no change.
		if ((NW & (NW - 1)) == 0) {
			// this wave holds ceil(left / NW) chunks; the waves that hold the window's first or last chunk (masks, liveness) count one more (17.7 -> 17.45 ms)
			const int32_t pos = (wave - ga) & (NW - 1);
			const int32_t left = gspan + 1 - pos + ((pos == 0 || ((gb - wave) & (NW - 1)) == 0) ? NW : 0);
			if (left > 2 * NW) __builtin_amdgcn_s_setprio(3);
			else if (left > NW) __builtin_amdgcn_s_setprio(2);
			else if (left > 0) __builtin_amdgcn_s_setprio(1);
			else __builtin_amdgcn_s_setprio(0);
			(void)busy;
		} else {
			const int n_busy = (int)act[0] + (int)act[1] + (K > 2 ? (int)act[K - 1] : 0);
			if (n_busy >= 3) __builtin_amdgcn_s_setprio(3);
			else if (n_busy == 2) __builtin_amdgcn_s_setprio(2);
			else if (n_busy == 1) __builtin_amdgcn_s_setprio(1);
			else __builtin_amdgcn_s_setprio(0);
			(void)busy;
		}
#ifdef MWF_B2_TIMING
		int n_act = 0;
#pragma unroll
		for (int k = 0; k < K; ++k) n_act += act[k] ? 1 : 0;
#endif

#ifdef MWF_B2_TIMING
		const uint64_t tm1 = __builtin_readcyclecounter();
#endif
		int n_stores = 0;
		// ---- every row must read as dead next to the chunks it was computed for (a later window reaches at most nH + 1 columns
		// beyond this one: the reference's pads, miniwfa.c:96-99); the waves next to the window's ends hold the fewest chunks.  These
		// stores go first: the store that may stay in flight across the barrier (relaxed_stores) is then a chunk's own.
		if (ga >= 1 && wave == (ga - 1) % NW) *(int2*)(rown + ((uint32_t)((ga - 1) << 9) + lane8)) = make_int2(kDeadPair, kDeadPair);
		if (wave == (gb + 1) % NW) *(int2*)(rown + ((uint32_t)((gb + 1) << 9) + lane8)) = make_int2(kDeadPair, kDeadPair);
#if MWF_B2_TIMING == 2
		bool chunk_timed = false;
		uint32_t ct[4] = {0, 0, 0, 0};
#endif
#pragma unroll
		for (int k = 0; k < K; ++k) {
			// a chunk that left the window: its columns are not computed any more, i.e. their E/F are dead.  It runs through the ordinary
			// code D more times (every column outside the window: masked dead — rare, a window edge crosses a chunk boundary inwards only
			// at a shrink), which ages the slot's registers and edge-table entries; after that there is nothing left to do.
			if (!act[k]) {
				if (idle[k] >= kAgeOut) continue; // uniform
				++idle[k];
			} else idle[k] = 0;
			// (the window bounds are laundered per slot: what the chunk code derives from them stays inside this branch instead of being
			// computed up front by waves that hold no active chunk)
			int32_t lo = uni(lo_p), hi = uni(hi_p);
			asm volatile("" : "+s"(lo), "+s"(hi));
			const int32_t g = gk[k], cb = g * kChunk, c0 = cb + 4 * lane;
			// ---- rows: H at the three lags (one 8-byte load each) and the word next to the chunk for the two gap-open rows
#if MWF_B2_TIMING == 2
			uint64_t tc0 = 0, tc1 = 0, tc2 = 0, tc3 = 0, tc4 = 0;
			const bool timed = !chunk_timed;
			if (timed) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tc0) :: "memory");
#endif
			const uint32_t off = (uint32_t)(cb << 1) + lane8;
			// (measured and dropped, round 4: the rows of EVERY chunk the wave will run requested at the top of the penalty — 16 more VGPRs — 19.4 against 17.35 ms)
			load_rows(pre, rowx, row1, row2, off);
			const int2 HX = pre.HX, O1 = pre.O1, O2 = pre.O2;
			const int32_t N1 = pre.N1, N2 = pre.N2;
			// gap-extension sources: E of column c-1, F of column c+1, e1 (e2) penalties ago; lane 0 / lane 63 take the neighbouring
			// slot's outer columns from the edge table
			const int32_t xE1 = *(const int32_t*)(lds2 + rE1 + k * NW * 16), xE2 = *(const int32_t*)(lds2 + rE2 + k * NW * 16 + 4);
			const int32_t xF1 = *(const int32_t*)(lds2 + rE1 + (k * NW + 2) * 16 + 8), xF2 = *(const int32_t*)(lds2 + rE2 + (k * NW + 2) * 16 + 12);
			const bool inside = cb >= lo && cb + kChunk - 1 <= hi; // uniform: every column of the chunk belongs to the window

			// ---- recurrence (dev::wf_cell, miniwfa.c:267-278) on pairs of columns
			const int32_t E1a = e1h[P1][k][0], E1b = e1h[P1][k][1], F1a = f1h[P1][k][0], F1b = f1h[P1][k][1];
			const int32_t E2a = e2h[P2][k][0], E2b = e2h[P2][k][1], F2a = f2h[P2][k][0], F2b = f2h[P2][k][1];
			const int32_t o1mA = FOLD ? 0 : left_of_A(O1.y, N1), o2mA = left_of_A(O2.y, N2), g1mA = left_of_A(E1b, xE1), g2mA = left_of_A(E2b, xE2);
			const int32_t o1pB = FOLD ? 0 : right_of_B(O1.x, N1), o2pB = right_of_B(O2.x, N2), g1pB = right_of_B(F1a, xF1), g2pB = right_of_B(F2a, xF2);
			const int32_t ONE = 0x00010001;
			// (FOLD: the E1 / F1 registers already hold the maximum with the row the gap opens from)
			int32_t ne1A = FOLD ? g1mA : pk_max(o1mA, g1mA), ne2A = pk_max(o2mA, g2mA);
			int32_t ne1B = FOLD ? E1a : pk_max(O1.x, E1a), ne2B = pk_max(O2.x, E2a);
			const int32_t pf1A = FOLD ? F1b : pk_max(O1.y, F1b), pf2A = pk_max(O2.y, F2b); // F before its + 1
			const int32_t pf1B = FOLD ? g1pB : pk_max(o1pB, g1pB), pf2B = pk_max(o2pB, g2pB);
			int32_t nf1A = pk_add(pf1A, ONE), nf2A = pk_add(pf2A, ONE), nf1B = pk_add(pf1B, ONE), nf2B = pk_add(pf2B, ONE);
			const int32_t mA = pk_add(HX.x, ONE), mB = pk_add(HX.y, ONE);
			int32_t hA = pk_max(pk_max(mA, pk_max(ne1A, ne2A)), pk_max(nf1A, nf2A));
			int32_t hB = pk_max(pk_max(mB, pk_max(ne1B, ne2B)), pk_max(nf1B, nf2B));
#if MWF_B2_TIMING == 2
			if (timed) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tc1) : "v"(hA), "v"(hB) : "memory");
#endif
			uint32_t tbw = 0;
			if (TB) {
				// The byte from the RESULTS (miniwfa.c:289-306): H is the maximum of m, e1, e2, f1, f2 and the reference's tie-breaking
				// (mismatch, then E1, E2, F1, F2 = codes 0, 1, 3, 2, 4) is the first of them that equals it; a gap state was extended iff it
				// differs from what opening it would have given.  n* = 1 where different: z = nm (1 + ne1 (2 + ne2 (2 nf1 - 1))).
				const int32_t TWO = 0x00020002, NEG1 = (int32_t)0xffffffffu, EIGHT = 0x00080008, C16 = 0x00100010, C32 = 0x00200020, C64 = 0x00400040;
				int32_t zA = pk_mad(pk_ne1(hA, nf1A), TWO, NEG1), zB = pk_mad(pk_ne1(hB, nf1B), TWO, NEG1);
				zA = pk_mad(pk_ne1(hA, ne2A), zA, TWO), zB = pk_mad(pk_ne1(hB, ne2B), zB, TWO);
				zA = pk_mad(pk_ne1(hA, ne1A), zA, ONE), zB = pk_mad(pk_ne1(hB, ne1B), zB, ONE);
				zA = pk_mad(pk_ne1(hA, mA), zA, 0), zB = pk_mad(pk_ne1(hB, mB), zB, 0);
				if (!FOLD) {
					zA = pk_mad(pk_ne1(ne1A, o1mA), EIGHT, zA), zB = pk_mad(pk_ne1(ne1B, O1.x), EIGHT, zB);
					zA = pk_mad(pk_ne1(pf1A, O1.y), C16, zA), zB = pk_mad(pk_ne1(pf1B, o1pB), C16, zB);
				}
				zA = pk_mad(pk_ne1(ne2A, o2mA), C32, zA), zB = pk_mad(pk_ne1(ne2B, O2.x), C32, zB);
				zA = pk_mad(pk_ne1(pf2A, O2.y), C64, zA), zB = pk_mad(pk_ne1(pf2B, o2pB), C64, zB);
				tbw = (uint32_t)zA | ((uint32_t)zB << 8); // bytes in column order: c0 = A.lo, c1 = B.lo, c2 = A.hi, c3 = B.hi
			}
			// ---- lane geometry of the chunk: j = k + 1 may reach rj = min(tl, ql - d); query index = j + d, d = c - 1 - tl
			const int32_t cbp = both16(cb);
			const int32_t xA = pk_sub(pk_sub(T0, cbp), RA);                  // ql - d of A's columns (garbage beyond cmax, where H is dead)
			const int32_t rjA = pk_minu(xA, TLp), rjB = pk_minu(pk_sub(xA, ONE), TLp);
			const int32_t dA = pk_sub(pk_add(RA, cbp), TL1), dB = pk_add(dA, ONE);
			// ---- the rare work in front of the extension — the chunk sticks out of the window, holds a window edge, a shrink is near — behind ONE
			// uniform test: an interior chunk (most chunks) pays one branch for the three (a uniform branch costs a wave 17-31 cycles,
			// profiles/r05/branch_rates_microbench.txt): 1024 x 10 kb -2 % score-only, -3 % with CIGAR.  (The rare work behind the extension — end
			// cell, good-bit words, flag word — under the same test as well: no further gain, seven more spilled SGPRs; left as it was.)
			// (not in the span geometry and the score-only five / six-slot copies on biased offsets: that much slot state leaves no scalar register for the
			// flag — 1250 x 50 kb +1.6 %, 1024 x 15 kb +1.7 %; with CIGAR the biased copies gain 2 % like the rest)
			const bool special = is_span(T, K) || (BI4 && !TB) || !inside || g == ga || g == gb || track_good;
			// ---- a chunk that sticks out of the window: the columns outside are not computed by the reference — dead
			int32_t outA = 0, outB = 0; // 0xffff in the halves of columns outside [lo, hi]
			uint32_t bits = 0, gbits = 0;
			if (special) { // uniform
			if (!inside) { // uniform
				const int32_t lo_r = both16(min(max(lo - cb, 0), 256)), hi_r1 = both16(min(max(hi - cb + 1, 0), 256));
				const int32_t RB = pk_add(RA, 0x00010001), RB1 = pk_add(RA, 0x00020002);
				outA = pk_nonzero_mask(pk_subsat(lo_r, RA) | pk_subsat(RB, hi_r1));
				outB = pk_nonzero_mask(pk_subsat(lo_r, RB) | pk_subsat(RB1, hi_r1));
				hA = bfi(outA, kDeadPair, hA), hB = bfi(outB, kDeadPair, hB);
				ne1A = bfi(outA, kDeadPair, ne1A), ne1B = bfi(outB, kDeadPair, ne1B);
				ne2A = bfi(outA, kDeadPair, ne2A), ne2B = bfi(outB, kDeadPair, ne2B);
				nf1A = bfi(outA, kDeadPair, nf1A), nf1B = bfi(outB, kDeadPair, nf1B);
				nf2A = bfi(outA, kDeadPair, nf2A), nf2B = bfi(outB, kDeadPair, nf2B);
			}
			// ---- edge rule (miniwfa.c:325-326): H is the max of the five, so "any live" == "H live"; one lane holds the edge column
			if (g == ga || g == gb) { // uniform
				if (g == ga) {
					const int32_t rel = lo - cb;
					const int32_t v = half_of(__builtin_amdgcn_readlane((rel & 1) ? hB : hA, rel >> 2), (rel >> 1) & 1);
					bits |= v >= -1 - B ? 1u : 0u;
				}
				if (g == gb) {
					const int32_t rel = hi - cb;
					const int32_t v = half_of(__builtin_amdgcn_readlane((rel & 1) ? hB : hA, rel >> 2), (rel >> 1) & 1);
					bits |= v >= -1 - B ? 2u : 0u;
				}
			}
			// good bits: some array holds an in-matrix offset (miniwfa.c:139-142) <=> j <= rj for a live value (dead: j is huge)
			if (track_good) { // uniform
				auto bad = [&](int32_t v, int32_t rj) { return pk_subsat(pk_add(v, ONEB), rj); }; // zero iff good
				const int32_t bA = pk_minu(pk_minu(bad(hA, rjA), pk_minu(bad(ne1A, rjA), bad(nf1A, rjA))), pk_minu(bad(ne2A, rjA), bad(nf2A, rjA))) | outA;
				const int32_t bB = pk_minu(pk_minu(bad(hB, rjB), pk_minu(bad(ne1B, rjB), bad(nf1B, rjB))), pk_minu(bad(ne2B, rjB), bad(nf2B, rjB))) | outB;
				gbits = (uint32_t)((bA & 0xffff) == 0) | (uint32_t)((bB & 0xffff) == 0) << 1 | (uint32_t)(((uint32_t)bA >> 16) == 0) << 2 | (uint32_t)(((uint32_t)bB >> 16) == 0) << 3;
				if (BI && (s_new & 255) == 0) { // uniform: the range checks of wide_bias
					const int32_t width = both16(kBiasMargin), from = both16(-1 - B - kBiasMargin), top = both16(32768 - kBiasMargin);
					auto stray = [&](int32_t v, int32_t lo_end) { return pk_subsat(width, pk_sub(v, lo_end)); }; // nonzero iff lo_end <= v < lo_end + kBiasMargin
					const int32_t any = stray(hA, from) | stray(hB, from) | stray(ne1A, from) | stray(ne1B, from) | stray(ne2A, from) | stray(ne2B, from) |
					                    stray(nf1A, from) | stray(nf1B, from) | stray(nf2A, from) | stray(nf2B, from) | stray(hA, top) | stray(hB, top);
					if (__ballot(any != 0)) bits |= 0x80u;
				}
			}
			} // special
			// ---- the new E/F are final: age the registers, publish this chunk's outer columns for the neighbouring slots
			if (FOLD) { // with the row read for the mismatch term: what the gap opens from e1 penalties from now (HX is not masked: it is dead outside ITS window)
				ne1A = pk_max(ne1A, HX.x), ne1B = pk_max(ne1B, HX.y), nf1A = pk_max(nf1A, HX.x), nf1B = pk_max(nf1B, HX.y);
				if (TB) { // the forward bits: this cell's E1 / F1 (dead outside the window) exceed the H a gap would be opened from
					const int32_t EIGHT = 0x00080008, C16 = 0x00100010;
					const int32_t fA = pk_mad(pk_ne1(ne1A, HX.x), EIGHT, pk_mad(pk_ne1(nf1A, HX.x), C16, 0));
					const int32_t fB = pk_mad(pk_ne1(ne1B, HX.y), EIGHT, pk_mad(pk_ne1(nf1B, HX.y), C16, 0));
					tbw |= (uint32_t)fA | ((uint32_t)fB << 8);
				}
			}
			e1h[P1][k][0] = ne1A, e1h[P1][k][1] = ne1B, f1h[P1][k][0] = nf1A, f1h[P1][k][1] = nf1B;
			e2h[P2][k][0] = ne2A, e2h[P2][k][1] = ne2B, f2h[P2][k][0] = nf2A, f2h[P2][k][1] = nf2B;
#pragma unroll
			for (int i = 0; i < 2; ++i) {
				if (E1 == 2) {
					asm volatile("v_swap_b32 %0, %1" : "+v"(e1h[0][k][i]), "+v"(e1h[1][k][i]));
					asm volatile("v_swap_b32 %0, %1" : "+v"(f1h[0][k][i]), "+v"(f1h[1][k][i]));
				}
				if (E2 == 2) {
					asm volatile("v_swap_b32 %0, %1" : "+v"(e2h[0][k][i]), "+v"(e2h[1][k][i]));
					asm volatile("v_swap_b32 %0, %1" : "+v"(f2h[0][k][i]), "+v"(f2h[1][k][i]));
				}
			}
			put_edge(k, ne1B, ne2B, nf1A, nf2A);

#if MWF_B2_TIMING == 2
			if (timed) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tc2) : "v"(hA), "v"(hB), "v"(rjA), "v"(dA) : "memory");
#endif
			// ---- match extension, first probe (FULL bases): j clamped to rj makes room = rj - j zero for dead and phantom offsets
			const int32_t jA = pk_minu(pk_add(hA, ONEB), rjA), jB = pk_minu(pk_add(hB, ONEB), rjB);
			const int32_t iqA = pk_add(jA, dA), iqB = pk_add(jB, dB);
			int32_t cnt[4]; // columns c0 (A.lo), c1 (B.lo), c2 (A.hi), c3 (B.hi)
			if (S2) {
				// Eight LDS reads (two dwords of each sequence for each of the four columns) go out back to back and are waited for ONCE:
				// left to itself the compiler recycles one register quad and pays four dependent LDS round trips.  Inline asm: the
				// reads keep their order (volatile), the single wait takes every result as an operand so that no use can move above it.
				// (Addresses are LDS byte addresses: the dynamic LDS of this kernel starts at 0 — it has no static LDS.)
				uint64_t tw[4], qw[4];
#pragma unroll
				for (int u = 0; u < 4; ++u) {
					const uint32_t J = (uint32_t)((u & 1) ? jB : jA), Q = (uint32_t)((u & 1) ? iqB : iqA);
					const uint32_t ta = (u & 2) ? (J >> 18) & 0x3ffcu : (J >> 2) & 0x3ffcu;
					const uint32_t qa = ((u & 2) ? (Q >> 20) : ((Q >> 4) & 0xfffu)) * 4u + (uint32_t)qoff;
					asm volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(tw[u]) : "v"(ta));
					asm volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(qw[u]) : "v"(qa));
				}
				asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(tw[0]), "+v"(tw[1]), "+v"(tw[2]), "+v"(tw[3]), "+v"(qw[0]), "+v"(qw[1]), "+v"(qw[2]), "+v"(qw[3]));
#pragma unroll
				for (int u = 0; u < 4; ++u) {
					const uint32_t J = (uint32_t)((u & 1) ? jB : jA), Q = (uint32_t)((u & 1) ? iqB : iqA);
					// v_alignbit uses bits 4:0 of the shift.  Bit 15 of the LOW half must not leak into a high half's shift: j never has it (j <= rj <=
					// tl < 32767), a query index can — column 0 (the pad column, lane 0 of chunk 0) clamps to index -1
					// (biased offsets: targets beyond 32 kb — j has a bit 15 too)
					const uint32_t tsh = (u & 2) ? (BI ? (J >> 15) & 30u : J >> 15) : J << 1, qsh = (u & 2) ? (Q >> 15) & 30u : Q << 1;
					cnt[u] = lead_eq2(__builtin_amdgcn_alignbit((uint32_t)(tw[u] >> 32), (uint32_t)tw[u], tsh) ^ __builtin_amdgcn_alignbit((uint32_t)(qw[u] >> 32), (uint32_t)qw[u], qsh));
				}
			} else {
				// six probe words per column: two columns in flight (one where three slots of state must fit 128 VGPRs)
				constexpr int PF = (K >= 3 && T >= 512) ? 1 : 2;
#pragma unroll
				for (int h2 = 0; h2 < 4 / PF; ++h2) {
					Probe8 pr[PF];
					int32_t jj[PF], aq[PF];
#pragma unroll
					for (int v = 0; v < PF; ++v) {
						const int u = PF * h2 + v;
						jj[v] = (int32_t)((uint32_t)((u & 1) ? jB : jA) >> ((u & 2) ? 16 : 0) & 0xffffu);
						aq[v] = (int32_t)((uint32_t)((u & 1) ? iqB : iqA) >> ((u & 2) ? 16 : 0) & 0xffffu) + qoff;
						probe8_issue(pr[v], jj[v], aq[v]);
					}
#pragma unroll
					for (int v = 0; v < PF; ++v) cnt[PF * h2 + v] = probe8_count(pr[v], jj[v], aq[v]);
				}
			}
			typedef unsigned short us2_t __attribute__((ext_vector_type(2)));
			const int32_t cA = MWF_BC(int32_t, (us2_t)__builtin_amdgcn_cvt_pk_u16((uint32_t)cnt[0], (uint32_t)cnt[2])); // saturating
			const int32_t cB = MWF_BC(int32_t, (us2_t)__builtin_amdgcn_cvt_pk_u16((uint32_t)cnt[1], (uint32_t)cnt[3]));
#if MWF_B2_TIMING == 2
			if (timed) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tc3) : "v"(cA), "v"(cB) : "memory");
#endif
			const int32_t FULLp = both16(FULL);
			const int32_t m9A = pk_minu(cA, pk_sub(rjA, jA)), m9B = pk_minu(cB, pk_sub(rjB, jB)); // > FULL: the whole probe matched, room left
			int32_t nmA = pk_minu(m9A, FULLp), nmB = pk_minu(m9B, FULLp);
			const int32_t pendp = pk_subsat(m9A, FULLp) | pk_subsat(m9B, FULLp);
			// ---- a run of >= FULL matches continues (the cells near the alignment path, and one first probe in 4^FULL by chance).  Each
			// lane first walks its own runs, four trips at most; what is still open then the whole wave walks.
			if (__ballot(pendp != 0)) {
				int32_t hv[4] = {half_of(hA, 0) + B, half_of(hB, 0) + B, half_of(hA, 1) + B, half_of(hB, 1) + B}; // true offsets
				int32_t nmat[4] = {(int32_t)((uint32_t)nmA & 0xffffu), (int32_t)((uint32_t)nmB & 0xffffu), (int32_t)((uint32_t)nmA >> 16), (int32_t)((uint32_t)nmB >> 16)};
				const uint32_t pend = (uint32_t)(((uint32_t)m9A & 0xffffu) > (uint32_t)FULL) | (uint32_t)(((uint32_t)m9B & 0xffffu) > (uint32_t)FULL) << 1 |
				                      (uint32_t)(((uint32_t)m9A >> 16) > (uint32_t)FULL) << 2 | (uint32_t)(((uint32_t)m9B >> 16) > (uint32_t)FULL) << 3;
				uint32_t open = 0;
#pragma unroll
				for (int i = 0; i < 4; ++i) {
					if (__ballot((pend >> i) & 1u) == 0) continue; // uniform
					if ((pend >> i) & 1u) {
						int32_t n = FULL; // pend is only set for a full first probe of an in-matrix cell with room left
						const int32_t j = hv[i] + 1, q = c0 + i - 1 - tl + j, rm = min(tl - j, ql - q), aqq = qoff + q;
						for (int trip = 0; n < rm; ++trip) {
							if (trip == 4) { open |= 1u << i; break; }
							if (S2) {
								const int32_t m = min(lead_eq2(seq16(0, j + n) ^ seq16(qoff, q + n)), 16);
								n += m;
								if (m < 16) break;
							} else {
								const uint32_t xa = seq4(j + n) ^ seq4(aqq + n), xb = seq4(j + n + 4) ^ seq4(aqq + n + 4);
								if (xa | xb) { n += xa ? min(lead_eq(xa), 4) : 4 + min(lead_eq(xb), 4); break; }
								n += 8;
							}
						}
						nmat[i] = min(n, rm);
					}
				}
				for (unsigned long long owners = __ballot(open != 0); owners; owners &= owners - 1) {
					const int32_t src = (int32_t)__builtin_ctzll(owners);
					const int32_t c0s = cb + 4 * src;
					const uint32_t ob = (uint32_t)__builtin_amdgcn_readlane((int32_t)open, src);
#pragma unroll
					for (int i = 0; i < 4; ++i) {
						if (!((ob >> i) & 1u)) continue; // uniform
						const int32_t hh = __builtin_amdgcn_readlane(hv[i], src);
						const int32_t j = hh + 1, q = c0s + i - 1 - tl + j, rm = min(tl - j, ql - q);
						const int32_t n = S2 ? run_wave16(qoff, j, q, rm, 80) : run_wave2(j, qoff + q, rm, 40);
						nmat[i] = lane == src ? n : nmat[i];
					}
				}
				nmA = pair_of(nmat[0], nmat[2]), nmB = pair_of(nmat[1], nmat[3]);
			}
			const int32_t hxA = pk_add(hA, nmA), hxB = pk_add(hB, nmB); // extended
			if (forecast) far = pk_max(far, pk_max(hxA, hxB));
			// ---- termination test of the extension sweep (miniwfa.c:405-409): only column ql+1 can hold the end cell
			unsigned long long fm = 0;
			int32_t done_info = 0;
			if ((uint32_t)(cfin - cb) < (uint32_t)kChunk && cfin >= lo && cfin <= hi) { // uniform
				const int32_t rel = cfin - cb, hi_half = (rel >> 1) & 1;
				const int32_t hv = half_of((rel & 1) ? hxB : hxA, hi_half) + B, nm = (int32_t)((uint32_t)((rel & 1) ? nmB : nmA) >> (hi_half ? 16 : 0) & 0xffffu);
				const uint32_t f = (uint32_t)(lane == (rel >> 2)) & (uint32_t)(hv == tl - 1) & inm_bit(ql - tl, hv - nm, tl, ql);
				done_info = (f && nm == 0) ? (int32_t)((tbw >> (8 * (rel & 3))) & 7u) : 0;
				fm = __ballot(f != 0);
			}
			*(int2*)(rown + off) = make_int2(hxA, hxB);
			++n_stores;
#if MWF_B2_TIMING == 2
			if (timed) {
				asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tc4) : "v"(hxA), "v"(hxB) : "memory");
				chunk_timed = true;
				ct[0] = (uint32_t)min(tc1 - tc0, (uint64_t)4095), ct[1] = (uint32_t)min(tc2 - tc1, (uint64_t)4095);
				ct[2] = (uint32_t)min(tc3 - tc2, (uint64_t)4095), ct[3] = (uint32_t)min(tc4 - tc3, (uint64_t)4095);
			}
#endif
			if (TB && c0 >= origin && c0 <= hi) *(uint32_t*)(M.tb + tb_used - origin + c0) = tbw;
			if (track_good) {
				unsigned long long *gword = M.good + (int64_t)newH * fresh(A).GW + g * 4;
#pragma unroll
				for (int i = 0; i < 4; ++i) {
					const unsigned long long m = __ballot((gbits >> i) & 1u);
					if (lane == 0) gword[i] = m;
				}
			}
			if (fm) bits |= 4u | (uint32_t)__builtin_amdgcn_readlane(done_info, (int32_t)__builtin_ctzll(fm)) << 4;
			if (bits && lane == 0) atomicOr((unsigned int*)&sh.flags[npar][0], bits);
		}

		if (forecast) {
			const int32_t m = wave_max(max(lo16(far), hi16(far))) + B;
			if (lane == 0 && m >= 0) atomicMax(&sh.word[3], m);
		}
		// Everything older than this penalty's last operations must be complete before another wave may load it (vmcnt retires
		// in issue order).  With every lag >= 3 the rows written now are first loaded two penalties from now: the youngest store
		// may stay in flight across the barrier.
#ifdef MWF_B2_TIMING
		const uint64_t tm2 = __builtin_readcyclecounter();
#endif
		// the coming penalty: its rows, and the request for the first active chunk's (five loads, younger than every store)
		bn = bn + RS == ring_bytes ? 0u : bn + RS, bx = bx + RS == ring_bytes ? 0u : bx + RS;
		b1 = b1 + RS == ring_bytes ? 0u : b1 + RS, b2 = b2 + RS == ring_bytes ? 0u : b2 + RS;
		const bool young_store = relaxed_stores && n_stores > 0 && !TB && !track_good;
		if (young_store) asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
		else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#ifdef MWF_B2_TIMING
		const uint64_t tm3 = __builtin_readcyclecounter();
#endif
		__builtin_amdgcn_s_barrier();
		asm volatile("" ::: "memory");

		// ---- bookkeeping, identical on every thread
		const uint32_t fl = (uint32_t)uni(sh.flags[npar][0]);
#ifdef MWF_B2_TIMING
		if (trace_band && tid == (fresh(A).max_iter < 0 ? (int32_t)-fresh(A).max_iter : 0) && s_new - 1 < fresh(A).dbg_cap) { // cycles: header | chunks, drain | barrier+flags; chunks this wave ran in bits 28..
			const uint64_t tm4 = __builtin_readcyclecounter();
#if MWF_B2_TIMING == 2 // the first active chunk of the wave: rows + recurrence | masks, liveness, geometry, edge stores | first probe | walks, store
			M.dbg[2 * (s_new - 1)] = (int32_t)(ct[0] | ct[1] << 12 | (uint32_t)n_act << 28);
			M.dbg[2 * (s_new - 1) + 1] = (int32_t)(ct[2] | ct[3] << 12);
			(void)tm4;
#else
			M.dbg[2 * (s_new - 1)] = (int32_t)(min((uint32_t)(tm1 - tm0), 65535u) | min((uint32_t)(tm2 - tm1), 65535u) << 16);
			M.dbg[2 * (s_new - 1) + 1] = (int32_t)(min((uint32_t)(tm3 - tm2), 4095u) | min((uint32_t)(tm4 - tm3), 65535u) << 12 | (uint32_t)n_act << 28);
#endif
		}
#endif
		if (fl & 1u) wf_lo = lo;
		if (fl & 2u) wf_hi = hi;
		const int32_t done = (int32_t)((fl >> 2) & 1u), payload = (int32_t)((fl >> 4) & 7u);
		if (BI && (fl & 0x80u)) { R.status = ST_BAND_OVERFLOW; return true; } // a value came near the end of its 16-bit range (wide_bias)
		s = s_new, curH = newH, par = npar;
		{
			const int32_t oldest = epos[D - 1];
#pragma unroll
			for (int a = D - 1; a > 0; --a) epos[a] = epos[a - 1];
			epos[0] = oldest;
		}
		// The slot mapping follows the window's start: down at once (the slot that wraps held a chunk beyond the window's reach under the old
		// mapping: long idle), up only when the chunks below the start have been out of the window for kAgeOut penalties — until then
		// they age (and, FOLD, fold the rows they computed) in place.  A slot that wrapped while it aged would run the chunk NWK further up
		// through the ordinary code, and that chunk's dead stores may lie beyond the end of the row (found by profiles/fuzz_fold.py: a
		// 2.3 kb unrelated pair on four slots lost cells of the NEXT row that way — n_iter off by 82).
		if (gl_next < gl) gl = gl_next, remap(gl), up_wait = 0;
		else if (gl_next > gl) {
			up_min = up_wait == 0 ? gl_next : min(up_min, gl_next);
			if (++up_wait > kAgeOut) gl = up_min, remap(gl), up_wait = 0;
		} else up_wait = 0;
		if (TB) tb_used += row_bytes;
		if ((s & 0xff) == 0) { // shrink (reference wf_stripe_shrink, miniwfa.c:144-171) on the interleaved good bits
			if (tid == 0) sh.red[0] = 0x7fffffff, sh.red[1] = -1;
			__syncthreads();
			const int32_t gfirst = wf_lo >> 8, n_words = ((wf_hi >> 8) - gfirst + 1) * 4, GWc = fresh(A).GW;
			for (int32_t q = tid; q < n_words; q += T) {
				const int32_t gg = gfirst + (q >> 2), kq = q & 3, base = gg * kChunk;
				unsigned long long m = 0;
				for (int32_t j = 0; j < nH; ++j)
					if (sh.rng_lo[j] <= sh.rng_hi[j] && sh.rng_lo[j] <= base + kChunk - 1 && sh.rng_hi[j] >= base) m |= M.good[(int64_t)j * GWc + gg * 4 + kq];
				m &= lane_mask(base, kq, wf_lo, wf_hi);
				if (m) {
					atomicMin(&sh.red[0], base + 4 * (int32_t)__builtin_ctzll(m) + kq);
					atomicMax(&sh.red[1], base + 4 * (63 - (int32_t)__builtin_clzll(m)) + kq);
				}
			}
			__syncthreads();
			const int32_t glo = uni(sh.red[0]), ghi = uni(sh.red[1]);
			if (ghi < 0) { R.status = ST_INTERNAL; return true; }
			wf_lo = glo, wf_hi = ghi;
		}
		cells += hi - lo + 1;
		if (cells > iter_limit || s > s_limit) { // miniwfa.c:422-425
			R.status = ST_STOPPED;
			return true;
		}
		if (done) {
			R.info = payload;
			return true;
		}
		if (forecast) { // will the window outgrow the chunks this workgroup holds? then hand the pair back now, with the estimate
			est_window = window_forecast(s, uni(sh.word[3]), tl, ql, (NWK - 1) * kChunk - 64, NWK >= 24 && NWK < 64); // (the window this geometry is chosen for: kBand*Window in mwf_plan.cpp)
			if (est_window) { R.status = ST_BAND_OVERFLOW; return true; }
		}
		return false;
	};
	for (;;)
		if (step()) break;
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	R.s = s, R.cells = est_window ? -(int64_t)est_window : cells; // (a pair handed back early: the window it is expected to need, negated, for the host's choice of the next kernel)
	return R;
}

// Workgroups share a CU: 2 x 512, 4 x 256, 8 x 128 or 16 x 64 threads = 4 waves per SIMD, i.e. at most 128 VGPRs; with traceback
// the smaller ones get 168 (3 per SIMD).  768 threads: one workgroup per CU.
template <int T, int K, int E1, int E2, bool TB, bool S2, bool BI4 = false, bool FOLD = false>
__global__ __launch_bounds__(T, is_span(T, K) ? (T == 1024 ? 4 : 3) : T == 1024 ? 4 : T == kWideT ? kWideWaves : T <= 512 ? ((TB && T < 512) ? 3 : 4) : kWaves768) void wfa_band2_kernel(const BatchArgs)
{
	constexpr int NWK = (T / 64) * K, D = (E1 > E2 ? E1 : E2) + 1;
	// the arguments are read from the kernarg segment where they are used (dev::kernel_args / dev::fresh), never held for the kernel's lifetime
	KArgs &A0 = kernel_args();
	// the sequence copy starts at LDS offset 0 (the probes' inline-asm reads take LDS byte addresses): true while this kernel has no
	// static LDS — trap rather than compute on the wrong bytes should that ever change
	if ((uint32_t)(uintptr_t)lds2 != 0u) __builtin_trap();
	// the bookkeeping words and the edge table sit behind the sequence copy
	typedef Band2Lds<D, NWK> LdsT;
	const int32_t lds_seq = A0.band_lds_seq;
	LdsT *const L = (LdsT*)(lds2 + lds_seq);
	Shared &sh = L->sh;
	const int32_t edge_base = lds_seq + (int32_t)offsetof(LdsT, edge);
	CigLocal cig_loc;
	cig_loc.base = 0, cig_loc.left = 0;
	for (int32_t round = 0;; ++round) {
		KArgs &A = fresh(A0);
		// a work counter, or — queue == null: a launch of one workgroup per pair — pair blockIdx.x and nothing else (no counter to zero first)
		// (one counter: at the ~3-6 million pairs per second of these geometries its ~12.7 ns per pair do not show — 20 000 x 1 kb 6.2 Gbp/s with the lane
		// kernel's partitioned counters, 6.3 without)
		if (threadIdx.x == 0) sh.item = A.queue ? (int32_t)atomicAdd(A.queue, 1) : (round == 0 ? (int32_t)blockIdx.x : A.n_pairs), sh.word[2] = 0;
		__syncthreads();
		const int32_t item = uni(sh.item);
		__syncthreads();
		if (item >= A.n_pairs) break;
		const int32_t pair = A.order ? A.order[item] : item;
		PairMem M;
		pair_mem(A, (int32_t)blockIdx.x, pair, M);
		const int32_t qoff = S2 ? ((M.tl >> 4) + 2) * 4 : ((M.tl + 3) & ~3) + 8; // both sequences start on a dword
		PassResult R;
		R.status = ST_OK, R.s = 0, R.info = 0, R.n_snap = 0, R.cells = 0;
		if (S2) {
			uint32_t bad = pack2bit<T>(M.ts, M.tl, 0);
			bad |= pack2bit<T>(M.qs, M.ql, qoff);
			// a base other than A/C/G/T: the host re-runs the pair on the byte-wise copy of this kernel
			if (bad) sh.word[2] = 1;
			__syncthreads();
			if (uni(sh.word[2])) R.status = ST_ALPHABET;
		} else {
			for (int32_t j = threadIdx.x; j < M.tl; j += T) lds2[j] = M.ts[j];
			for (int32_t j = threadIdx.x; j < M.ql; j += T) lds2[qoff + j] = M.qs[j];
			__syncthreads();
		}
		const bool trace = A.dbg && pair == A.debug_pair;
		if (R.status == ST_OK) R = band2_pass<T, K, E1, E2, TB, S2, BI4, FOLD>(A, M, sh, edge_base, qoff, trace);
		if (T == 512 && K == 4 && R.n_snap && threadIdx.x == 0 && fresh(A0).report_wide) atomicOr((unsigned int*)(fresh(A0).cig_head + 1), 1u); // (mwf_plan.cpp: PlanCache::wide_state)
		R.n_snap = 0;
		if (S2) M.t2 = lds2, M.q2 = lds2 + qoff; // the traceback's back-match reads the 2-bit copies in LDS
		if (TB && FOLD) M.tb_fwd = 1;
		finish_pair(fresh(A0), M, (int32_t)blockIdx.x, pair, R, R.status, 0, T <= 256 ? &cig_loc : nullptr); // (block mode: the geometries of the short pairs — thousands per launch)
	}
}

template <int T, int K, int E1, int E2>
constexpr int lds_tail() { return (int)sizeof(Band2Lds<(E1 > E2 ? E1 : E2) + 1, (T / 64) * K>); }

template <int T, int K, int E1, int E2, bool TB, bool S2, bool BI4 = false, bool FOLD = false>
void launch_variant(const BatchArgs &a, int grid, int lds, hipStream_t st)
{
	// the 512-thread (and wider) geometries with o1 == x: the folded form (band2_pass), two row loads less per chunk
	// (only on 2-bit sequence copies and with e1 == 2 — the default penalties and main.c's -a preset: the byte-wise geometry serves the rare pairs outside
	// plain A/C/G/T, and a folded copy of every kernel for e1 == 1 penalties bought nothing a benchmark showed; round 6 pruning: 132 -> 62 kernels in this object)
	if constexpr (!FOLD && T >= 512 && S2 && E1 == 2) {
		if (a.band_fold && a.pen.oe1 - a.pen.x == E1 && a.pen.oe1 < kFoldMaxLag) return launch_variant<T, K, E1, E2, TB, S2, BI4, true>(a, grid, lds, st);
	}
	// the attribute is per device and this may run on several host threads (mwf_wfa_batch_multi): set it on every launch that needs it
	if (lds > 48 * 1024) {
		(void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wfa_band2_kernel<T, K, E1, E2, TB, S2, BI4, FOLD>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
		(void)hipGetLastError();
	}
	hipLaunchKernelGGL((wfa_band2_kernel<T, K, E1, E2, TB, S2, BI4, FOLD>), dim3(grid), dim3(T), lds, st, a);
}

template <int T, int K, int E1, int E2, bool BI4 = false>
int launch_one(const BatchArgs &a0, int grid, int lds_seq, bool seq2, hipStream_t st)
{
	BatchArgs a = a0;
	a.band_lds_seq = lds_seq;
	const int lds = lds_seq + lds_tail<T, K, E1, E2>();
	if constexpr (is_span(T, K) || (T == 512 && K >= 4)) { // the 1024-thread and the 512 x 4 geometries exist on 2-bit sequence copies only (the host knows)
		if (!seq2) return -1;
		if (a.want_cigar) launch_variant<T, K, E1, E2, true, true, BI4>(a, grid, lds, st);
		else launch_variant<T, K, E1, E2, false, true, BI4>(a, grid, lds, st);
	} else if constexpr (T == 768) { // the byte-wise geometry: every pair outside plain A/C/G/T that the packed kernel takes runs here (the host knows: choose_kernel)
		if (seq2) return -1;
		if (a.want_cigar) launch_variant<T, K, E1, E2, true, false>(a, grid, lds, st);
		else launch_variant<T, K, E1, E2, false, false>(a, grid, lds, st);
	} else { // 64 ... 512 threads x 3 slots: 2-bit copies only
		if (!seq2) return -1;
		if (a.want_cigar) launch_variant<T, K, E1, E2, true, true>(a, grid, lds, st);
		else launch_variant<T, K, E1, E2, false, true>(a, grid, lds, st);
	}
	return hipGetLastError() == hipSuccess ? 0 : -2;
}

template <int T, int K, int E1, int E2, bool BI4 = false>
int occ_one(int lds_seq, bool seq2, bool tb)
{
	const int lds = lds_seq + lds_tail<T, K, E1, E2>();
	int n = 0;
	hipError_t e;
	if constexpr (is_span(T, K) || (T == 512 && K >= 4)) {
		if (!seq2) return 0;
		e = tb ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, wfa_band2_kernel<T, K, E1, E2, true, true, BI4>, T, lds)
		       : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, wfa_band2_kernel<T, K, E1, E2, false, true, BI4>, T, lds);
	} else if constexpr (T == 768) {
		if (seq2) return 0;
		e = tb ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, wfa_band2_kernel<T, K, E1, E2, true, false>, T, lds)
		       : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, wfa_band2_kernel<T, K, E1, E2, false, false>, T, lds);
	} else {
		if (!seq2) return 0;
		e = tb ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, wfa_band2_kernel<T, K, E1, E2, true, true>, T, lds)
		       : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, wfa_band2_kernel<T, K, E1, E2, false, true>, T, lds);
	}
	return e == hipSuccess ? n : 0;
}

} // namespace

// the packed kernel: (e1,e2) instantiated, sequences fit LDS (the host checks), every H lag >= 1
bool band2_supported(const Penalty &p)
{
	return ((p.e1 == 2 && p.e2 == 1) || (p.e1 == 2 && p.e2 == 2) || (p.e1 == 1 && p.e2 == 1)) && p.nH <= kMaxRing; // (window table in LDS; rows read as dead up to 256 columns beyond their window)
}

#ifdef MWF_BAND_DEV
#define MWF_BAND2_REST(FN, ...)
#else
#define MWF_BAND2_REST(FN, ...)                                                     \
	if (g.block == 256) MWF_BAND2_PEN(FN, 256, 3, __VA_ARGS__)                      \
	if (g.block == 128) MWF_BAND2_PEN(FN, 128, 3, __VA_ARGS__)                      \
	if (g.block == 64) MWF_BAND2_PEN(FN, 64, 3, __VA_ARGS__)
#endif
#ifdef MWF_BAND_DEV
#define MWF_BAND2_PEN(FN, T, K, ...) { if (a_e1 == 2 && a_e2 == 1) return FN<T, K, 2, 1>(__VA_ARGS__); }
#else
#define MWF_BAND2_PEN(FN, T, K, ...)                                                \
	{                                                                               \
		if (a_e1 == 2 && a_e2 == 1) return FN<T, K, 2, 1>(__VA_ARGS__);             \
		if (a_e1 == 2 && a_e2 == 2) return FN<T, K, 2, 2>(__VA_ARGS__);             \
		if (a_e1 == 1 && a_e2 == 1) return FN<T, K, 1, 1>(__VA_ARGS__);             \
	}
#endif
/* (the five / six-slot copies on biased offsets: gap extensions (2, 1) only — band2_biased512_supported; other penalty sets take the span geometry) */
#define MWF_BAND2_PEN4B(FN, ...)                                                    \
	{                                                                               \
		if (a_e1 == 2 && a_e2 == 1) return FN<512, kW4K, 2, 1, true>(__VA_ARGS__);     \
	}
#define MWF_BAND2_PEN4C(FN, ...)                                                    \
	{                                                                               \
		if (a_e1 == 2 && a_e2 == 1) return FN<512, kW4K + 1, 2, 1, true>(__VA_ARGS__); \
	}
#define MWF_BAND2_DISPATCH(FN, ...)                                                 \
	do {                                                                            \
		if (g.block == 512 && g.span > 8 * kW4K * 256 && g.packed == 2) MWF_BAND2_PEN4C(FN, __VA_ARGS__) /* ... six slots on biased offsets (pairs of up to ~21 kb) */ \
		if (g.block == 512 && g.span > 512 / 64 * 3 * 256 && g.packed == 2) MWF_BAND2_PEN4B(FN, __VA_ARGS__) /* ... five slots on biased offsets (pairs of up to ~18 kb) */ \
		if (g.block == 512 && g.span > 512 / 64 * 3 * 256) MWF_BAND2_PEN(FN, 512, 4, __VA_ARGS__) /* 32 chunks: windows of up to 7872 columns */ \
		if (g.block == 512) MWF_BAND2_PEN(FN, kWideT, kWideK, __VA_ARGS__) \
		if (g.block == 768) MWF_BAND2_PEN(FN, 768, 2, __VA_ARGS__)                  \
		if (g.block == 1024) MWF_BAND2_PEN(FN, kSpanT, kSpanK, __VA_ARGS__) \
		MWF_BAND2_REST(FN, __VA_ARGS__)                                             \
	} while (0)

int launch_band2(const BatchArgs &a, int grid, const BandGeom &g, void *stream)
{
	const int a_e1 = a.pen.e1, a_e2 = a.pen.e2;
	MWF_BAND2_DISPATCH(launch_one, a, grid, g.lds_bytes, g.seq2 != 0, (hipStream_t)stream);
	return -1;
}

int band2_span_chunks() { return kSpanT / 64 * kSpanK; }
int band2_biased512_chunks() { return 8 * kW4K; }
bool band2_biased512_supported(const Penalty &p) { return p.e1 == 2 && p.e2 == 1; }

int band2_kernel_occupancy(const Penalty &p, const BandGeom &g, bool cigar)
{
	const int a_e1 = p.e1, a_e2 = p.e2;
	MWF_BAND2_DISPATCH(occ_one, g.lds_bytes, g.seq2 != 0, cigar);
	return 0;
}

} // namespace mwf
