// mwf_engine.cpp — host side of libmwf_hip.so: the device engine (stream, memory pools, launch
// geometry, retry policy) and every entry point of include/miniwfa.h.
//
// Boundary: the reference's device-free API mwf_wfa_exact/auto/chain
// (miniwfa.c:603-615, :850-908) is kept; a call ships its pair(s) to HBM, runs the kernels of
// mwf_kernels.hip / mwf_band2.hip / mwf_lane.hip / mwf_mid.hip / mwf_sys.hip and copies back (s, n_iter, n_cigar, CIGAR).  r->cigar is
// allocated from the caller's kalloc arena exactly as the reference does (miniwfa.c:434).  kalloc arenas for
// scratch are replaced by device pools that only ever grow:
//   * one workspace per engine (ring, traceback arena, row table, CIGAR scratch, snapshots);
//   * ONE device allocation per batch (inputs, processing order, every result array), recycled through the
//     engine when the batch is freed, so a program that calls mwf_wfa_exact in a loop never reaches hipMalloc;
//   * one pinned staging buffer per engine: a call's inputs go up in one host-to-device copy, its fixed-size
//     results (and, when asked for, all its CIGARs) come back in one device-to-host copy each.
// Engines themselves are pooled per device and handed to whichever host thread calls next; mwf_wfa_batch_multi
// deals the pairs of one call over several devices (reference main.c:67-72 is the serial loop it replaces).
//
// There is no CPU alignment path in this file or anywhere in the library.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <shared_mutex>
#include <numeric>
#include <string>
#include <chrono>
#include <cmath>
#include <functional>
#include <thread>
#include <vector>
#include "miniwfa.h"
#include "kalloc.h"
#include "mwf_internal.h"

using namespace mwf;

namespace {

[[noreturn]] void fatal(const char *what, const char *detail)
{
	fprintf(stderr, "[libmwf_hip] fatal: %s%s%s\n", what, detail ? ": " : "", detail ? detail : "");
	abort();
}

// One device buffer that only grows.
struct DevBuf {
	void *p = nullptr;
	size_t bytes = 0;
};

constexpr size_t kPinHalfMax = (size_t)16 << 20; // pinned staging: two halves of at most this many bytes
constexpr int kCigBlock = 256, kCigBlockGrid = 8192, kCigBlockPairs = 4096; // CIGAR pool in block mode (batch_common)
constexpr int kQueueSlots = 64;                  // work counters zeroed at the start of an align call, one per launch
// The lane kernel's launches (tens of thousands of read pairs) take a SET of kLaneCounters work counters, each on a cache line of its own:
// counter c deals the pairs c, c + 64, c + 128 ... of the order to the waves with blockIdx % 64 == c.  (One counter for all: 40 000 atomics on
// one address, ~12.7 ns each — 0.51 of the kernel's 0.61 ms.)  kLaneSets sets per align call, behind the plain counters in the same buffer.
constexpr int kLaneCounters = 64, kLaneStride = 32, kLaneSets = 4, kQueueInts = kQueueSlots + kLaneSets * kLaneCounters * kLaneStride;
constexpr int kMaxDevices = 64;

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

} // namespace

struct mwf_gpu_s {
	int device = 0;
	hipStream_t stream = nullptr;
	bool own_stream = false;
	int n_cu = 0;
	size_t total_mem = 0;
	std::string err;
	// tunables
	int block = 0;              // 0: choose from the batch
	bool div_aware = true;      // weigh the size classes' length limits by the batch's estimated divergence (batches built from host memory)
	int64_t tun_gen = 0;        // bumped by every successful mwf_gpu_set(): a cached plan of an align (PlanCache) is only replayed under the tunables it was made under
	int slots_per_cu = 0;       // 0: occupancy of the kernel
	int64_t coop_min_len = 0;
	int64_t tb_budget_mb = 0;   // 0: automatic
	int force_kind = -1;
	int res_pin_on = 1;        // small score-only batches: results written straight into pinned host memory (0: always copied back)
	int lane_chunks = 0;       // its window: 64-column chunks of LDS rows (1-4; 0: three for pairs of up to 400 bases of target + query, else four); a penalty only passes over the chunks the window has reached
	int lane_max_len = 400;    // (measured: 20 000 x 400 bp @ 5 % 1.37 against 1.58 ms with 767 pairs re-run, x 500 bp @ 2 % 0.69 / 1.17, profiles/r03/lane_longer_pairs.txt)
	                           // pairs whose longer sequence has at most this many bases try the one-diagonal-per-lane kernel first (0: never)
	int mid_max_pairs = -1;    // a batch of at most this many pairs may use the one-workgroup-per-pair, rings-in-LDS kernel (mwf_mid.hip) for its mid-size pairs (-1: one per CU; 0: never)
	int mid_block = 0;         // its threads per workgroup: 0 by span (256 up to 512 columns, else 1024), 256, 1024
	int band_pack = -1;        // int16-packed E/F registers in the band kernel: 0 never, otherwise whenever the value ranges allow
	int wide_slots = 0;        // chunk slots per wave of the 512-thread packed geometry: 0 by batch (four until an align has shown that three hold every pair), 3, 4
	int band_span = 1;         // the 1024-thread geometry of the packed band kernel (80 chunks, biased offsets: pairs of up to ~60 kb whose windows stay below ~20 000 columns): 0 never, 2: every pair it can take (tests)
	int ring16 = 1;            // generic kernel with E2/F2 in LDS: 16-bit ring rows in HBM while target length + penalty fits 16 bits (0: never)
	int ring16_block = 0;      // its threads per workgroup (0: 512 score-only — two workgroups per CU with the 64 KB LDS copy —, 768 with traceback)
	bool ring16_off_once = false; // set around the re-run of pairs whose offsets outgrew 16 bits
	int seq2bit = 1;           // packed band kernel: 2-bit sequence copy in LDS for pairs of plain A/C/G/T (0: always bytes)
	bool acgt_off_once = false; // set around the re-run of pairs that are not plain ACGT
	int lds_e2 = 1;            // generic kernel: keep E2/F2 in LDS where that applies (0: never)
	int scalar_generic = 0;    // 1: the generic kernel's original one-column-per-lane pass everywhere (comparison / fallback)
	int64_t coop_spin_limit = 1 << 23; // polls (about a microsecond each) before the whole-device kernel gives up on a workgroup
	int64_t coop_tb_cap = (int64_t)96 << 30; // whole-device traceback arena: first allocation never above this ...
	int64_t coop_tb_mult = 1;                // ... times this; doubled after an overflow while memory lasts
	int64_t lowmem_budget_mb = 0; // whole-device low-memory mode: first-pass traceback above this many MB -> true two-pass (0: automatic)
	int sys_p = 8;             // whole-device (systolic) kernel: penalties per hand-off block (4, 8 or 16)
	int sys_p2 = 0;            // ... of the SECOND pass of its low-memory mode (0: the same)
	int sys_c = 0;             // its columns per lane: 0 automatic (1 while the window is expected to fit the slots that way, else 4), 1, 4
	// workspace (per-stream pool)
	DevBuf ring, sring, good, tb, row_off, row_lo, cig_scratch, snap, snap_meta, seg, queue, dbg, coop_edge, coop_misc;
	DevBuf retry_ids;          // pair ids of a re-run (finalize): kept by the engine — round 4 allocated and freed one per re-run, a hipMalloc + hipFree of ~0.15 ms behind a 0.5 ms launch
	DevBuf sys_box, sys_prog, sys_log, sys_ep, sys_park, sys_ring, sys_sring, sys_good;
	DevBuf spare_block, spare_cig; // allocations of freed batches, waiting for the next batch
	int queue_next = 0;            // next unused work counter of the current align call
	int lane_set_next = 0;         // ... and next unused set of lane-kernel counters
	bool queue_clean = false;      // the work counters were zeroed by this align call's reset kernel (else a launch that needs one zeroes it itself)
	// pinned staging
	void *pin = nullptr;
	size_t pin_half = 0;
	hipEvent_t pin_ev[2] = {nullptr, nullptr};
	void *res_pin = nullptr;            // 4 KB of pinned host memory the kernels write a small score-only batch's results into (no copy back)
	const void *res_pin_owner = nullptr; // the batch whose result pointers currently lie in it
	bool pin_busy[2] = {false, false}; // a copy out of that half may still be in flight (pin_ev tells)
	hipEvent_t ev0 = nullptr, ev1 = nullptr;
	bool ev_pending = false;
	mwf_gpu_stats_t stats{};
	int64_t dev_bytes = 0, dev_bytes_peak = 0; // device memory this engine holds right now / held at most since the last "trim"
	std::map<uint64_t, int> occ_cache;         // kernel variant -> resident workgroups per CU
	int coop_grid = -1;
	int coop_grid_cap = 0;      // "coop_grid": at most this many workgroups for the whole-device kernel (0: one per CU)
	int coop_launch = 1;        // "coop_launch": whole-device kernel through hipLaunchCooperativeKernel (0: plain launch)
};

struct mwf_gpu_batch_s {
	mwf_gpu_t *g = nullptr;
	int32_t n = 0;
	bool owns_inputs = false;
	DevBuf block;              // the batch's one device allocation: [order | inputs (when owned) | results]
	const uint8_t *d_seqs = nullptr;
	int64_t seq_bytes = 0;
	const int64_t *d_t_off = nullptr, *d_q_off = nullptr;
	const int32_t *d_tl = nullptr, *d_ql = nullptr;
	std::vector<int32_t> h_tl, h_ql;
	int64_t max_seq_lds = 0;   // LDS bytes the band kernel needs to hold the longest pair's sequences
	int64_t max_tl = 0;        // longest target (offsets are target indices: bounds what a 16-bit offset must hold)
	int32_t *d_order = nullptr;
	std::vector<int32_t> h_order;   // what d_order holds: pair ids, grouped by size class, longest first inside a class
	std::vector<int32_t> h_len_order; // pair ids, longest pair first (stable): what every grouping is dealt from
	std::vector<int8_t> h_class;    // size class of every pair (0 generic, then band kernels: 1 wide, 2 small, 3 tiny, 4 micro; 5: the 1024-thread span geometry)
	std::vector<int8_t> h_kind;     // kernel that ran the pair last (0 generic, 1 whole-device, 2 band)
	float div_est = 0;              // divergence of the batch as a k-mer sketch of a few of its pairs saw it while the batch was built from host memory (0: unknown)
	std::vector<int8_t> h_acgt;     // from the host's look at the bytes while a batch is built from host memory: 1 both sequences are plain
	                                // A/C/G/T, 0 not (such a pair goes to the byte-wise sequence copy at once); empty: unknown (wrapped device
	                                // buffers — the 2-bit copy finds out on the device and the pair comes back as ST_ALPHABET)
	std::vector<int8_t> h_flags;    // bit 0: runs as high-memory although opt.step > 0 (its penalty bound is below step);
	                                // bit 1: shared the whole-device kernel with other pairs; bit 2: walk variant of the low-memory mode
	// results: one region of the block, fetched by one copy
	unsigned long long *d_cig_head = nullptr;
	int32_t *d_status = nullptr, *d_s = nullptr, *d_ncig = nullptr, *d_dbg4 = nullptr;
	int64_t *d_iter = nullptr, *d_cigoff = nullptr, *d_cells1 = nullptr;
	size_t out_off = 0, out_bytes = 0;
	bool out_in_pin = false;        // the result arrays lie in the engine's pinned result page (small score-only batches)
	bool results_preinit = false;   // the device result arrays came up initialised with the batch's upload (status -1, s -2, CIGAR counter 0): its first align needs no reset kernel
	DevBuf cig;                     // CIGAR pool (allocated by the first CIGAR-mode align)
	uint32_t *d_cig_pool = nullptr;
	int64_t cig_pool_words = 0;
	int32_t cig_block = 0;          // > 0: the pool is sized for workgroups that take it in blocks of this many words (batches of thousands of pairs)
	// state of the last align
	bool aligned = false, finalized = false, busy = false;
	mwf_opt_t opt{};
	std::vector<int32_t> h_s, h_ncig, h_status;
	std::vector<int64_t> h_iter, h_cigoff, h_cells1;
	int64_t cig_used = 0;           // words of the pool in use (known after finalize)
	int32_t cig_block_left = 0;     // block mode: workgroups (= partly used blocks the pool has slack for) the launches of this align may still spend; reset by every align
	std::vector<uint32_t> h_cig;    // host copy of the used part of the pool (fetch_cigars)
	bool h_cig_valid = false;
	// What the last align worked out from the pair lengths alone — size classes, processing order, per-class maxima — keyed by the
	// options and tunables it depends on: lengths do not change between aligns of a batch, so the next align with the same key skips
	// the per-pair pass (40 000 read pairs: ~2.5 ms of host work per align before, profiles/r04/short_reads.txt)
	struct PlanCache {
		bool valid = false;
		int32_t opt_key[8] = {0, 0, 0, 0, 0, 0, 0, 0};
		int64_t tun_key[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
		int64_t max_len = 0, max_bound = 0;
		bool has_groups = false, mid_bytes = false;
		// The wide class (512-thread geometry) holds 24 chunks with three slots per wave and 32 with four (2 % slower where three suffice).  0: not known yet —
		// four slots, and the kernel reports whether three would have held every pair; 1: three hold this batch under these options; 2: four are needed.
		int8_t wide_state = 0;
		bool wide_measured = false; // this align's first launch of the class ran on four slots with the report word zeroed
		struct GI { int32_t n = 0; int64_t max_len = 0, max_bound = 0, max_bound1 = 0, max_tl = 0, max_seq_lds = 0; } gi[15];
		std::vector<int8_t> cls0, flags0;
	} plan;
	std::vector<char> host_out;     // the fixed-size results as they came back (finalize)
	size_t out_bytes_score = 0;     // leading part of the result region a score-only, high-memory align needs back: head, status, s, n_iter
	// geometry and counters of the last align of THIS batch (finalize() must not read the engine's: another batch may have
	// been aligned on the same engine in between)
	int32_t last_grid = 0, n_retries = 0;
	// debug band trace (tests)
	int32_t debug_pair = -1;
};

namespace {

#define HIP_TRY(g, call)                                                              \
	do {                                                                              \
		hipError_t e_ = (call);                                                       \
		if (e_ != hipSuccess) {                                                       \
			(g)->err = std::string(#call) + ": " + hipGetErrorString(e_);             \
			return -1;                                                                \
		}                                                                             \
	} while (0)

void account(mwf_gpu_t *g, int64_t delta)
{
	g->dev_bytes += delta;
	g->dev_bytes_peak = std::max(g->dev_bytes_peak, g->dev_bytes);
}

int ensure(mwf_gpu_t *g, DevBuf &b, size_t bytes)
{
	if (bytes <= b.bytes) return 0;
	if (b.p) {
		HIP_TRY(g, hipStreamSynchronize(g->stream));
		HIP_TRY(g, hipFree(b.p));
		account(g, -(int64_t)b.bytes);
		b.p = nullptr, b.bytes = 0;
	}
	size_t want = bytes < ((size_t)1 << 30) ? bytes + bytes / 8 + 256 : bytes; // small buffers get slack so they rarely regrow
	hipError_t e = hipMalloc(&b.p, want);
	if (e != hipSuccess) {
		(void)hipGetLastError();
		want = bytes;
		e = hipMalloc(&b.p, want);
	}
	if (e != hipSuccess) {
		b.p = nullptr;
		g->err = "hipMalloc of " + std::to_string(bytes) + " bytes failed: " + hipGetErrorString(e);
		return -1;
	}
	b.bytes = want;
	account(g, (int64_t)want);
	return 0;
}

void release(mwf_gpu_t *g, DevBuf &b)
{
	if (b.p) {
		(void)hipFree(b.p);
		account(g, -(int64_t)b.bytes);
	}
	b.p = nullptr, b.bytes = 0;
}

// A batch allocation: the engine's spare one when it is large enough, else a fresh hipMalloc.
int take_block(mwf_gpu_t *g, DevBuf &spare, DevBuf &out, size_t bytes)
{
	if (spare.p && spare.bytes >= bytes) {
		out = spare;
		spare = DevBuf{};
		return 0;
	}
	release(g, spare);
	out = DevBuf{};
	return ensure(g, out, std::max<size_t>(bytes, 4096));
}

// ... and back: the engine keeps the larger of the two
void give_block(mwf_gpu_t *g, DevBuf &spare, DevBuf &b)
{
	if (!b.p) return;
	if (!spare.p || spare.bytes < b.bytes) std::swap(spare, b);
	release(g, b);
}

// ---- pinned staging ------------------------------------------------------------------------------------------------

int pin_reserve(mwf_gpu_t *g, size_t half)
{
	half = std::min(std::max<size_t>(align_up(half, 4096), (size_t)64 << 10), kPinHalfMax);
	if (g->pin && g->pin_half >= half) return 0;
	HIP_TRY(g, hipStreamSynchronize(g->stream));
	g->pin_busy[0] = g->pin_busy[1] = false;
	if (g->pin) (void)hipHostFree(g->pin);
	g->pin = nullptr, g->pin_half = 0;
	HIP_TRY(g, hipHostMalloc(&g->pin, 2 * half, hipHostMallocDefault));
	g->pin_half = half;
	for (hipEvent_t &e : g->pin_ev)
		if (!e) HIP_TRY(g, hipEventCreateWithFlags(&e, hipEventDisableTiming));
	return 0;
}

struct Seg { const void *src; size_t len; }; // src == nullptr: `len` zero bytes

// memcpy into the pinned staging buffer; megabytes at a time go on a few host threads (one thread moves ~8-10 GB/s: the 20 MB of a
// 1024 x 10 kb batch took 1.2 ms of its 1.9 ms upload)
void par_memcpy(char *dst, const char *src, size_t n)
{
	if (n < ((size_t)2 << 20)) { memcpy(dst, src, n); return; }
	const size_t n_th = std::min<size_t>(4, std::min<size_t>(std::max(1u, std::thread::hardware_concurrency()), n >> 20));
	if (n_th <= 1) { memcpy(dst, src, n); return; }
	const size_t share = (n / n_th + 63) & ~(size_t)63;
	std::vector<std::thread> th;
	for (size_t k = 1; k < n_th; ++k) {
		const size_t at = k * share;
		if (at >= n) break;
		th.emplace_back([=]() { memcpy(dst + at, src + at, std::min(share, n - at)); });
	}
	memcpy(dst, src, std::min(share, n));
	for (std::thread &t : th) t.join();
}

// The concatenation of `segs` to device memory at `dst`: packed into the pinned halves by the host while the previous
// half is on its way.  One copy for a call whose inputs fit a half — and then the call does not wait for it: the sources have
// been read, everything that uses `dst` is ordered behind the copy on the engine's stream, and the half is only written again
// once its event has fired (a single short pair: 17 us of upload down to the packing and the enqueue).  A longer upload
// returns after its last copy completed.
int upload_segments(mwf_gpu_t *g, char *dst, const std::vector<Seg> &segs)
{
	size_t total = 0;
	for (const Seg &s : segs) total += s.len;
	if (total == 0) return 0;
	if (pin_reserve(g, total)) return -1;
	const size_t half = g->pin_half;
	size_t si = 0, so = 0, done = 0;
	const bool one_copy = total <= half;
	for (int h = g->pin_busy[0] && !g->pin_busy[1] ? 1 : 0; done < total; h ^= 1) {
		char *buf = (char*)g->pin + (size_t)h * half;
		if (g->pin_busy[h]) {
			HIP_TRY(g, hipEventSynchronize(g->pin_ev[h]));
			g->pin_busy[h] = false;
		}
		size_t fill = 0;
		while (fill < half && si < segs.size()) {
			const size_t take = std::min(half - fill, segs[si].len - so);
			if (segs[si].src) par_memcpy(buf + fill, (const char*)segs[si].src + so, take);
			else memset(buf + fill, 0, take);
			fill += take, so += take;
			if (so == segs[si].len) ++si, so = 0;
		}
		HIP_TRY(g, hipMemcpyAsync(dst + done, buf, fill, hipMemcpyHostToDevice, g->stream));
		HIP_TRY(g, hipEventRecord(g->pin_ev[h], g->stream));
		g->pin_busy[h] = true;
		done += fill;
	}
	if (one_copy) return 0;
	HIP_TRY(g, hipStreamSynchronize(g->stream));
	g->pin_busy[0] = g->pin_busy[1] = false;
	return 0;
}

// `bytes` from device memory into host memory at `dst`, through the pinned buffer when they fit one half
int download(mwf_gpu_t *g, void *dst, const void *src, size_t bytes)
{
	if (bytes == 0) return 0;
	if (bytes <= kPinHalfMax && pin_reserve(g, bytes) == 0) {
		// (an upload still on its way out of the pinned buffer is ahead of this copy on the stream)
		HIP_TRY(g, hipMemcpyAsync(g->pin, src, bytes, hipMemcpyDeviceToHost, g->stream));
		HIP_TRY(g, hipStreamSynchronize(g->stream));
		g->pin_busy[0] = g->pin_busy[1] = false;
		memcpy(dst, g->pin, bytes);
		return 0;
	}
	HIP_TRY(g, hipStreamSynchronize(g->stream));
	HIP_TRY(g, hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
	return 0;
}

// ---- penalties, kernel choice -----------------------------------------------------------------------------------------

Penalty make_penalty(const mwf_opt_t &o)
{
	Penalty p;
	p.x = o.x, p.o1 = o.o1, p.o2 = o.o2, p.e1 = o.e1, p.e2 = o.e2;
	p.oe1 = o.o1 + o.e1, p.oe2 = o.o2 + o.e2;
	int32_t mp = std::max(p.x, std::max(p.oe1, p.oe2)); // reference miniwfa.c:390-392
	p.nH = mp + 1, p.n1 = p.e1 + 1, p.n2 = p.e2 + 1;
	return p;
}

const char *validate(const mwf_opt_t &o)
{
	if (o.x < 1 || o.e1 < 1 || o.e2 < 1) return "x, e1 and e2 must be >= 1 (a zero lag would make a wavefront depend on itself)";
	if (o.o1 < 0 || o.o2 < 0) return "gap-open penalties must be >= 0";
	if (std::max<int64_t>(o.x, std::max<int64_t>((int64_t)o.o1 + o.e1, (int64_t)o.o2 + o.e2)) + 1 > kBigRing) return "max(x, o1+e1, o2+e2) must be < 4096";
	if (o.step < 0) return "step must be >= 0";
	return nullptr;
}

// Upper bound on the optimal penalty: delete the whole target, insert the whole query.
int64_t penalty_bound(const mwf_opt_t &o, int64_t tl, int64_t ql, bool honour_max_s)
{
	auto gap = [&](int64_t L) -> int64_t { return L == 0 ? 0 : std::min<int64_t>(o.o1 + L * o.e1, o.o2 + L * o.e2); };
	int64_t b = gap(tl) + gap(ql);
	// the core pass stops one penalty after max_s (miniwfa.c:422); the low-memory first pass never stops (:569-589)
	if (honour_max_s && o.max_s > 0) b = std::min<int64_t>(b, (int64_t)o.max_s + 1);
	return b;
}

// widest window the 64-, 128- and 256-thread packed band variants are chosen for: (waves x 3 chunks - 1) x 256 - 64 columns
constexpr int64_t kBandMicroWindow = (1 * 3 - 1) * 256 - 64, kBandTinyWindow = (2 * 3 - 1) * 256 - 64, kBandSmallWindow = (4 * 3 - 1) * 256 - 64;
constexpr int64_t kBandWideWindow = (8 * 3 - 1) * 256 - 64;
// The 512-thread geometry with FOUR chunk slots per wave (32 chunks, 119 VGPRs, still two workgroups per CU; 2-bit sequence copies only): 2 % slower than
// three slots on windows those hold (1024 x 10 kb @ 5 %: 17.7 against 17.4 ms) — and 17.7 against 24.8 ms on a batch in which ONE pair outgrows them
// late and is re-run alone (three of four seeds of that batch shape, profiles/r04/wide4.txt).  Taken when a forecast or the batch's last align says so.
constexpr int64_t kBandWide4Window = (8 * 4 - 1) * 256 - 64;
// ... and the 1024-thread span geometry (16 waves x kBand2SpanK chunks, offsets biased by the target length: mwf_band2.hip wide_bias)
constexpr int64_t kBandSpanMaxSeq = 62000;
inline int64_t band_span_window() { return ((int64_t)band2_span_chunks() - 1) * 256 - 64; }

struct Plan {
	int kind = 0;              // 0: generic kernel, 2: band kernel
	BandGeom band{0, 0, 0, 0, 0, 0};
	int block = 256, grid = 1;
	int32_t W = 0, GW = 0;
	int64_t ring_slot_ints = 0, rows_slot = 0, tb_slot_bytes = 0, cig_scratch_slot = 0;
	int64_t snap_slot_ints = 0, snap_meta_slot = 0, seg_slot = 0;
	bool low_mem = false, cigar = false;
};

// Which kernel serves a set of pairs.  The band kernel keeps E/F in registers and therefore only holds windows up to
// its span; it has no low-memory first pass.  kind: -1 automatic, 0 generic, 2 band.
void choose_kernel(const mwf_gpu_t *g, const mwf_opt_t &opt, const Penalty &P, int64_t max_len, int64_t max_bound,
                   int64_t max_seq_lds, int64_t max_tl, int want_kind, Plan &pl, int geom_block = 0, int64_t window_hint = 0)
{
	pl.kind = 0;
	const bool low_mem = (opt.flag & MWF_F_CIGAR) && opt.step > 0;
	// packed band kernel (mwf_band2.hip): 16-bit offsets
	const bool can_packed = band2_supported(P) && g->band_pack != 0;
	if (geom_block == 32 && want_kind != 0 && !low_mem && lane_supported(P)) { // the short-pair class: one wave per pair, one diagonal per lane
		// The rows of all chunks are allocated whatever the window does, and LDS is what bounds the waves per CU (four chunks with the default
		// penalties: 14.5 KB, eleven waves; three: thirteen).  Three hold penalties up to ~110: 40 000 x 150 bp @ 5 % 0.68 against 0.79 ms with one
		// pair re-run, 20 000 x 200 bp 0.52 / 0.61 with ten, 20 000 x 150 bp @ 10 % 0.75 / 0.93 with 499 (profiles/r03/lane_kernel_probe.txt).
		const int chunks = g->lane_chunks > 0 ? g->lane_chunks : max_len <= 400 ? 3 : 4;
		// 2-bit sequence copies from ~450 bases of target + query on (measured: 20 000 x 250 bp 0.624 against 0.650 ms with byte copies, 40 000 x
		// 150 bp 0.664 against 0.617 — packing costs more than the shorter extension trips save; profiles/r04/short_reads_step.txt)
		BandGeom lg{64, 1, 64 * chunks, lane_lds_bytes(P, chunks, max_seq_lds), (g->seq2bit != 0 && !g->acgt_off_once && max_len >= 450) ? 1 : 0, 1};
		if (lg.lds_bytes <= 60 * 1024) { // (deep rings — large gap-open costs — with a raised lane_max_len: the band classes below take the pairs)
			pl.kind = 2, pl.band = lg;
			return;
		}
	}
	if (geom_block == 33 && want_kind != 0 && !low_mem && mid_supported(P)) { // a few mid-size pairs: one workgroup per pair, every ring in LDS (mwf_mid.hip)
		// the span: as many 64-column groups as the LDS holds beside the sequences (a window is about twice the final penalty wide: a 2 kb
		// pair at 5 % needs ~1100 columns), never more than the widest possible window plus the dead margins
		const int64_t want_cols = (std::min<int64_t>(max_len + 1, 2 * max_bound + 3) + 2 * P.nH + 63) / 64 * 64;
		int groups = (int)std::min<int64_t>(want_cols / 64, 128);
		while (groups > 1 && mid_lds_bytes(P, groups, max_seq_lds) > 158 * 1024) --groups;
		const int lds = mid_lds_bytes(P, groups, max_seq_lds);
		if (lds <= 158 * 1024) {
			// eight waves while the windows stay below ~700 columns (pairs of up to ~1.2 kb at 5 %), else sixteen (measured: 1 kb 0.255 against
			// 0.314 ms, 2 kb 0.572 / 0.556, 4 kb 1.60 / 1.43; profiles/mid_kernel_probe.py)
			const int block = g->mid_block ? g->mid_block : (max_len <= 1000 ? 256 : max_len <= 2500 ? 512 : 1024); // (4 x 400 bp: 131 us on four waves, 145 on eight)
			const int seq2 = g->seq2bit != 0 && !g->acgt_off_once;
			pl.kind = 2, pl.band = BandGeom{block, 1, 64 * groups, lds, seq2, 2};
			return;
		}
	}
	if (want_kind == 0 || low_mem || !can_packed) return;
	if (geom_block == 514) { // the 512-thread geometry with four chunk slots on biased offsets (the caller checked the lengths: kBandSpanMaxSeq); 2-bit copies only
		const int64_t need_lds = ((max_len >> 4) + 4) * 4;
		if (g->seq2bit == 0 || g->acgt_off_once || need_lds > 70 * 1024) return;
		// (five chunk slots per wave while target + query stay below 3.5 of that span, else six)
		const int chunks = band2_biased512_chunks() + (max_len + 1 <= 7 * (int64_t)(band2_biased512_chunks() * 256) / 2 ? 0 : 8);
		pl.kind = 2, pl.band = BandGeom{512, 2, chunks * 256, (int)((need_lds + 15) / 16 * 16), 1, 0}; // (packed 2: the copy that computes on biased offsets)
		return;
	}
	if (geom_block == 1024) { // the span geometry (the caller checked the lengths of every pair: kBandSpanMaxSeq); 2-bit sequence copies only
		const int64_t need_lds = ((max_len >> 4) + 4) * 4;
		if (g->band_span == 0 || g->seq2bit == 0 || g->acgt_off_once || need_lds > 150 * 1024) return;
		pl.kind = 2, pl.band = BandGeom{1024, 1, (int)band2_span_chunks() * 256, (int)((need_lds + 15) / 16 * 16), 1, 0};
		return;
	}
	// (window_hint: pairs a kernel handed back early come with the window they are expected to need, dev::window_forecast — the re-run
	// takes the class that fits that, not the one that fits the worst case)
	const int64_t max_window = window_hint > 0 ? std::min<int64_t>(std::min<int64_t>(max_len + 1, 2 * max_bound + 3), window_hint) : std::min<int64_t>(max_len + 1, 2 * max_bound + 3);
	BandGeom bg;
	bg.packed = 0, bg.seq2 = 0, bg.lane = 0;
	// Packed variants (E/F registers as int16 pairs): valid when no offset (a target index, plus at most one per penalty for
	// offsets that ran past the matrix) and no penalty count can reach 32767.  They halve the state registers, which is
	// what lets several workgroups share a CU — one pair's barrier phase then overlaps another's compute:
	//   window <=  448:  64 threads x 3 chunks, sixteen pairs per CU (twelve with traceback): short reads
	//   window <= 1216: 128 threads x 3 chunks, eight pairs per CU (six with traceback)
	//   window <= 2752: 256 threads x 3 chunks, four (three)
	//   wider:          512 x 3, two per CU — with traceback too (35.4 ms on the 1024 x 10 kb batch with ~100 bytes of scratch per
	//                   lane, against 43.4 ms for 768 x 2 with one workgroup per CU)
	// (measured alternatives on the 1024 x 10 kb batch: 1024 threads x 2 chunks spills and runs 50 ms, 768 x 2 42 ms; 768 x 2 at two
	// workgroups per CU — 80 VGPRs, six spilled — 18.2 against 17.8 ms, round 4)
	// Pairs whose offsets do not fit 16 bits take the generic kernel — with 16-bit ring rows where those apply, else 32-bit rows.  (The
	// unpacked band kernel of round 1, mwf_band.hip, lost to it wherever both applied — 512 x 20 kb @ 1 %: 5.4 against 8.0 ms, 512 x 50 kb
	// @ 0.3 %: 4.2 / 5.9, and 1024 x 12 kb @ 5 %: 42 against 101 ms with its window overflows re-run, profiles/r03/mid_pairs_kernels.txt —
	// and was removed in round 4.)
	const bool range_ok = max_tl + max_bound < 32767;
	if (!range_ok) return;
	bg.packed = 1;
	bg.block = max_window <= kBandMicroWindow ? 64 : max_window <= kBandTinyWindow ? 128 : max_window <= kBandSmallWindow ? 256 : 512;
	// forced geometry (tests, tuning)
	if (g->block == 64 || g->block == 128 || g->block == 256 || g->block == 512 || g->block == 768) bg.block = g->block;
	// geometry picked by the caller for a size class (pairs short enough that their window should stay inside a small span)
	if (g->block == 0 && (geom_block == 64 || geom_block == 128 || geom_block == 256)) bg.block = geom_block;
	bg.span = bg.block / 64 * (bg.block != 768 ? 3 : 2) * 256;
	const bool four_slots = bg.block == 512 && g->block == 0 && window_hint > kBandWideWindow && window_hint <= kBandWide4Window;
	if (four_slots) bg.span = 512 / 64 * 4 * 256;
	if (want_kind != 2 && max_len + 1 > 4 * (int64_t)bg.span) return; // windows will mostly outgrow the span: go generic at once
	const int64_t lds_cap = bg.block >= 768 ? 140 * 1024 : bg.block >= 512 ? 70 * 1024 : bg.block == 256 ? 36 * 1024 : bg.block == 128 ? 18 * 1024 : 9 * 1024;
	// the packed kernel's sequence copy holds 2 bits per base unless that is switched off (or this is the re-run of pairs that
	// are not plain ACGT): a quarter of the LDS, half the LDS instructions per probe
	const bool seq2 = g->seq2bit != 0 && !g->acgt_off_once;
	const int64_t need_lds = seq2 ? ((max_len >> 4) + 4) * 4 : max_seq_lds;
	bg.lds_bytes = need_lds <= lds_cap ? (int)((need_lds + 15) / 16 * 16) : 0;
	bg.seq2 = seq2 && bg.lds_bytes > 0;
	// byte-wise copy (pairs outside plain ACGT) with wide windows: three slots of state plus six probe words per column do not fit the 128
	// VGPRs two 512-thread workgroups per CU leave each wave (~500 bytes of scratch); 768 x 2 holds the same 24 chunks without spilling
	if (!bg.seq2 && bg.block == 512 && four_slots) bg.span = 512 / 64 * 3 * 256; // (no byte-wise form of the four-slot geometry)
	if (!bg.seq2 && bg.block == 512 && g->block == 0 && max_seq_lds <= 140 * 1024) {
		bg.block = 768, bg.span = 768 / 64 * 2 * 256;
		bg.lds_bytes = (int)((max_seq_lds + 15) / 16 * 16);
	}
	if (bg.lds_bytes == 0) return; // the packed kernel keeps the sequences in LDS: what does not fit takes the generic kernel
	pl.kind = 2, pl.band = bg;
}

// resident workgroups per CU of a kernel variant (one runtime query per variant and engine)
int cached_occupancy(mwf_gpu_t *g, const Penalty &P, const Plan &pl, int lds_e2_cols, bool stream_pass, bool ring16 = false)
{
	uint64_t key;
	if (pl.kind == 2)
		key = 1ull | (uint64_t)pl.band.block << 4 | (uint64_t)(pl.band.packed == 1) << 16 | (uint64_t)(pl.band.packed == 2) << 15 | (uint64_t)(pl.band.lds_bytes > 0) << 17 | (uint64_t)pl.cigar << 18 |
		      (uint64_t)(pl.band.seq2 != 0) << 3 | (uint64_t)(pl.band.lane == 1) << 2 | (uint64_t)(pl.band.block == 512 && pl.band.span > 6144) << 1 | (uint64_t)(pl.band.block == 512 ? pl.band.span / 2048 : 0) << 56 | (uint64_t)(pl.band.lane == 2) << 19 | (uint64_t)P.e1 << 20 | (uint64_t)P.e2 << 28 | (uint64_t)pl.band.lds_bytes << 36;
	else key = 2ull | (uint64_t)pl.block << 4 | (uint64_t)stream_pass << 16 | (uint64_t)ring16 << 17 | (uint64_t)(P.nH > kMaxRing) << 18 | (uint64_t)lds_e2_cols << 20;
	auto it = g->occ_cache.find(key);
	if (it != g->occ_cache.end()) return it->second;
	const int per = pl.kind == 2 && pl.band.lane == 2 ? 1 // (mwf_mid.hip: most of a CU's LDS per workgroup)
	              : pl.kind == 2 && pl.band.lane ? lane_kernel_occupancy(pl.band.lds_bytes, pl.cigar)
	              : pl.kind == 2 ? band2_kernel_occupancy(P, pl.band, pl.cigar)
	              : P.nH > kMaxRing ? bigring_kernel_occupancy()
	                             : batch_kernel_occupancy(pl.block, stream_pass, lds_e2_cols, ring16);
	g->occ_cache[key] = per;
	return per;
}

int64_t tb_budget_bytes(mwf_gpu_t *g)
{
	if (g->tb_budget_mb > 0) return g->tb_budget_mb << 20;
	size_t fr = 0, tot = 0;
	if (hipMemGetInfo(&fr, &tot) != hipSuccess) fr = (size_t)8 << 30, tot = fr;
	// leave a fifth of what is free right now alone, and never claim more than a quarter of the device for one engine
	// (the drop-in API is re-entrant: other host threads have engines of their own); the arena is kept between calls
	int64_t b = (int64_t)(fr / 5 * 4) + (int64_t)g->tb.bytes;
	return std::min<int64_t>(b, std::min<int64_t>((int64_t)64 << 30, (int64_t)(tot / 4)));
}

// Per device: one-workgroup-per-pair launches take it shared (just around the launch), the whole-device kernel takes it
// exclusively for its whole run and first waits for everything already running on the device (any engine's stream), so that
// no other kernel of this process holds CUs while its workgroups wait for one another.
std::shared_mutex g_dev_gate[kMaxDevices];

// Run the one-workgroup-per-pair kernel over `n_items` pairs given by d_order (device) on at most `slots` workgroups.
// tb_total_budget < 0: the traceback budget is looked up here, and only when the arena has to grow.
int run_batch_kernel(mwf_gpu_t *g, mwf_gpu_batch_t *b, const mwf_opt_t &opt, const int32_t *d_order, int32_t n_items,
                     int slots, int64_t max_len, int64_t max_bound, int64_t max_bound1, bool timed,
                     int want_kind, int64_t max_tl, int64_t max_seq_lds, int timed_end, int geom_block, int *ran_kind, int64_t window_hint = 0)
{
	const Penalty P = make_penalty(opt);
	Plan pl;
	pl.cigar = (opt.flag & MWF_F_CIGAR) != 0;
	pl.low_mem = pl.cigar && opt.step > 0;
	// generic kernel: four waves per pair, eight once the windows are wide (measured on 1250 x 50 kb: 708 ms against 782 ms)
	pl.block = g->block > 0 && g->block != 768 ? g->block : (std::min<int64_t>(max_len + 1, 2 * max_bound + 3) >= 8192 ? 512 : 256);
	choose_kernel(g, opt, P, max_len, max_bound, max_seq_lds, max_tl, want_kind >= 0 ? want_kind : g->force_kind, pl, geom_block, window_hint);
	// `slots` is an upper bound from the caller (retries ask for fewer, larger slots); the chosen kernel's own residency
	// bounds it as well
	int per_cu, lds_e2_cols = 0;
	bool ring16 = false;
	if (pl.kind == 2) {
		pl.block = pl.band.block;
		per_cu = g->slots_per_cu > 0 ? g->slots_per_cu : cached_occupancy(g, P, pl, 0, false);
	} else {
		// wide windows (the 512-thread choice above), default gap extension: E2/F2 stay in LDS while the window fits 16 k columns
		if (g->lds_e2 && pl.block == 512 && g->block == 0 && P.e2 == 1 && !g->scalar_generic && !pl.low_mem && P.nH <= kMaxRing) {
			lds_e2_cols = 16384;
			// 16-bit ring rows halve the traffic of this HBM-bound kernel.  An offset is a target index (or runs past the matrix by
			// at most one per penalty), so they hold while target length + penalty < 65530: taken optimistically for pairs whose
			// penalty would have to exceed an eighth of their length to break that; a pair that does comes back as
			// ST_BAND_OVERFLOW and is re-run with 32-bit rows.  The LDS copy of E2/F2 is coded the same way: 64 KB instead of 128.
			// Round 2 took them only for batches of at least as many pairs as CUs (with fewer the decoding and coding of the rows was pure
			// overhead: 64 x 50 kb 88 ms against 73).  With the recurrence on the packed codes and 2-bit sequence copies the 16-bit kernel is
			// the faster one at every batch size (profiles/ring16_small_batches.py: 8 pairs 63.8 against 67.4 ms, 64: 67.9 / 71.3, 200: 72.6 /
			// 90.7; with traceback 67.6 / 77.8 ... 71.9 / 103.9): taken whenever the offsets fit.
			// (round 5: where the batch's divergence is known — estimate_divergence — the penalty is guessed from it, ~5.2 per diverged base
			// with the default costs plus a third: 50 kb pairs at 5 % reach penalty 12 500 and never fitted, 64 of 64 were run twice)
			const int64_t s_guess = b->div_est > 0 && g->div_aware ? (int64_t)(6.9 * b->div_est * (double)max_tl) + 256 : max_len / 8;
			ring16 = g->ring16 != 0 && !g->ring16_off_once && max_tl + std::max<int64_t>(s_guess, max_len / 8) < 65500;
			// 32-bit: one workgroup per CU either way (128 KB of LDS): twelve waves fit its 168-VGPR budget, 490 ms against 519 ms with eight
			// 16-bit: 512 threads, two workgroups per CU (64 KB of LDS each, 128 VGPRs) — 354 ms on 1250 x 50 kb against 375 ms for
			// 768 threads and one per CU (479 ms with 32-bit rows); with traceback the 512-thread copy spills too much: 768 (451 against 477 ms)
			pl.block = ring16 ? (g->ring16_block ? g->ring16_block : (pl.cigar ? 768 : 512)) : 768;
		}
		if (P.nH > kMaxRing) pl.block = 256; // the big-ring form of the generic kernel (launch_batch): one column per lane, 256 threads
		per_cu = g->slots_per_cu > 0 ? g->slots_per_cu : cached_occupancy(g, P, pl, lds_e2_cols, !g->scalar_generic, ring16);
	}
	slots = std::max(1, std::min(slots, g->n_cu * std::max(1, per_cu)));
	if (P.nH > kMaxRing) {
		// a deep ring is (nH + 2 n1 + 2 n2) rows of tl+ql columns per resident workgroup (times two in low-memory mode): fewer
		// workgroups rather than a tenth of the device in rings
		const int64_t per_slot = (int64_t)(P.nH + 2 * P.n1 + 2 * P.n2) * ((max_len + 3 + 255) / 256 * 256 + 512) * 4 * (pl.low_mem ? 2 : 1);
		slots = (int)std::max<int64_t>(1, std::min<int64_t>(slots, (int64_t)(g->total_mem / 10) / std::max<int64_t>(per_slot, 1)));
	}
	if (getenv("MWF_DEBUG"))
		fprintf(stderr, "[libmwf_hip] kernel kind %d: block %d packed %d lds %d B, %d workgroup(s) per CU, %d slots, %d pairs\n", pl.kind, pl.block,
		        pl.band.packed, pl.band.lds_bytes, per_cu, slots, n_items);
	pl.grid = std::max(1, std::min<int>(slots, n_items));
	// row stride: whole 256-column chunks plus room for the band kernel's neighbour loads past the last chunk
	pl.W = (int32_t)((max_len + 3 + 255) / 256 * 256 + 512);
	pl.GW = pl.W / 64 + 2;
	pl.ring_slot_ints = (int64_t)(P.nH + 2 * P.n1 + 2 * P.n2) * pl.W;
	if (pl.kind == 2 && pl.band.lane) pl.ring_slot_ints = 64; // its rings are in LDS
	const size_t S = (size_t)pl.grid;

	if (ensure(g, g->ring, S * pl.ring_slot_ints * 4)) return -1;
	if (ensure(g, g->good, S * (size_t)P.nH * pl.GW * 8)) return -1;
	if (pl.cigar) {
		pl.rows_slot = max_bound + 2;
		pl.cig_scratch_slot = max_len + 2;
		int64_t worst = (max_bound + 1) * (max_len + 1); // every penalty as wide as the whole matrix
		if (pl.low_mem && opt.step > 2 * P.nH) {
			// the second pass collapses the band to one diagonal at every checkpoint (miniwfa.c:413-416) and consecutive
			// checkpoints are at most step+nH penalties apart, so a row is never wider than about 2*(step+nH)
			worst = std::min(worst, (max_bound + 1) * std::min<int64_t>(max_len + 1, 2 * (int64_t)(opt.step + 2 * P.nH) + 8));
		}
		worst += 8 * (max_bound + 2);
		if (pl.kind == 2 && pl.band.lane == 1) worst = (std::min<int64_t>(max_bound, 256) + 2) * pl.band.span; // its rows: the span wide, fewer than 256 of them
		if (pl.kind == 2 && pl.band.lane == 2) worst = (max_bound + 2) * pl.band.span;                          // rows of the span's width, one per penalty
		// the device is only asked how much is free when the arena at hand cannot hold the worst case
		int64_t per = worst;
		if ((int64_t)g->tb.bytes < (int64_t)S * worst || g->tb_budget_mb > 0) per = std::min(per, std::max<int64_t>(tb_budget_bytes(g), (int64_t)g->tb.bytes) / (int64_t)S);
		if (g->tb_budget_mb > 0) per = std::min(per, (g->tb_budget_mb << 20) / (int64_t)S);
		pl.tb_slot_bytes = std::max<int64_t>(4096, per) / 4 * 4; // rows are padded to dwords
		if (ensure(g, g->tb, S * (size_t)pl.tb_slot_bytes)) return -1;
		if (ensure(g, g->row_off, S * (size_t)pl.rows_slot * 8)) return -1;
		if (ensure(g, g->row_lo, S * (size_t)pl.rows_slot * 4)) return -1;
		if (ensure(g, g->cig_scratch, S * (size_t)pl.cig_scratch_slot * 4)) return -1;
	}
	if (pl.low_mem) {
		const int64_t NS = P.nH + 2 * P.n1 + 2 * P.n2;
		const int64_t n_snap_max = max_bound1 / opt.step + 2; // first pass: bound without max_s
		pl.seg_slot = n_snap_max;
		pl.snap_meta_slot = n_snap_max * (4 + 4 * NS);
		// a snapshot holds every array-slice of the shadow ring; windows are at most min(2s+1, whole matrix) wide
		int64_t worst = 0;
		for (int64_t j = 1; j <= n_snap_max; ++j)
			worst += NS * std::min<int64_t>(max_len + 1, 2 * j * opt.step + 3);
		const int64_t budget = (int64_t)(std::max<int64_t>(tb_budget_bytes(g), (int64_t)g->snap.bytes) / 4 / (int64_t)S);
		pl.snap_slot_ints = std::max<int64_t>(1024, std::min(worst, budget));
		if (ensure(g, g->sring, S * pl.ring_slot_ints * 4)) return -1;
		if (ensure(g, g->snap, S * (size_t)pl.snap_slot_ints * 4)) return -1;
		if (ensure(g, g->snap_meta, S * (size_t)pl.snap_meta_slot * 4)) return -1;
		if (ensure(g, g->seg, S * (size_t)pl.seg_slot * 8)) return -1;
	}

	BatchArgs a;
	memset(&a, 0, sizeof(a));
	a.seqs = b->d_seqs, a.t_off = b->d_t_off, a.q_off = b->d_q_off, a.tl = b->d_tl, a.ql = b->d_ql;
	a.order = d_order, a.n_pairs = n_items;
	// A launch of one workgroup per pair on the kernels that take it (lane, mid, packed band) needs no work counter: workgroup i aligns
	// pair i.  Otherwise a fresh counter: the first kQueueSlots launches of an align call use the ones its reset kernel zeroed.
	// The lane kernel takes a set of 64 counters (kLaneCounters above).
	if (pl.kind == 2 && pl.band.lane == 1 && n_items > pl.grid) {
		a.queue_parts = kLaneCounters;
		if (g->queue_clean && g->lane_set_next < kLaneSets) a.queue = (int32_t*)g->queue.p + kQueueSlots + (g->lane_set_next++) * kLaneCounters * kLaneStride;
		else {
			a.queue = (int32_t*)g->queue.p + kQueueSlots;
			HIP_TRY(g, hipMemsetAsync(a.queue, 0, (size_t)kLaneCounters * kLaneStride * 4, g->stream)); // (stream order: the launch that used it last is complete by then)
		}
	} else if (pl.kind == 2 && pl.band.lane == 1) a.queue = nullptr;
	else if (pl.kind == 2 && (pl.band.lane || pl.band.packed) && n_items <= pl.grid && !g->queue_clean) a.queue = nullptr;
	else if (g->queue_clean && g->queue_next < kQueueSlots) a.queue = (int32_t*)g->queue.p + g->queue_next++;
	else {
		a.queue = (int32_t*)g->queue.p;
		HIP_TRY(g, hipMemsetAsync(g->queue.p, 0, 4, g->stream)); // (stream order: the launch that used it last is complete by then)
	}
	a.scalar_generic = g->scalar_generic;
	a.lds_e2_cols = lds_e2_cols;
	a.ring16 = ring16 ? 1 : 0;
	a.lane_chunks = pl.kind == 2 && pl.band.lane ? pl.band.span / 64 : 0;
	a.pen = P;
	a.want_cigar = pl.cigar ? 1 : 0;
	a.step = pl.low_mem ? opt.step : 0;
	a.max_s = opt.max_s, a.max_iter = opt.max_iter;
	a.debug_pair = b->debug_pair;
	a.ring = (int32_t*)g->ring.p;
	a.sring = pl.low_mem ? (int32_t*)g->sring.p : nullptr;
	a.ring_slot_ints = pl.ring_slot_ints, a.W = pl.W;
	a.good = (unsigned long long*)g->good.p, a.GW = pl.GW;
	a.tb = pl.cigar ? (uint8_t*)g->tb.p : nullptr, a.tb_slot_bytes = pl.tb_slot_bytes;
	a.row_off = pl.cigar ? (int64_t*)g->row_off.p : nullptr;
	a.row_lo = pl.cigar ? (int32_t*)g->row_lo.p : nullptr;
	a.rows_slot = pl.rows_slot;
	a.cig_scratch = pl.cigar ? (uint32_t*)g->cig_scratch.p : nullptr, a.cig_scratch_slot = pl.cig_scratch_slot;
	a.cig_pool = b->d_cig_pool, a.cig_head = b->d_cig_head, a.cig_pool_words = b->cig_pool_words;
	// block mode leaves up to one partly used block per workgroup behind; the pool's slack (batch_common) covers kCigBlockGrid of them per align —
	// one align can make several block-mode launches (size classes, byte-wise twins, re-runs): a launch the slack no longer covers takes words singly
	a.cig_block = 0;
	if (b->cig_block > 0 && pl.grid <= b->cig_block_left) a.cig_block = b->cig_block, b->cig_block_left -= pl.grid;
	a.report_wide = geom_block == 0 && window_hint == kBandWide4Window ? 1 : 0; // (the wide class's measuring align, mwf_gpu_batch_align)
	a.snap = pl.low_mem ? (int32_t*)g->snap.p : nullptr, a.snap_slot_ints = pl.snap_slot_ints;
	a.snap_meta = pl.low_mem ? (int32_t*)g->snap_meta.p : nullptr, a.snap_meta_slot = pl.snap_meta_slot;
	a.seg = pl.low_mem ? (int32_t*)g->seg.p : nullptr, a.seg_slot = pl.seg_slot;
	a.out_s = b->d_s, a.out_iter = b->d_iter, a.out_ncig = b->d_ncig, a.out_cigoff = b->d_cigoff;
	a.out_status = b->d_status, a.out_cells1 = b->d_cells1, a.out_dbg = b->d_dbg4;
	a.dbg = b->debug_pair >= 0 ? (int32_t*)g->dbg.p : nullptr;
	a.dbg_cap = b->debug_pair >= 0 ? (int32_t)(g->dbg.bytes / 8) : 0;

	// HIP events bracket the kernel only: every workspace allocation above is already done
	if (timed) HIP_TRY(g, hipEventRecord(g->ev0, g->stream));
	std::shared_lock<std::shared_mutex> gate(g_dev_gate[g->device % kMaxDevices]); // not while a whole-device kernel runs
	const int lrc = pl.kind == 2 && pl.band.lane == 2 ? launch_mid(a, pl.grid, pl.band.block, pl.band.lds_bytes, pl.band.seq2 != 0, g->stream)
	              : pl.kind == 2 && pl.band.lane ? launch_lane(a, pl.grid, pl.band.lds_bytes, pl.band.seq2 != 0, g->stream)
	              : pl.kind == 2 ? launch_band2(a, pl.grid, pl.band, g->stream)
	                             : launch_batch(a, pl.grid, pl.block, g->stream);
	gate.unlock();
	if (lrc != 0) {
		g->err = "kernel launch failed";
		return -1;
	}
	b->last_grid = std::max(b->last_grid, pl.grid);
	if (timed_end < 0 ? timed : timed_end != 0) { // the events bracket all launches of an align call, not the retries
		HIP_TRY(g, hipEventRecord(g->ev1, g->stream));
		g->ev_pending = true;
	}
	g->stats.n_launches += 1;
	g->stats.grid = std::max(g->stats.grid, pl.grid), g->stats.block = pl.block, g->stats.kernel_kind = pl.kind;
	g->stats.packed = pl.kind == 2 ? (pl.band.lane == 2 ? 33 : pl.band.lane ? 32 : (pl.band.packed ? 1 : 0)) : (ring16 ? 16 : 0);
	g->stats.lowmem_two_pass = pl.low_mem ? 1 : 0;
	if (ran_kind) *ran_kind = pl.kind;
	return 0;
}

// ---- whole-device kernel ------------------------------------------------------------------------------------------------

// Its workgroups wait for one another, so all of them must be resident: launches on one device are serialised process-wide
// (two host threads' engines would otherwise starve each other until the spin limit), and the call returns after the kernels
// completed.  Other processes' kernels can still hold CUs; that is what the bounded waits and the fallback are for.

int coop_grid_limit(mwf_gpu_t *g)
{
	if (g->coop_grid < 0) g->coop_grid = std::min(sys_max_grid(), g->n_cu);
	return g->coop_grid_cap > 0 ? std::min(g->coop_grid, g->coop_grid_cap) : g->coop_grid;
}

// Workgroups per pair when `n` pairs of at most `len` columns share the device: as many as the widest possible window can
// use when the pair is alone (it can then never outgrow them); when several pairs run side by side, as many as a window of
// a third of tl+ql needs (windows stay near a quarter at 3-5 % divergence) — a pair that does outgrow its group is re-run
// alone by finalize().  The per-penalty latency does not depend on the group size (C4-like 150 kb pair: 151 ms on 256
// workgroups, 141 ms on 64), so pairs side by side multiply the throughput.
int coop_group_size(int n_cu, int64_t len, bool alone, int ow = 256)
{
	int G = n_cu;
	// (the systolic kernel's slots own `ow` = 240 of their 256 columns, and a few chunks beyond the window take part)
	const int64_t chunks = alone ? len / ow + (ow == 256 ? 3 : 8) : len / ow / 3 + (ow == 256 ? 8 : 12);
	while (G > (alone ? 64 : 16) && chunks <= coop_chunk_slots(G / 2)) G /= 2;
	return G;
}

// Up to n_cu / group size pairs side by side on the whole-device kernel, each on its own group of workgroups (mwf_sys.hip).
// Everything is enqueued on the stream: first pass, and in low-memory mode the checkpoint walk over its traceback matrix
// and the second pass, then traceback + outputs.
int run_coop_group(mwf_gpu_t *g, mwf_gpu_batch_t *b, const mwf_opt_t &opt, const std::vector<int32_t> &pairs, int Gs, bool first, bool last)
{
	const Penalty P = make_penalty(opt);
	const bool cigar = (opt.flag & MWF_F_CIGAR) != 0, low_mem = cigar && opt.step > 0;
	const int n_groups = (int)pairs.size();
	if (n_groups < 1 || Gs < 1 || (int64_t)Gs * n_groups > coop_grid_limit(g)) { g->err = "whole-device kernel cannot be made resident"; return -1; }
	int64_t len = 0, bound = 0, bound1 = 0;
	bool traced = false;
	for (int32_t pair : pairs) {
		len = std::max<int64_t>(len, (int64_t)b->h_tl[pair] + b->h_ql[pair]);
		bound = std::max(bound, penalty_bound(opt, b->h_tl[pair], b->h_ql[pair], true));
		bound1 = std::max(bound1, penalty_bound(opt, b->h_tl[pair], b->h_ql[pair], false));
		traced |= b->debug_pair == pair;
	}
	const int32_t W = (int32_t)((len + 3 + 255) / 256 * 256 + 512), GW = W / 64 + 2;
	const int64_t TC = coop_chunk_slots(Gs);
	const size_t NG = (size_t)n_groups;
	// the systolic kernel (mwf_sys.hip) runs every pass but the provenance pass of the two-pass low-memory mode
	const bool use_sys = true;
	const int sysP = g->sys_p, sysP2 = g->sys_p2 > 0 && low_mem ? g->sys_p2 : g->sys_p;
	// Columns per lane of the systolic kernel, per pass: one column per lane (64-column slots that own 48) quarters the
	// per-column work a wave does per penalty — what a chain of penalties on a narrow window waits for — but needs five times
	// the slots; taken while the window is EXPECTED to fit them (first pass: a fifth of tl+ql, real pairs stay far below;
	// second pass of the low-memory mode: the band collapses at every checkpoint, miniwfa.c:413-416).  A pair whose window
	// outgrows the slots comes back as ST_BAND_OVERFLOW and is re-run with four columns per lane (finalize()).
	auto window_cap = [&](int c) -> int64_t { return (TC - 4) * (int64_t)sys_owned_cols(sysP, c) - 2 * (257 + sysP); };
	bool wide_again = false;
	for (int32_t pair : pairs) wide_again |= (b->h_flags[pair] & 16) != 0;
	// (first-pass window: a fifth of tl+ql — or, where the batch's divergence is known, 6.5 d (tl+ql): 50 kb pairs at 15 % outgrew the 64-column slots and ran twice)
	const int64_t est1 = std::min<int64_t>(len + 1, std::max<int64_t>(8192, b->div_est > 0 && g->div_aware ? (int64_t)(6.5 * b->div_est * (double)len) : len / 5));
	const int64_t est2 = std::min<int64_t>(len + 1, 2 * ((int64_t)opt.step + 2 * P.nH) + 8);
	const int c_first = g->sys_c ? g->sys_c : (!wide_again && est1 <= window_cap(1)) ? 1 : 4;
	const int c_second = g->sys_c ? g->sys_c : (!wide_again && est2 <= window_cap(1)) ? 1 : 4;
	// (the systolic kernel has private H rings per chunk slot, sys_ring below: no ring of whole rows — 1 GB for the 5 Mb pair — is allocated here any more)
	// Low-memory mode (opt.step > 0), two ways to the checkpoints:
	//   walk     — the first pass stores its whole traceback (s^2 bytes: 55 GB for the 5 Mb pair) and the checkpoints are
	//              read off it by walking the recorded choices back (fast while that fits the budget);
	//   two-pass — the reference's way (miniwfa.c:551-601): the first pass stores no traceback, carries provenance through
	//              shadow registers / rows and takes a snapshot every `step` penalties: a few GB for the 5 Mb pair.
	// Chosen by what the walk variant's arena would be against the budget ("lowmem_budget_mb", default 8 GB).
	bool two_pass = false;
	if (low_mem) {
		// automatic: 8 GB.  opt.step > 0 asks for the reference's low-memory mode (miniwfa.c:551-601; README.md:55-64: the 5 Mb MHC pair in 4 GB
		// instead of 50): a first pass that would hold more traceback than that takes the two-pass form — provenance carried through the
		// systolic kernel, snapshots of (nH + 2 e1 + 2 e2) array-slices every `step` penalties — and the device footprint stays within a few GB.
		const int64_t budget = g->lowmem_budget_mb > 0 ? g->lowmem_budget_mb << 20 : (int64_t)8 << 30;
		two_pass = std::max<int64_t>((int64_t)1 << 30, 6000 * len) * (int64_t)NG * g->coop_tb_mult * (c_first == 1 ? 12 : 9) / 8 > budget;
	}
	// granules crossing waves: [nH][TC][2 sides][4] x 8 bytes (twice for the two-pass mode: values and their provenance);
	// misc: flags, barrier words, pass state, then the flag ring
	const size_t gran_bytes = 4096; // (the per-penalty granule exchange of mwf_coop.hip is gone: the field remains for the layout of the misc block)
	const size_t flag_ring_bytes = 0; // (the flag ring of the removed per-penalty hand-off kernel)
	const size_t misc_bytes = 4096 + flag_ring_bytes;
	if (ensure(g, g->coop_edge, NG * gran_bytes)) return -1;
	if (ensure(g, g->coop_misc, NG * misc_bytes + 4096)) return -1; // (+ the pair ids behind the last group)
	int64_t rows_slot = 0, tb_bytes = 0, cig_scratch = 0, seg_slot = 0;
	if (cigar) {
		rows_slot = std::max(bound, bound1) + 2;
		cig_scratch = len + 2;
		// Arena: the worst case (every row as wide as the matrix) is out of reach for long pairs, so start from a cap that
		// holds the real ones (s^2 bytes: 51 GB for the MHC pair) and let finalize() double it after an overflow.  An
		// arena that is already large enough is reused as is, so repeated calls never re-allocate.
		// (first guess: 6000 bytes per column of the matrix' perimeter — 1.8 GB for a 150 kb pair that needs 0.7, 60 GB for the
		// 5 Mb pair that needs 55: a traceback of s^2 bytes with s about 2.5 % of tl+ql)
		const int64_t worst = NG * (rows_slot + 1) * (len + 8);
		int64_t guess = std::min<int64_t>(g->coop_tb_cap, std::max<int64_t>((int64_t)1 << 30, 6000 * len) * (int64_t)NG);
		// two-pass: only the second pass stores traceback, and its rows are at most about 2*(step+nH) wide (the band collapses
		// to one diagonal at every checkpoint, miniwfa.c:413-416); s is guessed as 3 % of tl+ql
		if (two_pass) guess = std::min<int64_t>(guess, std::max<int64_t>((int64_t)64 << 20, (len * 3 / 100 + 1024) * std::min<int64_t>(len + 1, 2 * (int64_t)(opt.step + 2 * P.nH) + 8)) * (int64_t)NG);
		// (the systolic kernel stores 256 bytes per penalty and chunk slot that takes part, a few slots beyond the window included)
		const int64_t lay = (two_pass ? c_second : c_first) == 1 ? 12 : (two_pass ? c_second : c_first) == 2 ? 10 : 9; // the pass whose traceback sets the size: 64-column slots own 48 (4/3 of the exact rows), 256-column ones 240 (the second pass of the low-memory mode is narrow: it fits whatever the first needed)
		// (+ a few slots of margin per epoch of 256 penalties; epochs by the guessed penalty, not by the worst case — 10 M penalties for the 5 Mb pair, 20 GB of margin)
		const int64_t ep_guess = (len * 3 / 100 + 1024) / 256 + 2;
		if (use_sys) guess = std::min<int64_t>(g->coop_tb_cap, guess / 8 * lay + ep_guess * 8 * 65536 * (int64_t)NG);
		int64_t want = std::min(use_sys ? worst / 8 * lay + (rows_slot / 256 + 2) * 8 * 65536 * (int64_t)NG : worst, g->tb_budget_mb > 0 ? (g->tb_budget_mb << 20) : guess * g->coop_tb_mult);
		if ((int64_t)g->tb.bytes >= want) want = (int64_t)g->tb.bytes;
		else {
			size_t fr = 0, tot = 0;
			if (hipMemGetInfo(&fr, &tot) != hipSuccess) fr = (size_t)8 << 30;
			want = std::min<int64_t>(want, (int64_t)(fr / 10 * 9) + (int64_t)g->tb.bytes);
		}
		tb_bytes = std::max<int64_t>(4096, want / (int64_t)NG) / 4 * 4; // per pair
		if (ensure(g, g->row_off, NG * (size_t)rows_slot * 8)) return -1;
		if (ensure(g, g->row_lo, NG * (size_t)rows_slot * 4)) return -1;
		if (ensure(g, g->cig_scratch, NG * (size_t)cig_scratch * 4)) return -1;
		if (low_mem) {
			seg_slot = bound1 / opt.step + 2;
			if (ensure(g, g->seg, NG * (size_t)seg_slot * 8)) return -1;
		}
		if (ensure(g, g->tb, NG * (size_t)tb_bytes)) return -1;
	}
	int64_t snap_slot_ints = 0, snap_meta_slot = 0;
	if (two_pass) {
		// shadow H rows; snapshots: (nH + 2 e1 + 2 e2) array-slices x the window at every `step` penalties, windows about as
		// wide as the penalty: ~ NS * s^2 / step ints with s guessed as 3 % of tl+ql (doubled with the arena after an overflow)
		// (systolic layout: a snapshot holds NS array-slices of every owned column of the chunks that take part in its epoch — the window plus
		// 2 x 265 columns and a chunk; the window at snapshot j is about 2 j step wide; s is guessed as 3 % of tl+ql, doubled with the arena after an overflow)
		const int64_t NS = P.nH + 2 * P.e1 + 2 * P.e2, s_guess = len * 3 / 100 + 1024, n_guess = s_guess / opt.step + 2;
		int64_t cols = 0;
		for (int64_t j = 1; j <= n_guess; ++j) cols += std::min<int64_t>(len + 1, 2 * j * opt.step) + 1100;
		snap_meta_slot = (bound1 / opt.step + 256 / opt.step + 4) * 8; // (+ the snapshots of the last epoch's surplus penalties)
		snap_slot_ints = std::min<int64_t>(std::max<int64_t>((int64_t)4 << 20, NS * cols) * g->coop_tb_mult, ((int64_t)48 << 30) / 4) / 4 * 4;
		if (ensure(g, g->sys_sring, NG * (size_t)TC * P.nH * 256 * 4)) return -1;
		if (ensure(g, g->snap, NG * (size_t)snap_slot_ints * 4)) return -1;
		if (ensure(g, g->snap_meta, NG * (size_t)snap_meta_slot * 4)) return -1;
	}
	if (traced) {
		if (ensure(g, g->dbg, (size_t)8 * (bound + 2))) return -1;
		HIP_TRY(g, hipMemsetAsync(g->dbg.p, 0, g->dbg.bytes, g->stream));
	}

	const int64_t sys_rows = std::max(bound, bound1) + 2, sys_log_ints = 2 * (sys_rows + 256 + 8), sys_ep_words = 2 * (sys_rows / 256 + 3);
	const int64_t sys_box_group = TC * 2 * std::max(sys_box_ints(sysP, two_pass), sys_box_ints(sysP2, false)), sys_park_group = TC * (two_pass ? 16 : 8) * 64 * 4;
	if (use_sys) {
		if (ensure(g, g->sys_ring, NG * (size_t)TC * P.nH * 256 * 4)) return -1;
		if (ensure(g, g->sys_good, NG * (size_t)P.nH * TC * 4 * 8)) return -1;
		if (ensure(g, g->sys_box, NG * (size_t)sys_box_group * 4)) return -1;
		if (ensure(g, g->sys_prog, NG * (size_t)TC * 64)) return -1;
		if (ensure(g, g->sys_log, NG * (size_t)sys_log_ints * 4)) return -1;
		if (ensure(g, g->sys_ep, NG * (size_t)sys_ep_words * 8)) return -1;
		if (ensure(g, g->sys_park, NG * (size_t)sys_park_group * 4)) return -1;
	}
	BatchArgs a;
	memset(&a, 0, sizeof(a));
	a.seqs = b->d_seqs, a.t_off = b->d_t_off, a.q_off = b->d_q_off, a.tl = b->d_tl, a.ql = b->d_ql;
	a.n_pairs = b->n;
	a.pen = P;
	a.want_cigar = cigar ? 1 : 0;
	a.step = low_mem ? opt.step : 0;
	a.max_s = opt.max_s, a.max_iter = opt.max_iter;
	a.debug_pair = b->debug_pair;
	a.ring = (int32_t*)g->ring.p, a.ring_slot_ints = (int64_t)P.nH * W, a.W = W;
	a.good = (unsigned long long*)g->good.p, a.GW = GW;
	a.tb = cigar ? (uint8_t*)g->tb.p : nullptr, a.tb_slot_bytes = tb_bytes;
	a.row_off = cigar ? (int64_t*)g->row_off.p : nullptr, a.row_lo = cigar ? (int32_t*)g->row_lo.p : nullptr, a.rows_slot = rows_slot;
	a.cig_scratch = cigar ? (uint32_t*)g->cig_scratch.p : nullptr, a.cig_scratch_slot = cig_scratch;
	a.cig_pool = b->d_cig_pool, a.cig_head = b->d_cig_head, a.cig_pool_words = b->cig_pool_words;
	a.seg = low_mem ? (int32_t*)g->seg.p : nullptr, a.seg_slot = seg_slot;
	a.sring = two_pass ? (int32_t*)g->sys_sring.p : nullptr; // (provenance of the systolic kernel's private H rings: the same shape, set_cols())
	a.snap = two_pass ? (int32_t*)g->snap.p : nullptr, a.snap_slot_ints = snap_slot_ints;
	a.snap_meta = two_pass ? (int32_t*)g->snap_meta.p : nullptr, a.snap_meta_slot = snap_meta_slot;
	a.out_s = b->d_s, a.out_iter = b->d_iter, a.out_ncig = b->d_ncig, a.out_cigoff = b->d_cigoff;
	a.out_status = b->d_status, a.out_cells1 = b->d_cells1, a.out_dbg = b->d_dbg4;
	a.dbg = traced && n_groups == 1 ? (int32_t*)g->dbg.p : nullptr; // the band trace is a single-pair diagnostic
	a.dbg_cap = a.dbg ? (int32_t)(g->dbg.bytes / 8) : 0;
	a.coop_pair = pairs[0];
	a.coop_spin_limit = (uint32_t)g->coop_spin_limit;
	a.coop_groups = n_groups, a.coop_group_size = Gs;
	a.coop_edge = (int32_t*)g->coop_edge.p, a.coop_edge_stride = (int64_t)(gran_bytes / 4);
	a.coop_sedge_off = two_pass ? (int64_t)(gran_bytes / 8) : 0;
	a.coop_flags = (int32_t*)g->coop_misc.p, a.coop_misc_stride = (int64_t)misc_bytes; // per group: flags | +1024 barrier words | +2048 pass state | +4096 flag ring
	a.coop_sync = (unsigned int*)((char*)g->coop_misc.p + 1024);
	a.coop_state = (int32_t*)((char*)g->coop_misc.p + 2048);
	int32_t *d_ids = (int32_t*)((char*)g->coop_misc.p + NG * misc_bytes);
	HIP_TRY(g, hipMemcpyAsync(d_ids, pairs.data(), NG * 4, hipMemcpyHostToDevice, g->stream));
	a.coop_pair_ids = d_ids;

	// the same launch on the systolic kernel: a private H ring per chunk slot, its own good-bit rows, hand-off boxes, edge log
	BatchArgs as = a;
	if (use_sys) {
		as.ring = (int32_t*)g->sys_ring.p, as.good = (unsigned long long*)g->sys_good.p; // (sized for four columns per lane)
		as.rows_slot = sys_rows;
		as.sys_p = sysP;
		as.sys_coop_launch = g->coop_launch;
		as.sys_spread = 1; // consecutive chunks on consecutive workgroups: 763 against 787 ms on the 5 Mb pair, 63.5 against 65.0 on the 150 kb pair
		as.sys_box = (int32_t*)g->sys_box.p, as.sys_box_stride = sys_box_group;
		as.sys_prog = (unsigned long long*)g->sys_prog.p, as.sys_prog_stride = TC * 8;
		as.sys_log = (int32_t*)g->sys_log.p, as.sys_log_stride = sys_log_ints;
		as.sys_ep = cigar ? (int64_t*)g->sys_ep.p : nullptr, as.sys_ep_stride = sys_ep_words;
		as.sys_park = (int32_t*)g->sys_park.p, as.sys_park_stride = sys_park_group;
	}
	auto reset_sys = [&](bool all) -> int { // counters at zero, nothing published
		for (size_t q = 0; q < NG; ++q) {
			char *m = (char*)g->coop_misc.p + q * misc_bytes;
			if (all) { HIP_TRY(g, hipMemsetAsync(m, 0, 4096, g->stream)); }
			else HIP_TRY(g, hipMemsetAsync(m + 1024, 0, 1024, g->stream));
		}
		HIP_TRY(g, hipMemsetAsync(g->sys_prog.p, 0, NG * (size_t)TC * 64, g->stream));
		return 0;
	};
	std::unique_lock<std::shared_mutex> lock(g_dev_gate[g->device % kMaxDevices]);
	HIP_TRY(g, hipDeviceSynchronize()); // kernels of other engines (other host threads) on this device: let them drain first
	if (reset_sys(true)) return -1;
	if (first) HIP_TRY(g, hipEventRecord(g->ev0, g->stream));
	auto set_cols = [&](int c) { as.sys_c = c, as.ring_slot_ints = TC * P.nH * 64 * c, as.GW = (int32_t)(TC * c); };
	set_cols(c_first);
	a.coop_pass = as.coop_pass = two_pass ? 3 : low_mem ? 1 : 0;
	g->stats.lowmem_two_pass = two_pass ? 1 : 0;
	if (launch_sys_pass(as, Gs * n_groups, g->stream)) { g->err = "kernel launch failed (whole-device pass)"; return -1; }
	g->stats.n_launches += 1;
	if (low_mem) {
		if (two_pass ? launch_sys_trace(as, g->stream) : launch_sys_walk(as, g->stream)) { g->err = "kernel launch failed (checkpoints)"; return -1; }
		if (reset_sys(false)) return -1; // barrier counters and progress words of the second pass
		a.coop_pass = as.coop_pass = 2;
		as.sys_p = sysP2;
		set_cols(c_second);
		// the second pass is not traced: the band trace of a low-memory run is that of its second pass, traced below
		if (launch_sys_pass(as, Gs * n_groups, g->stream)) { g->err = "kernel launch failed (second pass)"; return -1; }
		g->stats.n_launches += 2;
	}
	if (launch_sys_finish(as, g->stream)) { g->err = "kernel launch failed (traceback)"; return -1; }
	g->stats.n_launches += 1;
	if (last) {
		HIP_TRY(g, hipEventRecord(g->ev1, g->stream));
		g->ev_pending = true;
	}
	g->stats.grid = Gs * n_groups, g->stats.block = 512, g->stats.kernel_kind = 1;
	b->last_grid = std::max(b->last_grid, Gs * n_groups);
	for (int32_t pair : pairs) b->h_kind[pair] = 1, b->h_flags[pair] = (int8_t)((b->h_flags[pair] & ~(2 | 32)) | (n_groups > 1 ? 2 : 0) | ((c_first == 1 || (low_mem && c_second == 1)) ? 32 : 0));
	HIP_TRY(g, hipStreamSynchronize(g->stream)); // the device stays ours until the kernels are through
	return 0;
}

// one pair with the device to itself
int run_coop_pair(mwf_gpu_t *g, mwf_gpu_batch_t *b, const mwf_opt_t &opt, int32_t pair, bool first, bool last)
{
	const int G = coop_group_size(coop_grid_limit(g), (int64_t)b->h_tl[pair] + b->h_ql[pair], true, sys_owned_cols(g->sys_p, 4));
	return run_coop_group(g, b, opt, std::vector<int32_t>{pair}, G, first, last);
}

// can the whole-device traceback arena still grow? (free memory beyond what it already holds)
bool coop_can_grow(mwf_gpu_t *g)
{
	size_t fr = 0, tot = 0;
	if (hipMemGetInfo(&fr, &tot) != hipSuccess) return false;
	return (int64_t)(fr / 10 * 9) > (int64_t)g->tb.bytes / 2; // room for at least half as much again
}

int finalize(mwf_gpu_t *g, mwf_gpu_batch_t *b);

// ---- batches ------------------------------------------------------------------------------------------------------------

// Carve the batch's one allocation.  [order | t_off q_off tl ql seqs (owned inputs) | results]; the input part is laid
// out exactly as upload_segments() streams it.
struct BlockLayout {
	size_t order = 0, t_off = 0, q_off = 0, tl = 0, ql = 0, seqs = 0, in_end = 0;
	size_t head = 0, status = 0, s = 0, ncig = 0, iter = 0, cigoff = 0, cells1 = 0, score_end = 0, out_end = 0, dbg4 = 0, total = 0;
};

BlockLayout layout_block(size_t n, size_t seq_bytes, bool owned)
{
	const size_t N = std::max<size_t>(n, 1);
	BlockLayout L;
	size_t at = 0;
	L.order = at, at += align_up(N * 4, 16);
	if (owned) {
		L.t_off = at, at += N * 8;
		L.q_off = at, at += N * 8;
		L.tl = at, at += align_up(N * 4, 16);
		L.ql = at, at += align_up(N * 4, 16);
		L.seqs = at, at += seq_bytes + 64; // word-sized probes may read past the last base
	}
	L.in_end = at;
	at = align_up(at, 256);
	L.head = at, at += 64;
	L.status = at, at += align_up(N * 4, 8);
	L.s = at, at += align_up(N * 4, 8);
	L.iter = at, at += N * 8;
	L.score_end = at; // a score-only, high-memory align needs nothing behind this back
	L.ncig = at, at += align_up(N * 4, 8);
	L.cigoff = at, at += N * 8;
	L.cells1 = at, at += N * 8;
	L.out_end = at;
	L.dbg4 = at, at += N * 16;
	L.total = at;
	return L;
}

mwf_gpu_batch_t *batch_common(mwf_gpu_t *g, int32_t n, const int32_t *h_tl, const int32_t *h_ql, size_t seq_bytes, bool owned, BlockLayout &L)
{
	mwf_gpu_batch_t *b = new mwf_gpu_batch_t();
	b->g = g, b->n = n, b->owns_inputs = owned;
	b->h_tl.assign(h_tl, h_tl + n);
	b->h_ql.assign(h_ql, h_ql + n);
	int64_t words = 0;
	for (int32_t i = 0; i < n; ++i) {
		words += (int64_t)h_tl[i] + h_ql[i] + 1;
		b->max_tl = std::max<int64_t>(b->max_tl, h_tl[i]);
		b->max_seq_lds = std::max<int64_t>(b->max_seq_lds, (((int64_t)h_tl[i] + 3) & ~3LL) + 8 + (((int64_t)h_ql[i] + 3) & ~3LL) + 16);
	}
	b->cig_pool_words = std::max<int64_t>(words, 1);
	// Thousands of pairs: one atomic on the pool's head per pair is ~12.7 ns on a single address (0.5 ms for 40 000 reads) — workgroups take the pool
	// in blocks of kCigBlock words instead (dev::finish_pair).  A block is abandoned with less than a quarter of it unused and every workgroup leaves
	// one partly used: 4/3 of the worst case plus a block per workgroup (at most kCigBlockGrid of them, run_batch_kernel) always holds.
	// (short pairs only: that is where thousands of CIGARs per millisecond are written — and where a third more pool is a few megabytes)
	if (n >= kCigBlockPairs && words / n <= 2048) b->cig_block = kCigBlock, b->cig_pool_words = b->cig_pool_words / 3 * 4 + 4 + (int64_t)(kCigBlockGrid + 1) * kCigBlock;
	L = layout_block((size_t)n, seq_bytes, owned);
	if (take_block(g, g->spare_block, b->block, L.total)) {
		delete b;
		return nullptr;
	}
	char *base = (char*)b->block.p;
	b->d_order = (int32_t*)(base + L.order);
	b->d_cig_head = (unsigned long long*)(base + L.head);
	b->d_status = (int32_t*)(base + L.status), b->d_s = (int32_t*)(base + L.s), b->d_ncig = (int32_t*)(base + L.ncig);
	b->d_iter = (int64_t*)(base + L.iter), b->d_cigoff = (int64_t*)(base + L.cigoff), b->d_cells1 = (int64_t*)(base + L.cells1);
	b->d_dbg4 = (int32_t*)(base + L.dbg4);
	b->out_off = L.head, b->out_bytes = L.out_end - L.head, b->out_bytes_score = L.score_end - L.head;
	// longest pairs first, so the persistent workgroups finish together
	b->h_order.resize((size_t)n);
	std::iota(b->h_order.begin(), b->h_order.end(), 0);
	{
		// (round 5: this sort was 2 of the 3 ms a 40 000-read batch's upload took — stable_sort through an indirect comparison.  Batches of equal
		// or already descending lengths need none; the others sort 64-bit keys (length descending, index ascending = the stable order) directly.)
		bool sorted = true;
		for (int32_t i = 1; i < n && sorted; ++i) sorted = (int64_t)h_tl[i - 1] + h_ql[i - 1] >= (int64_t)h_tl[i] + h_ql[i];
		int64_t max_sum = 0;
		for (int32_t i = 0; i < n; ++i) max_sum = std::max<int64_t>(max_sum, (int64_t)h_tl[i] + h_ql[i]);
		if (!sorted && max_sum < 65536 && n >= 4096) { // reads: one counting pass (stable, longest first)
			std::vector<int32_t> cnt((size_t)max_sum + 2, 0);
			for (int32_t i = 0; i < n; ++i) ++cnt[(size_t)(max_sum - ((int64_t)h_tl[i] + h_ql[i])) + 1];
			for (size_t k = 1; k < cnt.size(); ++k) cnt[k] += cnt[k - 1];
			for (int32_t i = 0; i < n; ++i) b->h_order[(size_t)cnt[(size_t)(max_sum - ((int64_t)h_tl[i] + h_ql[i]))]++] = i;
		} else if (!sorted) {
			std::vector<uint64_t> key((size_t)n);
			for (int32_t i = 0; i < n; ++i) key[i] = ((uint64_t)(0xffffffffu - (uint32_t)((int64_t)h_tl[i] + h_ql[i])) << 32) | (uint32_t)i; // (tl + ql < 2^31)
			std::sort(key.begin(), key.end());
			for (int32_t i = 0; i < n; ++i) b->h_order[i] = (int32_t)(uint32_t)key[i];
		}
	}
	b->h_len_order = b->h_order;
	b->h_class.assign((size_t)n, 0), b->h_kind.assign((size_t)n, 0), b->h_flags.assign((size_t)n, 0);
	return b;
}

// Every byte one of A, C, G, T (what the packed band kernel's 2-bit sequence copy can hold)?  Eight bytes per step: the code
// the kernel would store, (byte >> 1) & 3, stands for exactly one letter; the byte must be that letter.
bool plain_acgt(const uint8_t *p, size_t n)
{
	uint64_t bad = 0;
	size_t i = 0;
	for (; i + 8 <= n; i += 8) {
		uint64_t x;
		memcpy(&x, p + i, 8);
		const uint64_t code = (x >> 1) & 0x0303030303030303ull, lo1 = code & 0x0101010101010101ull, hi1 = (code >> 1) & 0x0101010101010101ull;
		const uint64_t expect = 0x4141414141414141ull + (lo1 & ~hi1) * 0x02u + (hi1 & ~lo1) * 0x13u + (hi1 & lo1) * 0x06u; // A 0x41, C 0x43, T 0x54, G 0x47
		bad |= x ^ expect;
	}
	for (; i < n; ++i) {
		const uint32_t x = p[i], code = (x >> 1) & 3u;
		bad |= x ^ ((0x47544341u >> (8 * code)) & 0xffu);
	}
	return bad == 0;
}

// How diverged are the pairs of a batch?  The size classes below are drawn from the pair LENGTHS for a prior of 5 % (window ~ 0.28 (tl+ql)); at 15 % and
// 30 % every pair of a batch outgrew its class and was run twice (profiles/r04/chooser_regression.txt).  The reference has no classes to get wrong
// (one loop serves any divergence, miniwfa.c:396-426); here a k-mer sketch of a few pairs says where the batch stands before anything is launched:
// the share f of the query's 8-mers (prefix of up to 1500 bases) that occur in the target's prefix is about (1 - d)^8 plus chance hits.
// A few microseconds per sampled pair, at most 16 pairs.
float estimate_divergence(int32_t n, const int32_t *tl, const int32_t *ql, const std::function<const uint8_t*(int32_t, bool)> &seq)
{
	constexpr int K = 8;
	constexpr uint32_t MASK = (1u << (2 * K)) - 1;
	std::vector<uint64_t> bits((size_t)1 << (2 * K - 6));
	double sum = 0;
	int used = 0;
	const int want = 16;
	for (int k = 0; k < want && k < n; ++k) {
		const int32_t i = (int32_t)((int64_t)k * n / std::min(want, n));
		const int32_t lt = std::min(tl[i], 1500), lq = std::min(ql[i], 1500);
		if (lt < 4 * K || lq < 4 * K) continue;
		std::fill(bits.begin(), bits.end(), 0);
		const uint8_t *t = seq(i, true), *q = seq(i, false);
		uint32_t h = 0;
		for (int32_t j = 0; j < lt; ++j) {
			h = ((h << 2) | ((t[j] >> 1) & 3u)) & MASK;
			if (j >= K - 1) bits[h >> 6] |= 1ull << (h & 63);
		}
		int32_t hit = 0, tot = 0;
		h = 0;
		for (int32_t j = 0; j < lq; ++j) {
			h = ((h << 2) | ((q[j] >> 1) & 3u)) & MASK;
			if (j >= K - 1) ++tot, hit += (int32_t)((bits[h >> 6] >> (h & 63)) & 1u);
		}
		const double fp = 1.0 - std::exp(-(double)(lt - K + 1) / (double)(MASK + 1)); // chance hits
		double f = ((double)hit / tot - fp) / (1.0 - fp);
		f = std::min(1.0, std::max(f, 1e-3));
		sum += 1.0 - std::pow(f, 1.0 / K);
		++used;
	}
	return used ? (float)(sum / used) : 0.f;
}

// A batch from host memory: pair i is (ts[i], tl[i]) / (qs[i], ql[i]) when `ts` is given, else it lies in `packed` at
// t_off[i] / q_off[i].  Everything goes up in one stream of copies through the pinned buffer.
mwf_gpu_batch_t *batch_from_host(mwf_gpu_t *g, int32_t n, const int32_t *tl, const char *const *ts, const int32_t *ql, const char *const *qs,
                                 const char *packed, int64_t packed_bytes, const int64_t *p_t_off, const int64_t *p_q_off)
{
	(void)hipSetDevice(g->device);
	static const bool timing = getenv("MWF_UPLOAD_TIMING") != nullptr; // (diagnostics: where a batch's upload goes)
	const auto tm0 = std::chrono::steady_clock::now();
	auto lap = [&](const char *what) {
		if (timing) fprintf(stderr, "[libmwf_hip] upload: %s at %.3f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tm0).count());
	};
	std::vector<int64_t> t_off, q_off;
	int64_t seq_bytes = packed_bytes;
	if (ts) {
		t_off.resize((size_t)n), q_off.resize((size_t)n);
		seq_bytes = 0;
		for (int32_t i = 0; i < n; ++i) {
			t_off[i] = seq_bytes, seq_bytes += tl[i];
			q_off[i] = seq_bytes, seq_bytes += ql[i];
		}
		p_t_off = t_off.data(), p_q_off = q_off.data();
	}
	BlockLayout L;
	mwf_gpu_batch_t *b = batch_common(g, n, tl, ql, (size_t)seq_bytes, true, L);
	if (!b) return nullptr;
	lap("batch_common (lengths, order, device block)");
	b->seq_bytes = seq_bytes;
	// the host touches every byte anyway: note which pairs the 2-bit sequence copy cannot hold, so that they never take the
	// device round trip through ST_ALPHABET
	b->h_acgt.resize((size_t)n);
	auto classify = [&](int32_t i0, int32_t i1) {
		for (int32_t i = i0; i < i1; ++i) {
			const uint8_t *pt = ts ? (const uint8_t*)ts[i] : (const uint8_t*)packed + p_t_off[i];
			const uint8_t *pq = ts ? (const uint8_t*)qs[i] : (const uint8_t*)packed + p_q_off[i];
			b->h_acgt[i] = plain_acgt(pt, (size_t)tl[i]) && plain_acgt(pq, (size_t)ql[i]) ? 1 : 0;
		}
	};
	b->div_est = estimate_divergence(n, tl, ql, [&](int32_t i, bool target) -> const uint8_t* {
		return target ? (ts ? (const uint8_t*)ts[i] : (const uint8_t*)packed + p_t_off[i]) : (ts ? (const uint8_t*)qs[i] : (const uint8_t*)packed + p_q_off[i]);
	});
	std::vector<std::thread> th; // (joined behind the packing below: the classification is first needed by an align)
	if (seq_bytes < ((int64_t)2 << 20) || n < 16) classify(0, n);
	else { // megabytes of sequence: a few host threads, equal shares of the bytes (one thread does ~8 GB/s), WHILE this thread packs the batch into the pinned buffer
		const int n_th = (int)std::min<int64_t>(4, std::min<int64_t>(std::max(1u, std::thread::hardware_concurrency()), seq_bytes >> 20));
		int32_t i0 = 0;
		int64_t acc = 0, done_bytes = 0;
		for (int k = 0; k < n_th; ++k) {
			const int64_t want = (seq_bytes - done_bytes) / (n_th - k);
			int32_t i1 = i0;
			for (acc = 0; i1 < n && (acc < want || k + 1 == n_th); ++i1) acc += (int64_t)tl[i1] + ql[i1];
			done_bytes += acc;
			th.emplace_back(classify, i0, k + 1 == n_th ? n : i1);
			i0 = i1;
		}
	}
	lap("alphabet classification started");
	char *base = (char*)b->block.p;
	b->d_t_off = (const int64_t*)(base + L.t_off), b->d_q_off = (const int64_t*)(base + L.q_off);
	b->d_tl = (const int32_t*)(base + L.tl), b->d_ql = (const int32_t*)(base + L.ql);
	b->d_seqs = (const uint8_t*)(base + L.seqs);
	const size_t N = (size_t)n;
	std::vector<Seg> segs;
	segs.reserve(ts ? 2 * N + 12 : 12);
	auto pad_to = [&](size_t have, size_t want) { if (want > have) segs.push_back(Seg{nullptr, want - have}); };
	segs.push_back(Seg{b->h_order.data(), N * 4}), pad_to(L.order + N * 4, L.t_off);
	segs.push_back(Seg{p_t_off, N * 8});
	segs.push_back(Seg{p_q_off, N * 8});
	segs.push_back(Seg{tl, N * 4}), pad_to(L.tl + N * 4, L.ql);
	segs.push_back(Seg{ql, N * 4}), pad_to(L.ql + N * 4, L.seqs);
	if (ts) {
		for (int32_t i = 0; i < n; ++i) {
			if (tl[i]) segs.push_back(Seg{ts[i], (size_t)tl[i]});
			if (ql[i]) segs.push_back(Seg{qs[i], (size_t)ql[i]});
		}
	} else if (packed_bytes > 0) segs.push_back(Seg{packed, (size_t)packed_bytes});
	segs.push_back(Seg{nullptr, 64});
	// a small batch (the single pair of a drop-in call): its result arrays come up initialised with the same copy — every pair "not run",
	// CIGAR counter at zero — so that its first align launches no reset kernel
	static const int32_t kNotRun[64] = {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1,
	                                    -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1};
	static const int32_t kNotFinal[64] = {-2, -2, -2, -2, -2, -2, -2, -2, -2, -2, -2, -2, -2, -2, -2, -2, -2, -2, -2, -2, -2, -2, -2, -2, -2, -2, -2, -2, -2, -2, -2, -2,
	                                      -2, -2, -2, -2, -2, -2, -2, -2, -2, -2, -2, -2, -2, -2, -2, -2, -2, -2, -2, -2, -2, -2, -2, -2, -2, -2, -2, -2, -2, -2, -2, -2};
	if (n >= 1 && n <= 64) {
		pad_to(L.in_end, L.head);
		segs.push_back(Seg{nullptr, 64});                                   // head: the CIGAR pool's counter
		segs.push_back(Seg{kNotRun, N * 4}), pad_to(L.status + N * 4, L.s);
		segs.push_back(Seg{kNotFinal, N * 4});
		b->results_preinit = true;
	}
	const int up_rc = upload_segments(g, base, segs);
	for (std::thread &t : th) t.join();
	if (up_rc) {
		mwf_gpu_batch_free(b);
		return nullptr;
	}
	lap("packed into the pinned buffer, copies enqueued, classification joined");
	return b;
}

} // namespace

extern "C" {

/* ------------------------------------------------------------------ engine */

int mwf_gpu_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
	return n;
}

mwf_gpu_t *mwf_gpu_create(int device, void *stream)
{
	int n = mwf_gpu_device_count();
	if (device < 0 || device >= n) return nullptr;
	if (hipSetDevice(device) != hipSuccess) return nullptr;
	mwf_gpu_t *g = new mwf_gpu_t();
	g->device = device;
	hipDeviceProp_t prop;
	if (hipGetDeviceProperties(&prop, device) != hipSuccess) { delete g; return nullptr; }
	g->n_cu = prop.multiProcessorCount;
	g->total_mem = prop.totalGlobalMem;
	if (stream) g->stream = (hipStream_t)stream;
	else {
		if (hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking) != hipSuccess) { delete g; return nullptr; }
		g->own_stream = true;
	}
	(void)hipEventCreate(&g->ev0);
	(void)hipEventCreate(&g->ev1);
	if (ensure(g, g->queue, kQueueInts * 4) || hipMemsetAsync(g->queue.p, 0, kQueueInts * 4, g->stream) != hipSuccess) {
		mwf_gpu_destroy(g);
		return nullptr;
	}
	return g;
}

static void trim(mwf_gpu_t *g)
{
	(void)hipStreamSynchronize(g->stream);
	for (DevBuf *b : {&g->ring, &g->sring, &g->good, &g->tb, &g->row_off, &g->row_lo, &g->cig_scratch, &g->snap, &g->snap_meta, &g->seg, &g->dbg,
	                  &g->coop_edge, &g->coop_misc, &g->sys_box, &g->sys_prog, &g->sys_log, &g->sys_ep, &g->sys_park, &g->sys_ring, &g->sys_sring, &g->sys_good, &g->retry_ids,
	                  &g->spare_block, &g->spare_cig})
		release(g, *b);
	g->dev_bytes_peak = g->dev_bytes;
}

void mwf_gpu_destroy(mwf_gpu_t *g)
{
	if (!g) return;
	(void)hipSetDevice(g->device);
	trim(g);
	release(g, g->queue);
	if (g->pin) (void)hipHostFree(g->pin);
	if (g->res_pin) (void)hipHostFree(g->res_pin);
	for (hipEvent_t e : {g->ev0, g->ev1, g->pin_ev[0], g->pin_ev[1]})
		if (e) (void)hipEventDestroy(e);
	if (g->own_stream) (void)hipStreamDestroy(g->stream);
	delete g;
}

const char *mwf_gpu_last_error(const mwf_gpu_t *g) { return g ? g->err.c_str() : "no engine (no gfx950 device could be opened)"; }

int mwf_gpu_set(mwf_gpu_t *g, const char *name, int64_t value)
{
	if (!g || !name) return -1;
	if (!strcmp(name, "block")) {
		if (value != 0 && value != 64 && value != 128 && value != 256 && value != 512 && value != 768 && value != 1024) return -1;
		g->block = (int)value;
	} else if (!strcmp(name, "slots_per_cu")) g->slots_per_cu = (int)value;
	else if (!strcmp(name, "coop_min_len")) g->coop_min_len = value;
	else if (!strcmp(name, "tb_budget_mb")) g->tb_budget_mb = value;
	else if (!strcmp(name, "force_kind")) g->force_kind = (int)value;
	else if (!strcmp(name, "seq2bit")) g->seq2bit = (int)value;
	else if (!strcmp(name, "ring16")) g->ring16 = (int)value;
	else if (!strcmp(name, "ring16_block") && (value == 0 || value == 512 || value == 768)) g->ring16_block = (int)value;
	else if (!strcmp(name, "band_pack")) g->band_pack = (int)value;
	else if (!strcmp(name, "band_span") && value >= 0 && value <= 2) g->band_span = (int)value;
	else if (!strcmp(name, "wide_slots") && (value == 0 || value == 3 || value == 4)) g->wide_slots = (int)value;
	else if (!strcmp(name, "lane_chunks") && value >= 0 && value <= 4) g->lane_chunks = (int)value;
	else if (!strcmp(name, "host_results")) g->res_pin_on = value != 0;
	else if (!strcmp(name, "lane_max_len")) g->lane_max_len = (int)std::max<int64_t>(0, std::min<int64_t>(value, 8000));
	else if (!strcmp(name, "mid_max_pairs")) g->mid_max_pairs = (int)std::max<int64_t>(-1, std::min<int64_t>(value, 1 << 20));
	else if (!strcmp(name, "mid_block") && (value == 0 || value == 256 || value == 512 || value == 1024)) g->mid_block = (int)value;
	else if (!strcmp(name, "lds_e2")) g->lds_e2 = value != 0;
	else if (!strcmp(name, "scalar_generic")) g->scalar_generic = value != 0;
	else if (!strcmp(name, "coop_spin_limit")) g->coop_spin_limit = std::max<int64_t>(0, std::min<int64_t>(value, 0x7fffffff));
	else if (!strcmp(name, "coop_tb_cap_mb")) g->coop_tb_cap = std::max<int64_t>(1, value) << 20;
	else if (!strcmp(name, "lowmem_budget_mb")) g->lowmem_budget_mb = std::max<int64_t>(0, value);
	else if (!strcmp(name, "coop_grid")) g->coop_grid_cap = (int)std::max<int64_t>(0, value);
	else if (!strcmp(name, "coop_launch")) g->coop_launch = value != 0;
	else if (!strcmp(name, "sys_p") && sys_p_supported((int)value)) g->sys_p = (int)value; // (8; 4 and 16 only in builds with -DMWF_SYS_ALL_P)
	else if (!strcmp(name, "sys_p2") && (value == 0 || sys_p_supported((int)value))) g->sys_p2 = (int)value;
	else if (!strcmp(name, "sys_c") && (value == 0 || sys_c_supported((int)value))) g->sys_c = (int)value; // (2: builds with -DMWF_SYS_C2 only — the host's box / traceback layout must be the launched kernel's)
	else if (!strcmp(name, "div_aware")) g->div_aware = value != 0;
	else if (!strcmp(name, "trim")) { (void)hipSetDevice(g->device); trim(g); }
	else return -1;
	++g->tun_gen; // (whatever the tunable: no hand-kept list of "the ones that classify" to forget an entry of)
	return 0;
}

void mwf_gpu_get_stats(const mwf_gpu_t *g, mwf_gpu_stats_t *st)
{
	if (!g || !st) return;
	mwf_gpu_t *m = const_cast<mwf_gpu_t*>(g);
	if (m->ev_pending) {
		(void)hipSetDevice(m->device);
		float ms = 0;
		if (hipEventSynchronize(m->ev1) == hipSuccess && hipEventElapsedTime(&ms, m->ev0, m->ev1) == hipSuccess) m->stats.kernel_ms = ms;
		m->ev_pending = false;
	}
	m->stats.dev_bytes = m->dev_bytes, m->stats.dev_bytes_peak = m->dev_bytes_peak;
	*st = m->stats;
}

/* ------------------------------------------------------------------ batches */

mwf_gpu_batch_t *mwf_gpu_batch_upload(mwf_gpu_t *g, int32_t n, const char *seqs, int64_t seq_bytes,
                                      const int64_t *t_off, const int32_t *tl, const int64_t *q_off, const int32_t *ql)
{
	if (!g || n < 0 || seq_bytes < 0) return nullptr;
	return batch_from_host(g, n, tl, nullptr, ql, nullptr, seqs, seq_bytes, t_off, q_off);
}

mwf_gpu_batch_t *mwf_gpu_batch_wrap(mwf_gpu_t *g, int32_t n, const void *d_seqs, int64_t seq_bytes,
                                    const int64_t *d_t_off, const int32_t *d_tl, const int64_t *d_q_off, const int32_t *d_ql,
                                    const int32_t *h_tl, const int32_t *h_ql)
{
	if (!g || n < 0) return nullptr;
	(void)hipSetDevice(g->device);
	BlockLayout L;
	mwf_gpu_batch_t *b = batch_common(g, n, h_tl, h_ql, 0, false, L);
	if (!b) return nullptr;
	b->d_seqs = (const uint8_t*)d_seqs, b->seq_bytes = seq_bytes;
	b->d_t_off = d_t_off, b->d_q_off = d_q_off, b->d_tl = d_tl, b->d_ql = d_ql;
	if (upload_segments(g, (char*)b->block.p, std::vector<Seg>{Seg{b->h_order.data(), (size_t)n * 4}})) {
		mwf_gpu_batch_free(b);
		return nullptr;
	}
	return b;
}

void mwf_gpu_batch_free(mwf_gpu_batch_t *b)
{
	if (!b) return;
	mwf_gpu_t *g = b->g;
	(void)hipSetDevice(g->device);
	if (b->busy) (void)hipStreamSynchronize(g->stream); // kernels of an align nobody waited for may still use the block
	if (g->res_pin_owner == b) g->res_pin_owner = nullptr;
	give_block(g, g->spare_block, b->block);
	give_block(g, g->spare_cig, b->cig);
	delete b;
}

int mwf_gpu_batch_align(mwf_gpu_t *g, mwf_gpu_batch_t *b, const mwf_opt_t *opt)
{
	if (!g || !b || !opt) return -1;
	if (const char *why = validate(*opt)) { g->err = why; return -2; }
	(void)hipSetDevice(g->device);
	b->opt = *opt;
	b->aligned = false, b->finalized = false, b->h_cig_valid = false;
	b->last_grid = 0, b->n_retries = 0;
	g->stats = mwf_gpu_stats_t{};
	if (b->n == 0) { b->aligned = b->finalized = true; return 0; }
	const bool cigar = (opt->flag & MWF_F_CIGAR) != 0;
	{ // Where the result arrays lie.  A small score-only batch (the single pair of a drop-in call) gets them in a page of pinned host
	  // memory that the kernels write directly: results() then waits for the stream and reads them — no copy to enqueue and wait
	  // for (a 200 bp call: ~8 us).  The CIGAR counter is device-side atomics: CIGAR-mode batches keep everything in the block.
		const BlockLayout L = layout_block((size_t)b->n, 0, false); // (only differences between result offsets are used)
		bool pin = g->res_pin_on && !cigar && b->n <= 64 && L.out_end - L.head <= 4096 && (g->res_pin_owner == nullptr || g->res_pin_owner == b);
		if (pin && !g->res_pin && hipHostMalloc(&g->res_pin, 4096, hipHostMallocDefault) != hipSuccess) (void)hipGetLastError(), g->res_pin = nullptr, pin = false;
		if (pin != b->out_in_pin || (pin && g->res_pin_owner != b)) {
			if (b->busy) HIP_TRY(g, hipStreamSynchronize(g->stream)); // an align nobody waited for still writes the old arrays
			char *base = pin ? (char*)g->res_pin - L.head : (char*)b->block.p + (b->out_off - L.head);
			b->d_status = (int32_t*)(base + L.status), b->d_s = (int32_t*)(base + L.s), b->d_ncig = (int32_t*)(base + L.ncig);
			b->d_iter = (int64_t*)(base + L.iter), b->d_cigoff = (int64_t*)(base + L.cigoff), b->d_cells1 = (int64_t*)(base + L.cells1);
			if (g->res_pin_owner == b && !pin) g->res_pin_owner = nullptr;
			if (pin) g->res_pin_owner = b;
			b->out_in_pin = pin;
		}
	}
	if (cigar && !b->d_cig_pool) {
		if (take_block(g, g->spare_cig, b->cig, (size_t)b->cig_pool_words * 4)) return -1;
		b->d_cig_pool = (uint32_t*)b->cig.p;
	}
	// the plan of the last align applies when nothing it was derived from changed: the options that classify pairs and the tunables
	mwf_gpu_batch_t::PlanCache &PC = b->plan;
	{
		const int32_t ok[8] = {opt->flag & MWF_F_CIGAR, opt->x, opt->o1, opt->e1, opt->o2, opt->e2, opt->step, opt->max_s};
		// the tunables as a generation count (every mwf_gpu_set() bumps it) + the engine the plan was made on + the traced pair
		const int64_t tk[14] = {g->tun_gen, (int64_t)(intptr_t)g, b->debug_pair, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
		if (PC.valid && (memcmp(ok, PC.opt_key, sizeof(ok)) || memcmp(tk, PC.tun_key, sizeof(tk)))) PC.valid = false;
		if (!PC.valid) memcpy(PC.opt_key, ok, sizeof(ok)), memcpy(PC.tun_key, tk, sizeof(tk)), PC.has_groups = false, PC.wide_state = 0;
	}
	int64_t max_len = 0, max_bound = 0;
	if (PC.valid) max_len = PC.max_len, max_bound = PC.max_bound;
	else {
		for (int32_t i = 0; i < b->n; ++i) {
			max_len = std::max<int64_t>(max_len, (int64_t)b->h_tl[i] + b->h_ql[i]);
			max_bound = std::max(max_bound, penalty_bound(*opt, b->h_tl[i], b->h_ql[i], true));
		}
		PC.max_len = max_len, PC.max_bound = max_bound, PC.valid = true;
	}
	if (max_len + 4 >= ((int64_t)1 << 31)) { g->err = "tl+ql must be below 2^31-4"; return -2; }
	const int slots = 1 << 30; // as many as the chosen kernel can keep resident (run_batch_kernel bounds it)
	if (b->debug_pair >= 0 && ensure(g, g->dbg, (size_t)8 * (max_bound + 2))) return -1;
	const bool was_busy = b->busy;
	b->busy = true;
	PC.wide_measured = false;
	g->queue_next = 0, g->lane_set_next = 0;
	// Every pair "not run", CIGAR pool and work counters at zero: one small kernel — unless the result arrays can be written from here
	// (the pinned result page of a small score-only batch: the single pair of a drop-in call) or came up initialised with the batch
	// (a small batch's first align); the kernels of such a call then run without a work counter where they can (run_batch_kernel).
	bool preset = false;
	if (b->out_in_pin && !was_busy) {
		for (int32_t i = 0; i < b->n; ++i) b->d_status[i] = -1, b->d_s[i] = -2;
		preset = true;
	} else if (b->results_preinit) preset = true;
	b->results_preinit = false;
	g->queue_clean = !preset;
	b->cig_block_left = kCigBlockGrid;
	if (!preset && launch_reset(b->d_status, b->d_s, b->n, b->d_cig_head, (int32_t*)g->queue.p, kQueueInts, g->stream)) { g->err = "kernel launch failed (reset)"; return -1; }
	std::fill(b->h_flags.begin(), b->h_flags.end(), 0);
	// a few long pairs: each one gets the whole device in turn
	const Penalty P0 = make_penalty(*opt);
	// (round 3: from 20 000 bases of target + query on, a batch of up to sixteen pairs is faster on the whole-device kernel than one
	// workgroup per pair on any other — 1 x 12 kb 4.1 against 6.4 ms, 16 x 25 kb 11.5 / 20.5, 1 x 32 kb 9.6 / 26.1, a single 10 kb pair
	// 3.4 against 5.8 ms on the packed band kernel; profiles/r03/few_long_pairs.txt)
	// (with traceback already from 15 000 on: 1 x 10 kb 5.1 against 7.2 ms on the packed band kernel, 16 x 8 kb 5.2 / 6.2)
	const int64_t coop_len = g->coop_min_len > 0 ? g->coop_min_len : ((opt->flag & MWF_F_CIGAR) ? 15000 : 20000);
	// The whole-device kernel takes time ~ (tl+ql) per round of pairs side by side (each on its own group of workgroups); the generic kernel
	// runs up to 256 (512) pairs at once in time ~ (tl+ql)^2.  Measured at 3 % divergence (profiles/few_long_pairs.py, round 3): 50 kb pairs
	// 22 ms per round of 16 against 60 ms for any number of them on the generic kernel, 100 kb pairs 44 ms per round of 8 against 245 ms,
	// 150 kb pairs 68 ms against 545 ms — the whole-device kernel wins while the batch needs fewer than about 2.7 (tl+ql)/100 000 rounds.
	bool coop = g->force_kind == 1;
	int n_cu_coop = 0;
	if (coop || (g->force_kind < 0 && max_len >= coop_len && coop_supported(P0))) {
		n_cu_coop = coop_grid_limit(g);
		const int coop_side_by_side = n_cu_coop > 0 ? std::max(1, n_cu_coop / coop_group_size(n_cu_coop, max_len, false, sys_owned_cols(g->sys_p, 4))) : 1;
		int64_t coop_max_pairs = std::max<int64_t>(1, std::min<int64_t>(256, std::max<int64_t>(1, max_len * 27 / 1000000) * coop_side_by_side));
		if (max_len < 65536) coop_max_pairs = std::min<int64_t>(coop_max_pairs, 16); // (measured up to sixteen)
		coop = coop || b->n <= coop_max_pairs;
	}
	if (coop) {
		if (!coop_supported(P0)) { g->err = "whole-device kernel does not support these penalties"; return -2; }
		if (n_cu_coop < 1) { g->err = "whole-device kernel cannot be made resident"; return -1; }
		std::vector<int32_t> idx(b->h_order.begin(), b->h_order.end());
		std::stable_sort(idx.begin(), idx.end(), [&](int32_t x, int32_t y) { return (int64_t)b->h_tl[x] + b->h_ql[x] > (int64_t)b->h_tl[y] + b->h_ql[y]; }); // longest first
		for (size_t at = 0; at < idx.size();) {
			const int64_t len0 = (int64_t)b->h_tl[idx[at]] + b->h_ql[idx[at]];
			const int Gs = coop_group_size(n_cu_coop, len0, false, sys_owned_cols(g->sys_p, 4));
			const size_t n_side = std::min<size_t>(idx.size() - at, (size_t)std::max(1, n_cu_coop / Gs));
			const bool last = at + n_side == idx.size();
			if (n_side <= 1 || b->debug_pair >= 0) { // alone (also: band traces are single-pair diagnostics)
				if (run_coop_pair(g, b, *opt, idx[at], at == 0, at + 1 == idx.size())) return -1;
				at += 1;
				continue;
			}
			if (run_coop_group(g, b, *opt, std::vector<int32_t>(idx.begin() + at, idx.begin() + at + n_side), Gs, at == 0, last)) return -1;
			at += n_side;
		}
		b->aligned = true;
		return 0;
	}
	// Size classes.  One long pair must not push a thousand short ones onto the slow kernel (mwf_wfa_chain's gap fills are
	// exactly such a mix): pairs are grouped by what their window can grow to, and every group runs on the kernel that suits
	// it — generic (largest workspace) first, so that later groups never have to grow a buffer.  Kernel and block size forced
	// by the caller (tests, tuning) keep the whole batch in one group.
	// Low-memory mode (opt.step > 0): a pair whose penalty cannot reach `step` never takes a snapshot (the first one is due at
	// penalty step-1, miniwfa.c:585), so its low-memory result IS its high-memory result, n_iter included — such pairs (the
	// gap fills of mwf_wfa_auto's chain fallback, which inherit step = 5000) run in the classes as high-memory pairs; only
	// genuinely long pairs go through the two-pass kernel (group 5).
	const bool low_mem = cigar && opt->step > 0;
	const bool classes = g->force_kind < 0 && g->block == 0 && band2_supported(P0) && g->band_pack != 0;
	// groups 0-4: the size classes, 5: two-pass low-memory pairs, 6-9: classes 1-4 again for the pairs the host knows not to be
	// plain A/C/G/T (byte-wise sequence copy from the start)
	// 10: short pairs on the one-diagonal-per-lane kernel (mwf_lane.hip); what outgrows its 64 columns moves to the band classes
	// 11: mid-size pairs of a small batch on the one-workgroup-per-pair, rings-in-LDS kernel (mwf_mid.hip); what outgrows its span moves to the band classes
	typedef mwf_gpu_batch_t::PlanCache::GI GroupInfo;
	GroupInfo gi[15];
	bool mid_bytes = false;
	// (12: the pairs of class 10 the host knows not to be plain A/C/G/T — reads with an N —: the lane kernel on byte-wise copies)
	// 13: pairs too long (or with windows too wide) for the 512-thread packed geometry on its 1024-thread span geometry: targets of up to ~60 kb on biased
	// 16-bit offsets, windows of up to ~16 000 columns; what outgrows it moves to the generic kernel
	// 14: pairs of up to ~16 kb per sequence whose worst-case penalty rules out plain 16-bit offsets: the 512-thread geometry with four chunk slots, which computes on
	// biased offsets with range checks like the span geometry but keeps two pairs per CU (windows of up to 7872 columns; what outgrows them moves to the span geometry)
	static const int run_order[15] = {5, 0, 13, 14, 1, 6, 2, 7, 3, 8, 4, 9, 11, 10, 12}; // largest workspace first
	if (PC.has_groups) { // same lengths, same options, same tunables as last time: classes, order (already on the device) and maxima as they were
		b->h_class = PC.cls0, b->h_flags = PC.flags0;
		for (int c = 0; c < 15; ++c) gi[c] = PC.gi[c];
		mid_bytes = PC.mid_bytes;
	} else {
		const bool lane_ok = g->lane_max_len > 0 && lane_supported(P0);
		const int mid_cap = g->mid_max_pairs < 0 ? g->n_cu : g->mid_max_pairs;
		const bool mid_ok = g->force_kind < 0 && g->block == 0 && mid_cap > 0 && b->n <= mid_cap && mid_supported(P0);
		const bool know_acgt = !b->h_acgt.empty() && g->seq2bit != 0;
		const bool pack_pen = g->band_pack != 0 && band2_supported(P0);
		std::vector<int8_t> cls((size_t)b->n); // group of every pair
		int32_t count[15] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
		const bool span_pen = pack_pen && g->band_span != 0 && g->seq2bit != 0;
		// The classes' length limits stand for "the window stays inside the span at ~5 % divergence".  Where the batch's own divergence is known
		// (estimate_divergence: batches built from host memory) the lengths are weighed by it: three times as diverged = as if three times as long.
		// Only upwards of the prior, and a little downwards: a class too narrow costs a second run, one too wide a few per cent.
		const double div_r = b->div_est > 0 && g->div_aware ? std::min(8.0, std::max(0.7, (double)b->div_est / 0.05)) : 1.0;
		for (int32_t i = 0; i < b->n; ++i) {
			const int64_t tl = b->h_tl[i], ql = b->h_ql[i], len = tl + ql;
			const int64_t lenw = (int64_t)((double)len * div_r); // the pair's length as the classes' limits should see it
			// the window the pair is expected to reach (0.27 (tl+ql) at 5 %), plus 15 %, where the divergence is known: the long classes' length limits
			// were drawn for ~3 % (configs[4]) and sent 20 kb pairs at 15 % and 50 kb pairs at 5 % through the span geometry for nothing
			const int64_t exp_win = b->div_est > 0 && g->div_aware ? std::min<int64_t>(len + 1, (int64_t)(6.2 * b->div_est * (double)len) + 64) : 0;
			const int64_t bound1 = penalty_bound(*opt, tl, ql, false);
			const int64_t bound = opt->max_s > 0 ? std::min<int64_t>(bound1, (int64_t)opt->max_s + 1) : bound1; // (= penalty_bound(..., true))
			const bool step0 = low_mem && bound1 < opt->step;
			int c = low_mem && !step0 ? 5 : 0;
			const bool packable = tl + bound < 32767 && pack_pen;
			if (classes && c == 0) {
				const int64_t window = std::min<int64_t>(len + 1, 2 * bound + 3);
				// A window cannot outgrow min(tl+ql+1, 2 x penalty bound + 3); in practice it stays far below tl+ql (a quarter of
				// it at 5 % divergence), so a pair is also given to a small kernel when it is merely short — if its window does
				// outgrow that span, finalize() moves it to the wide band kernel, and from there to the generic one.
				// (a pair too long for the packed band kernel goes to the generic kernel with 16-bit ring rows where those apply: faster than
				// the unpacked band kernel and no window overflows to re-run, see choose_kernel)
				// (... unless the span geometry of the packed kernel takes it: windows of up to ~16 000 columns — a 50 kb pair at 3 % —, which is
				// where the pairs whose windows will mostly fit are drawn: tl + ql below seven spans; bench.py long_batches)
				const bool span_ok = span_pen && tl <= kBandSpanMaxSeq && ql <= kBandSpanMaxSeq && !(know_acgt && !b->h_acgt[i]);
				if (span_ok && g->band_span == 2) c = 13;
				// (round 5: the limits leave the mean window at 5 % — 0.27 (tl+ql) — a quarter of margin below each class's widest window; round 4's left 4-18 %,
				// and 2 kb pairs, just inside the 128-thread class, were re-run at 7.8 %: profiles/r04/chooser_regression.txt)
				else if (packable && (window <= kBandMicroWindow || lenw + 1 <= 1400)) c = 4;
				else if (packable && (window <= kBandTinyWindow || lenw + 1 <= 3600)) c = 3;
				else if (packable && (window <= kBandSmallWindow || lenw + 1 <= 8200)) c = 2;
				else if (packable && (lenw + 1 <= 4 * (int64_t)(8 * 3 * 256) || window <= kBandWideWindow)) c = 1;
				// (tl + ql up to 3.5 of its spans: a 12 kb pair at 5 % needs ~6000 of the 7872 columns; 512 x 15 kb @ 5 % — windows of ~7500 — lost 44 pairs to late
				// overflows, 30.7 against 24.9 ms on the span geometry from the start)
				else if (span_ok && g->wide_slots != 3 && (exp_win ? exp_win + 768 <= ((int64_t)band2_biased512_chunks() + 8 - 1) * 256 - 64 : len + 1 <= 7 * (int64_t)((band2_biased512_chunks() + 8) * 256) / 2)) c = 14;
				else if (span_ok && ((exp_win ? exp_win + 768 <= band_span_window() : len + 1 <= 7 * band2_span_chunks() * 256) || window <= band_span_window())) c = 13;
			}
			b->h_class[i] = (int8_t)(c == 5 ? 0 : c == 13 ? 5 : c == 14 ? 1 : c);
			b->h_flags[i] = (int8_t)(step0 ? 1 : 0);
			// short pairs: a window of 64 diagonals holds them while the penalty stays below ~45 (a 200 bp pair at 5 %)
			// (in a batch small enough for the mid kernel the lane kernel keeps the pairs of up to 320 bases: 16 x 400 bp 0.31 ms on the lane
			// kernel — pairs that outgrow its chunks are re-run — against 0.13 on the mid kernel, 1 x 300 bp 56 against 68 us; profiles/r04/lane_vs_mid.txt)
			const bool to_lane = classes && lane_ok && c >= 1 && c <= 4 && (int64_t)((double)std::max(tl, ql) * div_r) <= (mid_ok ? std::min(g->lane_max_len, 320) : g->lane_max_len) && std::abs(tl - ql) <= 24;
			if (to_lane) c = (know_acgt && !b->h_acgt[i]) ? 12 : 10, b->h_class[i] = 4;
			// a few mid-size pairs: a workgroup each, rings in LDS (a penalty then costs a fraction of what it costs the band kernels).  Admitted
			// when the span the LDS can hold beside the sequences covers the window of a pair at ~6 % divergence (about 0.3 (tl+ql)); 16-bit offsets.
			if (mid_ok && !to_lane && c <= 4 && (!low_mem || step0) && tl + bound < 32760) {
				const int64_t seq_lds = ((tl + 7) & ~7LL) + 16 + ((ql + 7) & ~7LL) + 32;
				const int64_t window = std::min<int64_t>(len + 1, 2 * bound + 3);
				const int64_t want = std::min<int64_t>(window, lenw * 34 / 100 + 128) + 2 * P0.nH;
				int groups = (int)std::min<int64_t>((window + 2 * P0.nH + 63) / 64, 128);
				while (groups > 1 && mid_lds_bytes(P0, groups, seq_lds) > 158 * 1024) --groups;
				if (mid_lds_bytes(P0, groups, seq_lds) <= 158 * 1024 && (int64_t)groups * 64 >= want && std::abs(tl - ql) < groups * 32) {
					b->h_class[i] = (int8_t)((c >= 1 && c <= 4) || (classes && packable) ? 2 : 0); // where an overflow goes: the wide packed band kernel, else generic
					c = 11;
					mid_bytes |= know_acgt && !b->h_acgt[i]; // a pair the host knows not to be plain A/C/G/T: the (few) pairs of this class all take the byte-wise copy
				}
			}
			if (c >= 1 && c <= 4 && know_acgt && !b->h_acgt[i] && packable) c += 5;
			GroupInfo &G = gi[c];
			cls[i] = (int8_t)c, ++count[c];
			G.max_len = std::max(G.max_len, len), G.max_bound = std::max(G.max_bound, bound);
			G.max_bound1 = std::max(G.max_bound1, bound1);
			G.max_tl = std::max<int64_t>(G.max_tl, tl);
			G.max_seq_lds = std::max<int64_t>(G.max_seq_lds, ((tl + 3) & ~3LL) + 8 + ((ql + 3) & ~3LL) + 16);
		}
		// The processing order: groups in run order, longest first inside a group (the persistent workgroups finish together).  h_order is
		// already sorted longest first (batch_common) and that order is stable: one pass over it deals the pairs to their groups.
		std::vector<int32_t> start(15, 0), order((size_t)b->n);
		{
			int32_t at = 0;
			for (int c : run_order) start[c] = at, at += count[c], gi[c].n = count[c];
		}
		if (b->h_len_order.empty()) {
			b->h_len_order.resize((size_t)b->n);
			std::iota(b->h_len_order.begin(), b->h_len_order.end(), 0);
			std::stable_sort(b->h_len_order.begin(), b->h_len_order.end(), [&](int32_t x, int32_t y) {
				return (int64_t)b->h_tl[x] + b->h_ql[x] > (int64_t)b->h_tl[y] + b->h_ql[y];
			});
		}
		for (int32_t i : b->h_len_order) order[(size_t)start[cls[i]]++] = i;
		if (order != b->h_order) {
			b->h_order.swap(order);
			if (upload_segments(g, (char*)b->d_order, std::vector<Seg>{Seg{b->h_order.data(), b->h_order.size() * 4}})) return -1; // (waits for earlier work on the stream first)
		}
		PC.cls0 = b->h_class, PC.flags0 = b->h_flags, PC.mid_bytes = mid_bytes, PC.has_groups = true;
		for (int c = 0; c < 15; ++c) PC.gi[c] = gi[c];
	}
	int n_groups = 0, done_groups = 0;
	for (const GroupInfo &G : gi) n_groups += G.n > 0;
	mwf_opt_t opt_hi = *opt;
	opt_hi.step = 0;
	size_t at = 0;
	for (int c : run_order) {
		const GroupInfo &G = gi[c];
		if (G.n == 0) continue;
		++done_groups;
		int ran = 0;
		const int cc = c == 14 ? 8 : c == 13 ? 7 : c == 11 ? 6 : (c == 10 || c == 12) ? 5 : c > 5 ? c - 5 : c;
		g->acgt_off_once = (c > 5 && c < 10) || (c == 11 && mid_bytes) || c == 12;
		const int rc = run_batch_kernel(g, b, c == 5 ? *opt : opt_hi, b->d_order + at, G.n, slots, G.max_len, G.max_bound, G.max_bound1,
		                                done_groups == 1, (classes || c >= 11) ? (c == 0 || c == 5 ? 0 : 2) : -1, G.max_tl, G.max_seq_lds, done_groups == n_groups,
		                                cc == 8 ? 514 : cc == 7 ? 1024 : cc == 6 ? 33 : cc == 5 ? 32 : cc == 4 ? 64 : cc == 3 ? 128 : cc == 2 ? 256 : 0, &ran,
		                                (c == 1 && (g->wide_slots == 4 || (g->wide_slots == 0 && PC.wide_state != 1 && g->queue_clean))) ? kBandWide4Window : 0);
		if (c == 1 && g->wide_slots == 0 && PC.wide_state == 0 && g->queue_clean && ran == 2 && g->stats.block == 512) PC.wide_measured = true;
		g->acgt_off_once = false;
		if (rc) return -1;
		// (bit 64: the pair ran on the plain three-slot 512-thread geometry — the only one whose LATE overflow says "this batch's wide class needs four slots")
		const bool three_slots = c == 1 && ran == 2 && g->stats.block == 512 && !(g->wide_slots == 4 || (g->wide_slots == 0 && PC.wide_state != 1 && g->queue_clean));
		for (size_t j = at; j < at + (size_t)G.n; ++j) {
			const int32_t i = b->h_order[j];
			b->h_kind[i] = (int8_t)ran, b->h_flags[i] = (int8_t)((b->h_flags[i] & ~64) | (three_slots ? 64 : 0));
		}
		at += (size_t)G.n;
	}
	b->aligned = true;
	return 0;
}

} // extern "C"

namespace {

// Wait for the batch; re-run what did not fit where it ran:
//   window outgrew a band kernel's span   -> the wide band kernel, from there the generic kernel
//   traceback / snapshot arena too small  -> the same kernel on fewer workgroups (= larger slots)
//   whole-device kernel: a pair that shared the device gets it alone; the arena grows while memory lasts; a wait that
//   gave up (workgroups not resident) or a window beyond the device's span falls back to the generic kernel.
int finalize(mwf_gpu_t *g, mwf_gpu_batch_t *b)
{
	if (b->finalized) return 0;
	if (!b->aligned) { g->err = "batch was not aligned"; return -1; }
	const size_t n = (size_t)b->n;
	b->h_s.resize(n), b->h_ncig.resize(n), b->h_status.resize(n), b->h_iter.resize(n), b->h_cigoff.resize(n), b->h_cells1.resize(n);
	std::vector<char> &host = b->host_out;
	if (host.size() < b->out_bytes) host.resize(b->out_bytes);
	const BlockLayout L = layout_block(n, 0, false); // (only differences between result offsets are used)
	// a score-only, high-memory align: n_cigar, CIGAR offsets and first-pass cells are zero by construction — only head, status, s and
	// n_iter come back (16 of the 36 bytes per pair)
	const bool lean = !(b->opt.flag & MWF_F_CIGAR);
	const size_t need = lean ? b->out_bytes_score : b->out_bytes;
	auto fetch = [&]() -> int {
		if (n == 0) { HIP_TRY(g, hipStreamSynchronize(g->stream)); return 0; }
		if (b->out_in_pin) { // the kernels wrote the engine's pinned result page (score-only: no CIGAR words)
			HIP_TRY(g, hipStreamSynchronize(g->stream));
			memcpy(host.data(), g->res_pin, b->out_bytes);
			memset(host.data(), 0, 8);
			b->h_cig_valid = false;
		} else if (lean) {
			b->h_cig_valid = false;
			if (download(g, host.data(), (const char*)b->block.p + b->out_off, need)) return -1;
			memset(host.data() + need, 0, b->out_bytes - need);
		} else {
			// a small CIGAR-mode batch (the single pair of a drop-in call): the head of its CIGAR pool comes back with the results, one wait
			// for both copies — when the pool's used part turns out to fit it, fetch_cigars() has nothing left to copy
			const size_t spec = (b->opt.flag & MWF_F_CIGAR) && b->d_cig_pool && n <= 64 ? (size_t)std::min<int64_t>(b->cig_pool_words, 1024) : 0;
			const size_t cig_at = align_up(b->out_bytes, 64);
			b->h_cig_valid = false;
			if (spec > 0 && pin_reserve(g, cig_at + spec * 4) == 0) {
				HIP_TRY(g, hipMemcpyAsync(g->pin, (const char*)b->block.p + b->out_off, b->out_bytes, hipMemcpyDeviceToHost, g->stream));
				HIP_TRY(g, hipMemcpyAsync((char*)g->pin + cig_at, b->d_cig_pool, spec * 4, hipMemcpyDeviceToHost, g->stream));
				HIP_TRY(g, hipStreamSynchronize(g->stream));
				g->pin_busy[0] = g->pin_busy[1] = false;
				memcpy(host.data(), g->pin, b->out_bytes);
				unsigned long long used = 0;
				memcpy(&used, host.data(), 8);
				if (used <= spec) {
					b->h_cig.assign((const uint32_t*)((const char*)g->pin + cig_at), (const uint32_t*)((const char*)g->pin + cig_at) + used);
					b->h_cig_valid = true; // (a re-run below fetches again and decides again)
				}
			} else if (download(g, host.data(), (const char*)b->block.p + b->out_off, b->out_bytes)) return -1;
		}
		const char *o = host.data() - L.head;
		memcpy(&b->cig_used, o + L.head, 8);
		b->cig_used = std::min<int64_t>(b->cig_used, b->cig_pool_words); // (the head advances in whole blocks: its last step may point past the pool)
		memcpy(b->h_status.data(), o + L.status, n * 4), memcpy(b->h_s.data(), o + L.s, n * 4), memcpy(b->h_ncig.data(), o + L.ncig, n * 4);
		memcpy(b->h_iter.data(), o + L.iter, n * 8), memcpy(b->h_cigoff.data(), o + L.cigoff, n * 8), memcpy(b->h_cells1.data(), o + L.cells1, n * 8);
		return 0;
	};
	if (fetch()) return -1;
	if (b->plan.wide_measured) { // the four-slot kernels' report (reset to 0 by the align's reset kernel): did any pair need more than three slots hold?
		uint32_t aux = 0;
		memcpy(&aux, host.data() + 8, 4);
		if (b->plan.wide_state == 0) b->plan.wide_state = (aux & 1u) ? 2 : 1;
		b->plan.wide_measured = false;
	}
	b->busy = false;
	mwf_opt_t opt_hi = b->opt;
	opt_hi.step = 0;
	int tb_slots = std::max(1, b->last_grid);
	const int grid0 = std::max(1, b->last_grid);
	bool coop_warned = false;
	const char *fail = nullptr;
	for (int round = 0; round < 16 && !fail; ++round) {
		// where every unfinished pair goes next: route = kind (0 generic, 1 whole-device alone, 2 band) and, for the band kernel, the class
		std::vector<int32_t> to_generic[2], to_generic32[2], to_band_wide[2], to_band_span[2], to_band_bytes[2], same_fewer[3][2], coop_alone;
		bool grow_coop = false;
		for (size_t i = 0; i < n; ++i) {
			const int32_t st = b->h_status[i];
			if (st == ST_OK || st == ST_STOPPED) continue;
			const int kind = b->h_kind[i], step0 = b->h_flags[i] & 1;
			static const bool dbg_route = getenv("MWF_DEBUG_REROUTE") != nullptr; // (diagnostics: why a pair is run again)
			if (dbg_route) fprintf(stderr, "[libmwf_hip] re-route: pair %zu (tl %d ql %d) status %d kind %d class %d flags %d n_iter word %lld round %d\n", i, b->h_tl[i], b->h_ql[i], st, kind, (int)b->h_class[i], (int)b->h_flags[i], (long long)b->h_iter[i], round);
			if (st == ST_BAND_OVERFLOW && kind == 0) {
				to_generic32[step0].push_back((int32_t)i); // an offset outgrew the generic kernel's 16-bit ring rows: 32-bit rows
			} else if (st == ST_ALPHABET && kind == 2 && b->h_class[i] == 5) {
				b->h_class[i] = 0, to_generic[step0].push_back((int32_t)i); // (the span geometry has no byte-wise form)
			} else if (st == ST_ALPHABET && kind == 2) {
				to_band_bytes[step0].push_back((int32_t)i); // not plain ACGT: the byte-wise band kernel of the same class
			} else if (st == ST_BAND_OVERFLOW && kind == 2) {
				// (a pair handed back EARLY carries the window it is expected to need, negated, where n_iter would be: one that no band class
				// holds goes straight to the generic kernel)
				const int64_t est = b->h_iter[i] < 0 ? -b->h_iter[i] : 0;
				// (a pair of the wide class that outgrew its three chunk slots per wave LATE — that geometry carries no forecast — is re-run alone, ~7 ms for a
				// 10 kb pair beside the batch's 17: the next align of this batch takes the four-slot geometry for the class)
				// (only a pair that really ran on that geometry says so: class-14 pairs on biased offsets and re-runs of mid / lane pairs carry class 1 as well)
				if ((b->h_flags[i] & 64) && est == 0) b->plan.wide_state = 2;
				b->h_flags[i] &= ~64;
				// (... only when the forecast is half again beyond the widest class: it is an estimate, and the generic kernel is several times slower)
				// what outgrew (or is forecast to outgrow) the 512-thread geometry: the 1024-thread span geometry, if the pair fits that
				const bool span_ok = b->h_class[i] >= 1 && b->h_class[i] <= 4 && g->band_span != 0 && g->seq2bit != 0 && g->force_kind < 0 && g->block == 0 &&
				                     b->h_tl[i] <= kBandSpanMaxSeq && b->h_ql[i] <= kBandSpanMaxSeq && est <= band_span_window();
				// (a forecast the four-slot 512-thread geometry holds — the re-run takes four slots for it, see rerun(); beyond it the span geometry at once:
				// round 4 sent forecasts of up to 1.5 x the THREE-slot window here and re-ran them on three slots, which by their own forecast could not hold them)
				if (b->h_class[i] >= 2 && est <= kBandWide4Window) b->h_class[i] = 1, to_band_wide[step0].push_back((int32_t)i);
				else if (span_ok) b->h_class[i] = 5, to_band_span[step0].push_back((int32_t)i);
				// (a forecast also says whether the generic kernel's 16-bit ring rows can hold the pair — offsets up to 65 532, i.e. target length
				// + final penalty, about half the window: 50 kb pairs at 15 % ran them for nothing before taking the 32-bit rows)
				else if (est > 0 && (int64_t)b->h_tl[i] + est / 2 + 64 > 65000) b->h_class[i] = 0, to_generic32[step0].push_back((int32_t)i);
				else b->h_class[i] = 0, to_generic[step0].push_back((int32_t)i);
			} else if (kind == 1 && st == ST_INTERNAL && !(b->h_flags[i] & 8)) {
				// a wait between workgroups of the whole-device kernel ran into its spin limit (they were not all resident, e.g.
				// the device is shared): the one-workgroup kernel needs no such thing
				fprintf(stderr, "[libmwf_hip] warning: whole-device kernel gave up waiting for a workgroup on pair %d; re-running it on one workgroup (slow)\n", (int)i);
				b->h_flags[i] |= 8;
				to_generic[step0].push_back((int32_t)i);
			} else if (kind == 1 && (st == ST_BAND_OVERFLOW || st == ST_TB_OVERFLOW || st == ST_SNAP_OVERFLOW)) {
				if (b->h_flags[i] & 2) coop_alone.push_back((int32_t)i); // had a share of the workgroups and of the arena: now alone
				else if (st == ST_BAND_OVERFLOW && (b->h_flags[i] & 32) && !(b->h_flags[i] & 16)) {
					b->h_flags[i] |= 16; // its window outgrew the 64-column slots: again with 256-column ones
					coop_alone.push_back((int32_t)i);
				} else if (st == ST_BAND_OVERFLOW) {
					if (!coop_warned) fprintf(stderr, "[libmwf_hip] warning: wavefront of pair %d outgrew the whole-device kernel's span; re-running it on one workgroup (slow)\n", (int)i);
					coop_warned = true;
					to_generic[step0].push_back((int32_t)i);
				} else if (g->tb_budget_mb == 0 && g->coop_tb_mult < ((int64_t)1 << 20) && coop_can_grow(g)) grow_coop = true, coop_alone.push_back((int32_t)i);
				else if (b->opt.step > 0 && !step0) to_generic[0].push_back((int32_t)i); // the first-pass traceback does not fit: true two-pass mode
				else fail = "traceback";
			} else if (st == ST_TB_OVERFLOW || st == ST_SNAP_OVERFLOW) {
				if (tb_slots == 1) fail = st == ST_TB_OVERFLOW ? "traceback" : "low-memory snapshots"; // already had the whole budget
				same_fewer[kind == 2 ? 2 : 0][step0].push_back((int32_t)i);
			} else {
				g->err = "pair " + std::to_string(i) + " failed on the device with status " + std::to_string(st);
				return -3;
			}
			if (fail) {
				g->err = std::string(fail) + " of pair " + std::to_string(i) + " (tl=" + std::to_string(b->h_tl[i]) + ", ql=" + std::to_string(b->h_ql[i]) +
				         ") do not fit in device memory" + (b->opt.step > 0 ? "" : "; set opt.step > 0 (low-memory mode)");
				return -4;
			}
		}
		size_t n_redo = coop_alone.size();
		for (int z = 0; z < 2; ++z) n_redo += to_generic[z].size() + to_generic32[z].size() + to_band_wide[z].size() + to_band_span[z].size() + to_band_bytes[z].size() + same_fewer[0][z].size() + same_fewer[2][z].size();
		if (n_redo == 0) break;
		b->n_retries += (int32_t)n_redo;
		if (grow_coop) g->coop_tb_mult *= 2;
		for (int32_t i : coop_alone)
			if (run_coop_pair(g, b, b->opt, i, false, false)) return -1;
		const bool shrink = !same_fewer[0][0].empty() || !same_fewer[0][1].empty() || !same_fewer[2][0].empty() || !same_fewer[2][1].empty();
		if (shrink) tb_slots = std::max(1, tb_slots / 8);
		auto pl_low_mem = [](const mwf_opt_t &o) { return (o.flag & MWF_F_CIGAR) && o.step > 0; };
		auto rerun = [&](std::vector<int32_t> &ids, int step0, int want_kind, int slots, bool use_forecast = false, int geom = 0) -> int {
			if (ids.empty()) return 0;
			int64_t hint = 0;
			if (use_forecast) { // every pair of the re-run came back with a forecast: the class that holds the widest of them (+ 25 %)
				for (int32_t i : ids) {
					if (b->h_iter[i] >= 0) { hint = 0; break; }
					hint = std::max<int64_t>(hint, -b->h_iter[i] * 5 / 4 + 64);
				}
				// (every forecast of this re-run is at most kBandWide4Window: the margin must not push the hint past the four-slot geometry, back onto three slots)
				if (hint > kBandWide4Window && want_kind == 2 && geom == 0) hint = kBandWide4Window;
			}
			std::stable_sort(ids.begin(), ids.end(), [&](int32_t x, int32_t y) { return (int64_t)b->h_tl[x] + b->h_ql[x] > (int64_t)b->h_tl[y] + b->h_ql[y]; });
			const mwf_opt_t &o = step0 ? opt_hi : b->opt;
			int64_t max_len = 0, max_bound = 0, max_bound1 = 0, max_tl = 0, max_seq_lds = 0;
			for (int32_t i : ids) {
				max_len = std::max<int64_t>(max_len, (int64_t)b->h_tl[i] + b->h_ql[i]);
				max_bound = std::max(max_bound, penalty_bound(o, b->h_tl[i], b->h_ql[i], true));
				max_bound1 = std::max(max_bound1, penalty_bound(o, b->h_tl[i], b->h_ql[i], false));
				max_tl = std::max<int64_t>(max_tl, b->h_tl[i]);
				max_seq_lds = std::max<int64_t>(max_seq_lds, (((int64_t)b->h_tl[i] + 3) & ~3LL) + 8 + (((int64_t)b->h_ql[i] + 3) & ~3LL) + 16);
			}
			DevBuf &tmp = g->retry_ids; // the ids of this re-run (the previous re-run's kernels are through: every rerun() ends with a stream synchronisation)
			if (ensure(g, tmp, std::max<size_t>(ids.size() * 4, 4096))) return -1;
			int rc = upload_segments(g, (char*)tmp.p, std::vector<Seg>{Seg{ids.data(), ids.size() * 4}});
			int ran = 0;
			// a handful of short pairs that outgrew the lane kernel (one read in tens of thousands): the mid kernel, whose span holds the widest
			// window such a pair can have at all, takes a fraction of what a lone workgroup of the band classes takes (one 150 bp pair: 0.18 ms
			// of band kernel in every align of the 40 000-pair batch, profiles/r04/rocprof_lane_kernel_40000x150bp.txt)
			const Penalty Pm = make_penalty(o);
			const int mid_cap = g->mid_max_pairs < 0 ? g->n_cu : g->mid_max_pairs;
			const bool to_mid = want_kind == 2 && use_forecast && g->force_kind < 0 && g->block == 0 && (int)ids.size() <= mid_cap && mid_supported(Pm) && max_len <= 1200 &&
			                    max_tl + max_bound < 32760 && !(pl_low_mem(o));
			if (rc == 0) rc = run_batch_kernel(g, b, o, (const int32_t*)tmp.p, (int32_t)ids.size(), slots, max_len, max_bound, max_bound1, false,
			                                   want_kind, max_tl, max_seq_lds, 0, to_mid ? 33 : geom, &ran, hint);
			if (rc == 0) rc = hipStreamSynchronize(g->stream) == hipSuccess ? 0 : -1;
			if (rc) return -1;
			for (int32_t i : ids) b->h_kind[i] = (int8_t)ran;
			return 0;
		};
		const int wide = 1 << 30;
		for (int z = 0; z < 2; ++z) {
			if (rerun(to_generic[z], z, 0, grid0)) return -1;
			g->ring16_off_once = true;
			const int rc32 = rerun(to_generic32[z], z, 0, grid0);
			g->ring16_off_once = false;
			if (rc32) return -1;
			if (rerun(to_band_wide[z], z, 2, wide, true)) return -1;
			if (rerun(to_band_span[z], z, 2, wide, false, 1024)) return -1;
			g->acgt_off_once = true;
			const int rc_bytes = rerun(to_band_bytes[z], z, 2, wide);
			g->acgt_off_once = false;
			if (rc_bytes) return -1;
			if (rerun(same_fewer[0][z], z, 0, std::max(1, std::min<int>(tb_slots, (int)same_fewer[0][z].size())))) return -1;
			bool all_span = !same_fewer[2][z].empty();
			for (int32_t i : same_fewer[2][z]) all_span = all_span && b->h_class[i] == 5;
			if (rerun(same_fewer[2][z], z, 2, std::max(1, std::min<int>(tb_slots, (int)same_fewer[2][z].size())), false, all_span ? 1024 : 0)) return -1;
		}
		if (fetch()) return -1;
	}
	// nothing may be handed out as a result that is not one
	for (size_t i = 0; i < n; ++i)
		if (b->h_status[i] != ST_OK && b->h_status[i] != ST_STOPPED) {
			g->err = "pair " + std::to_string(i) + " is still unfinished after every retry (status " + std::to_string(b->h_status[i]) + ")";
			return -3;
		}
	g->stats.cells = 0, g->stats.cells_pass1 = 0, g->stats.n_retries = b->n_retries;
	for (size_t i = 0; i < n; ++i) g->stats.cells += b->h_iter[i], g->stats.cells_pass1 += b->h_cells1[i];
	b->finalized = true;
	return 0;
}

// every CIGAR of the batch in one copy (the used part of the pool)
int fetch_cigars(mwf_gpu_t *g, mwf_gpu_batch_t *b)
{
	if (b->h_cig_valid) return 0;
	if (int rc = finalize(g, b)) return rc;
	if (b->h_cig_valid) return 0; // came back with the results (small batch)
	b->h_cig.resize((size_t)std::max<int64_t>(b->cig_used, 0));
	if (b->cig_used > 0 && download(g, b->h_cig.data(), b->d_cig_pool, (size_t)b->cig_used * 4)) return -1;
	b->h_cig_valid = true;
	return 0;
}

} // namespace

extern "C" {

int mwf_gpu_batch_results(mwf_gpu_t *g, mwf_gpu_batch_t *b, int32_t *s, int64_t *n_iter, int32_t *n_cigar)
{
	if (!g || !b) return -1;
	(void)hipSetDevice(g->device);
	if (int rc = finalize(g, b)) return rc;
	const size_t n = (size_t)b->n;
	if (s && n) memcpy(s, b->h_s.data(), n * 4);
	if (n_iter && n) memcpy(n_iter, b->h_iter.data(), n * 8);
	if (n_cigar && n) memcpy(n_cigar, b->h_ncig.data(), n * 4);
	return 0;
}

const int32_t *mwf_gpu_batch_dev_scores(const mwf_gpu_batch_t *b) { return b ? b->d_s : nullptr; }
const int64_t *mwf_gpu_batch_dev_iters(const mwf_gpu_batch_t *b) { return b ? b->d_iter : nullptr; }
const int32_t *mwf_gpu_batch_dev_status(const mwf_gpu_batch_t *b) { return b ? b->d_status : nullptr; }

int32_t mwf_gpu_batch_cigar(mwf_gpu_t *g, mwf_gpu_batch_t *b, int32_t i, uint32_t *dst, int32_t cap)
{
	if (!g || !b || i < 0 || i >= b->n) return -1;
	(void)hipSetDevice(g->device);
	if (int rc = finalize(g, b)) return rc;
	const int32_t nc = b->h_ncig[i];
	if (nc > cap) return -5;
	if (nc <= 0) return nc;
	if (b->h_cig_valid) memcpy(dst, b->h_cig.data() + b->h_cigoff[i], (size_t)nc * 4);
	else HIP_TRY(g, hipMemcpy(dst, b->d_cig_pool + b->h_cigoff[i], (size_t)nc * 4, hipMemcpyDeviceToHost));
	return nc;
}

int mwf_gpu_batch_fetch_cigars(mwf_gpu_t *g, mwf_gpu_batch_t *b)
{
	if (!g || !b) return -1;
	(void)hipSetDevice(g->device);
	return fetch_cigars(g, b);
}

/* test hook: trace the band of one pair (columns lo,hi per penalty); returns penalties traced */
int32_t mwf_gpu_debug_band(mwf_gpu_t *g, mwf_gpu_batch_t *b, const mwf_opt_t *opt, int32_t pair, int32_t *lohi, int32_t cap)
{
	if (!g || !b) return -1;
	b->debug_pair = pair;
	int rc = mwf_gpu_batch_align(g, b, opt);
	if (rc == 0) rc = finalize(g, b);
	b->debug_pair = -1;
	if (rc) return rc;
	const int32_t n = std::min<int32_t>(cap, std::max(0, b->h_s[pair] >= 0 ? b->h_s[pair] : 0));
	if (n > 0) HIP_TRY(g, hipMemcpy(lohi, g->dbg.p, (size_t)n * 8, hipMemcpyDeviceToHost));
	return n;
}

/* ------------------------------------------------------------------ drop-in entry points */

void mwf_opt_init(mwf_opt_t *opt) // reference miniwfa.c:11-18
{
	memset(opt, 0, sizeof(*opt));
	opt->x = 4;
	opt->o1 = 4, opt->e1 = 2;
	opt->o2 = 15, opt->e2 = 1;
	opt->kmer = 13, opt->max_occ = 2, opt->min_len = 30;
}

} // extern "C"

namespace {

// Engines (stream + device pools) are pooled per device: a call takes an idle one — or creates one — and puts it back, so
// any number of host threads may call the drop-in API concurrently (the reference is re-entrant), each
// holding its own engine while it is inside a call, and an engine's warm pools serve whichever thread calls next.
// (Deliberately never destroyed: tearing streams down from static destructors can run after the HIP runtime's own teardown.)
struct EnginePool {
	std::mutex mu;
	std::vector<mwf_gpu_t*> idle[kMaxDevices];
};
EnginePool &engine_pool() { static EnginePool *p = new EnginePool(); return *p; }

mwf_gpu_t *acquire_engine(int dev)
{
	if (dev < 0 || dev >= kMaxDevices || dev >= mwf_gpu_device_count()) {
		char what[192];
		snprintf(what, sizeof(what), "device ordinal %d (MWF_DEVICE / devices[]) is not one of the %d visible gfx950 devices; this library has no CPU path", dev, mwf_gpu_device_count());
		fatal(what, nullptr);
	}
	EnginePool &P = engine_pool();
	{
		std::lock_guard<std::mutex> lock(P.mu);
		std::vector<mwf_gpu_t*> &v = P.idle[dev];
		if (!v.empty()) {
			mwf_gpu_t *g = v.back();
			v.pop_back();
			return g;
		}
	}
	mwf_gpu_t *g = mwf_gpu_create(dev, nullptr);
	if (!g) fatal("cannot open a HIP device; this library has no CPU path", nullptr);
	return g;
}

void release_engine(mwf_gpu_t *g)
{
	EnginePool &P = engine_pool();
	std::lock_guard<std::mutex> lock(P.mu);
	P.idle[g->device % kMaxDevices].push_back(g);
}

// a whole non-negative decimal number, or -1
int parse_ordinal(const char *e)
{
	if (!e || !*e) return -1;
	char *end = nullptr;
	const long v = strtol(e, &end, 10);
	return (*end == 0 && v >= 0 && v < 1 << 20) ? (int)v : -1;
}

int default_device()
{
	const char *e = getenv("MWF_DEVICE");
	if (!e) return 0;
	const int v = parse_ordinal(e);
	if (v < 0) fatal("MWF_DEVICE must be a device ordinal (a non-negative number)", e);
	return v;
}

// What one device produced for its share of a call, in host memory; the caller's thread turns it into mwf_rst_t's (kalloc
// arenas are not thread-safe, so nothing is allocated from `km` on a worker thread).
struct HostResult {
	std::vector<int32_t> s, ncig, dbg4;
	std::vector<int64_t> iter, cigoff;
	std::vector<uint32_t> cig;
	std::string err;
};

// Align pairs ids[0..m) of the caller's arrays on device `dev`.
void run_share(int dev, const mwf_opt_t *opt, const std::vector<int32_t> &ids, const int32_t *tl, const char *const *ts,
               const int32_t *ql, const char *const *qs, HostResult &R)
{
	const int32_t m = (int32_t)ids.size();
	if (m == 0) return;
	mwf_gpu_t *g = acquire_engine(dev);
	std::vector<int32_t> ltl((size_t)m), lql((size_t)m);
	std::vector<const char*> lts((size_t)m), lqs((size_t)m);
	for (int32_t j = 0; j < m; ++j) ltl[j] = tl[ids[j]], lql[j] = ql[ids[j]], lts[j] = ts[ids[j]], lqs[j] = qs[ids[j]];
	mwf_gpu_batch_t *b = batch_from_host(g, m, ltl.data(), lts.data(), lql.data(), lqs.data(), nullptr, 0, nullptr, nullptr);
	const bool cigar = (opt->flag & MWF_F_CIGAR) != 0;
	if (!b) R.err = std::string("batch upload failed: ") + mwf_gpu_last_error(g);
	else {
		R.s.resize((size_t)m), R.ncig.resize((size_t)m), R.iter.resize((size_t)m);
		if (mwf_gpu_batch_align(g, b, opt) || mwf_gpu_batch_results(g, b, R.s.data(), R.iter.data(), R.ncig.data()) || (cigar && fetch_cigars(g, b)))
			R.err = std::string("alignment failed: ") + mwf_gpu_last_error(g);
		else {
			if (cigar) R.cig.swap(b->h_cig), R.cigoff = b->h_cigoff;
			if ((opt->flag & MWF_F_DEBUG) && cigar) {
				R.dbg4.resize((size_t)m * 4);
				if (hipMemcpy(R.dbg4.data(), b->d_dbg4, (size_t)m * 16, hipMemcpyDeviceToHost) != hipSuccess) R.dbg4.clear();
			}
		}
		mwf_gpu_batch_free(b);
	}
	release_engine(g);
}

void fill_results(void *km, const mwf_opt_t *opt, const std::vector<int32_t> &ids, const HostResult &R, mwf_rst_t *r)
{
	for (size_t j = 0; j < ids.size(); ++j) {
		mwf_rst_t &o = r[ids[j]];
		memset(&o, 0, sizeof(mwf_rst_t)); // reference miniwfa.c:387
		o.s = R.s[j], o.n_iter = R.iter[j];
		if ((opt->flag & MWF_F_CIGAR) && R.s[j] >= 0 && R.ncig[j] > 0) {
			// reference krelocate()s the CIGAR into the caller's arena (miniwfa.c:434); a zero-length one stays NULL
			o.n_cigar = R.ncig[j];
			o.cigar = (uint32_t*)kmalloc(km, (size_t)R.ncig[j] * 4);
			memcpy(o.cigar, R.cig.data() + R.cigoff[j], (size_t)R.ncig[j] * 4);
		}
		if ((opt->flag & MWF_F_DEBUG) && !R.dbg4.empty() && R.s[j] >= 0) // reference miniwfa.c:367 prints the traceback end state
			fprintf(stderr, "s0=%d, s=%d, i=%d, k=%d\n", R.s[j] - 1, R.dbg4[4 * j], R.dbg4[4 * j + 1], R.dbg4[4 * j + 2]);
	}
}

} // namespace

extern "C" {

void mwf_wfa_batch_multi(void *km, const mwf_opt_t *opt, int32_t n, const int32_t *tl, const char *const *ts,
                         const int32_t *ql, const char *const *qs, mwf_rst_t *r, int32_t n_dev, const int32_t *devices)
{
	if (n <= 0) return;
	std::vector<int32_t> devs;
	if (devices && n_dev > 0) devs.assign(devices, devices + n_dev);
	else {
		const int have = mwf_gpu_device_count();
		if (have < 1) fatal("cannot open a HIP device; this library has no CPU path", nullptr);
		const int want = n_dev > 0 ? std::min<int>(n_dev, have) : have;
		for (int d = 0; d < want; ++d) devs.push_back(d);
	}
	const int D = (int)std::min<size_t>(devs.size(), (size_t)n);
	std::vector<std::vector<int32_t>> share((size_t)D);
	if (D == 1) {
		share[0].resize((size_t)n);
		std::iota(share[0].begin(), share[0].end(), 0);
	} else {
		// Longest first, each to the device with the least work so far.  Work of a pair ~ cells ~ s^2 ~ (tl+ql)^2 at equal
		// divergence; nothing crosses devices while they work.
		std::vector<int32_t> idx((size_t)n);
		std::iota(idx.begin(), idx.end(), 0);
		std::stable_sort(idx.begin(), idx.end(), [&](int32_t x, int32_t y) { return (int64_t)tl[x] + ql[x] > (int64_t)tl[y] + ql[y]; });
		std::vector<double> load((size_t)D, 0.0);
		for (int32_t i : idx) {
			const int d = (int)(std::min_element(load.begin(), load.end()) - load.begin());
			const double len = (double)tl[i] + ql[i] + 1;
			load[d] += len * len;
			share[d].push_back(i);
		}
	}
	std::vector<HostResult> res((size_t)D);
	if (D == 1) run_share(devs[0], opt, share[0], tl, ts, ql, qs, res[0]);
	else {
		std::vector<std::thread> th;
		for (int d = 1; d < D; ++d) th.emplace_back(run_share, devs[d], opt, std::cref(share[d]), tl, ts, ql, qs, std::ref(res[d]));
		run_share(devs[0], opt, share[0], tl, ts, ql, qs, res[0]);
		for (std::thread &t : th) t.join();
	}
	for (int d = 0; d < D; ++d)
		if (!res[d].err.empty()) fatal(res[d].err.c_str(), nullptr);
	for (int d = 0; d < D; ++d) fill_results(km, opt, share[d], res[d], r); // merged back in the caller's order
}

void mwf_wfa_batch(void *km, const mwf_opt_t *opt, int32_t n, const int32_t *tl, const char *const *ts,
                   const int32_t *ql, const char *const *qs, mwf_rst_t *r)
{
	// MWF_DEVICES = "all" or a count: deal the batch over that many devices; otherwise the one device MWF_DEVICE names
	const char *e = getenv("MWF_DEVICES");
	if (e && n > 1) {
		const int32_t k = !strcmp(e, "all") ? 0 : parse_ordinal(e);
		if (k < 0 || (k == 0 && strcmp(e, "all"))) fatal("MWF_DEVICES must be \"all\" or a positive device count", e);
		if (k != 1) { mwf_wfa_batch_multi(km, opt, n, tl, ts, ql, qs, r, k, nullptr); return; }
	}
	const int32_t dev = default_device();
	mwf_wfa_batch_multi(km, opt, n, tl, ts, ql, qs, r, 1, &dev);
}

void mwf_wfa_exact(void *km, const mwf_opt_t *opt, int32_t tl, const char *ts, int32_t ql, const char *qs, mwf_rst_t *r)
{
	const int32_t dev = default_device();
	mwf_wfa_batch_multi(km, opt, 1, &tl, &ts, &ql, &qs, r, 1, &dev);
}

// mwf_wfa_chain lives in mwf_chain.cpp

void mwf_wfa_auto(void *km, const mwf_opt_t *opt0, int32_t tl, const char *ts, int32_t ql, const char *qs, mwf_rst_t *r)
{
	mwf_opt_t opt = *opt0; // reference miniwfa.c:900-907
	opt.step = 0, opt.max_iter = 100000000;
	mwf_wfa_exact(km, &opt, tl, ts, ql, qs, r);
	if (r->s < 0) {
		if (opt.flag & MWF_F_CIGAR) opt.step = 5000;
		opt.max_iter = -1;
		mwf_wfa_chain(km, &opt, tl, ts, ql, qs, r);
	}
}

} // extern "C"
