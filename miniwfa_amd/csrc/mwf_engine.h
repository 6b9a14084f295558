// mwf_engine.h — what the three host files of libmwf_hip.so share: the engine and batch objects, the device-memory helpers
// (mwf_memory.cpp), the plan / launch / retry layer (mwf_plan.cpp) and the C ABI (mwf_engine.cpp).  Internal: nothing here is exported.
#pragma once
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

namespace mwf {
namespace host {


[[noreturn]] inline void fatal(const char *what, const char *detail)
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


} // namespace host
} // namespace mwf

using namespace mwf;
using namespace mwf::host;


struct mwf_gpu_s {
	int device = 0;
	hipStream_t stream = nullptr;
	bool own_stream = false;
	int n_cu = 0;
	size_t total_mem = 0;
	std::string err;
	// tunables
	int block = 0;              // 0: choose from the batch
	bool dev_retry = true;      // batches of reads: the pairs the lane kernel hands back are re-run by a follow-up launch from a device-side list, without the host
	int retry_mode = 0;         // set around run_batch_kernel(): 1 the launch fills the list, 2 the launch takes its pairs from it
	int retry_slot = 0;         // ... which of the batch's kRetrySlots lists
	bool band_fold = true;      // packed band kernel: the folded score-only form where the penalties allow it (o1 == x)
	bool div_aware = true;      // weigh the size classes' length limits by the batch's estimated divergence (batches built from host memory)
	int64_t tun_gen = 0;        // bumped by every successful mwf_gpu_set(): a cached plan of an align (PlanCache) is only replayed under the tunables it was made under
	int64_t coop_min_len = 0;
	int64_t tb_budget_mb = 0;   // 0: automatic
	int force_kind = -1;
	int res_pin_on = 1;        // small score-only batches: results written straight into pinned host memory (0: always copied back)
	int lane_chunks = 0;       // its window: 64-column chunks of LDS rows (1-4; 0: three for pairs of up to 400 bases of target + query, else four); a penalty only passes over the chunks the window has reached
	int lane_max_len = 325;    // (weighed by the batch's divergence / 5 % where known.  Round 3 set 400 against the band kernel of that round; against round 6's 64-thread geometry the lane kernel
	                           // wins up to ~370 bases at 5 %, ~200 at 10 %, everywhere at 2 %, and loses by half beyond: 2000 x 450 bp @ 5 % 1.46 against 0.77 ms, profiles/r06/lane_crossover.txt)
	                           // pairs whose longer sequence has at most this many bases try the one-diagonal-per-lane kernel first (0: never)
	int mid_max_pairs = -1;    // a batch of at most this many pairs may use the one-workgroup-per-pair, rings-in-LDS kernel (mwf_mid.hip) for its mid-size pairs (-1: one per CU; 0: never)
	int mid_block = 0;         // its threads per workgroup: 0 by span (256 up to 512 columns, else 1024), 256, 1024
	int band_pack = -1;        // int16-packed E/F registers in the band kernel: 0 never, otherwise whenever the value ranges allow
	int wide_slots = 0;        // chunk slots per wave of the 512-thread packed geometry: 0 by batch (four until an align has shown that three hold every pair), 3, 4
	int band_span = 1;         // the 1024-thread geometry of the packed band kernel (80 chunks, biased offsets: pairs of up to ~60 kb whose windows stay below ~20 000 columns): 0 never, 2: every pair it can take (tests)
	int ring16 = 1;            // generic kernel with E2/F2 in LDS: 16-bit ring rows in HBM while target length + penalty fits 16 bits (0: never)
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
		struct GI { int32_t n = 0; int64_t max_len = 0, max_bound = 0, max_bound1 = 0, max_tl = 0, max_seq_lds = 0, max_exp_win = 0; } gi[15];
		std::vector<int8_t> cls0, flags0;
	} plan;
	std::vector<char> host_out;     // the fixed-size results as they came back (finalize)
	size_t out_bytes_score = 0;     // leading part of the result region a score-only, high-memory align needs back: head, status, s, n_iter
	// geometry and counters of the last align of THIS batch (finalize() must not read the engine's: another batch may have
	// been aligned on the same engine in between)
	int32_t last_grid = 0, n_retries = 0;
	int32_t *d_retry_ids = nullptr;  // [kRetrySlots][kRetryCap] pairs a launch handed back for its follow-up launch; their count is the third word of the head (d_cig_head + 2)
	bool dev_retry_used = false;     // this align made such a follow-up launch: the count is added to n_retries
	// debug band trace (tests)
	int32_t debug_pair = -1;
};


#define HIP_TRY(g, call)                                                              \
	do {                                                                              \
		hipError_t e_ = (call);                                                       \
		if (e_ != hipSuccess) {                                                       \
			(g)->err = std::string(#call) + ": " + hipGetErrorString(e_);             \
			return -1;                                                                \
		}                                                                             \
	} while (0)

namespace mwf {
namespace host {

// ---- mwf_memory.cpp: device buffers that only grow, recycled batch blocks, the pinned staging buffer, batches from host memory
void account(mwf_gpu_t *g, int64_t delta);
int ensure(mwf_gpu_t *g, DevBuf &b, size_t bytes);
void release(mwf_gpu_t *g, DevBuf &b);
int take_block(mwf_gpu_t *g, DevBuf &spare, DevBuf &out, size_t bytes);
void give_block(mwf_gpu_t *g, DevBuf &spare, DevBuf &b);
int pin_reserve(mwf_gpu_t *g, size_t half);
struct Seg { const void *src; size_t len; }; // src == nullptr: `len` zero bytes
int upload_segments(mwf_gpu_t *g, char *dst, const std::vector<Seg> &segs);
int download(mwf_gpu_t *g, void *dst, const void *src, size_t bytes);
struct BlockLayout {
	size_t order = 0, t_off = 0, q_off = 0, tl = 0, ql = 0, seqs = 0, in_end = 0;
	size_t head = 0, status = 0, s = 0, ncig = 0, iter = 0, cigoff = 0, cells1 = 0, score_end = 0, out_end = 0, dbg4 = 0, retry = 0, total = 0;
};
constexpr int kRetryCap = 256; // ids a launch can hand to its follow-up launch on the device (BatchArgs::retry_ids)
constexpr int kRetrySlots = 2; // lists (and counters: words 2 and 3 of the head) per batch: one per lane class, so that the second class's follow-up never sees the first one's pairs
BlockLayout layout_block(size_t n, size_t seq_bytes, bool owned);
mwf_gpu_batch_t *batch_common(mwf_gpu_t *g, int32_t n, const int32_t *h_tl, const int32_t *h_ql, size_t seq_bytes, bool owned, BlockLayout &L);
mwf_gpu_batch_t *batch_from_host(mwf_gpu_t *g, int32_t n, const int32_t *tl, const char *const *ts, const int32_t *ql, const char *const *qs,
                                 const char *packed, int64_t packed_bytes, const int64_t *p_t_off, const int64_t *p_q_off);
float estimate_divergence_device(mwf_gpu_t *g, mwf_gpu_batch_t *b);
extern "C" int mwf_gpu_test_hook(mwf_gpu_t *g, const char *name, int64_t value); // mwf_engine.cpp: forced kernels / geometries / failure paths for tests/ and profiles/
// ---- mwf_async.cpp: submit / wait and the opt-in coalescing of single calls
int64_t coalesce_window_us();
void exact_coalesced(void *km, const mwf_opt_t *opt, int32_t tl, const char *ts, int32_t ql, const char *qs, mwf_rst_t *r); // 8-mer sketch of a few pairs of a device-resident batch (0: unknown)

// ---- mwf_plan.cpp: penalties, kernel choice, launches, the whole-device passes, re-runs
Penalty make_penalty(const mwf_opt_t &o);
const char *validate(const mwf_opt_t &o);
int64_t penalty_bound(const mwf_opt_t &o, int64_t tl, int64_t ql, bool honour_max_s);
int coop_grid_limit(mwf_gpu_t *g);
int finalize(mwf_gpu_t *g, mwf_gpu_batch_t *b);
int fetch_cigars(mwf_gpu_t *g, mwf_gpu_batch_t *b);

} // namespace host
} // namespace mwf
