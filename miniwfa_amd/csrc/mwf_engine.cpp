// mwf_engine.cpp — host side of libmwf_hip.so: the device engine (stream, memory pool, launch
// geometry, retry policy) and every entry point of include/miniwfa.h.
//
// Boundary: the reference's device-free API mwf_wfa_exact/auto/chain
// (miniwfa.c:603-615, :850-908) is kept; a call ships its pair(s) to HBM, runs the kernels of
// mwf_kernels.hip and copies back (s, n_iter, n_cigar, CIGAR).  r->cigar is allocated from the
// caller's kalloc arena exactly as the reference does (miniwfa.c:434).  kalloc arenas for
// scratch are replaced by one device workspace per engine that only ever grows (a per-stream
// hipMalloc pool): ring, traceback arena, row table, CIGAR scratch, snapshots.
//
// There is no CPU alignment path in this file or anywhere in the library.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
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
	int slots_per_cu = 0;       // 0: occupancy of the kernel
	int64_t coop_min_len = 0;
	int64_t tb_budget_mb = 0;   // 0: automatic
	int force_kind = -1;
	int band_pack = -1;        // int16-packed E/F registers in the band kernel: 0 never, otherwise whenever the value ranges allow
	int lds_e2 = 1;            // generic kernel: keep E2/F2 in LDS where that applies (0: never)
	int scalar_generic = 0;    // 1: the generic kernel's original one-column-per-lane pass everywhere (comparison / fallback)
	int64_t coop_spin_limit = 1 << 23; // polls (about a microsecond each) before the whole-device kernel gives up on a workgroup
	int64_t coop_tb_cap = (int64_t)96 << 30; // whole-device traceback arena: covers the 5 Mb pairs; doubles on overflow
	// workspace (per-stream pool)
	DevBuf ring, sring, good, tb, row_off, row_lo, cig_scratch, snap, snap_meta, seg, queue, dbg, coop_edge, coop_misc;
	hipEvent_t ev0 = nullptr, ev1 = nullptr;
	bool ev_pending = false;
	mwf_gpu_stats_t stats{};
};

struct mwf_gpu_batch_s {
	mwf_gpu_t *g = nullptr;
	int32_t n = 0;
	bool owns_inputs = false;
	const uint8_t *d_seqs = nullptr;
	int64_t seq_bytes = 0;
	const int64_t *d_t_off = nullptr, *d_q_off = nullptr;
	const int32_t *d_tl = nullptr, *d_ql = nullptr;
	std::vector<int32_t> h_tl, h_ql;
	int64_t max_seq_lds = 0;   // LDS bytes the band kernel needs to hold the longest pair's sequences
	int64_t max_tl = 0;        // longest target (offsets are target indices: bounds what a 16-bit offset must hold)
	int32_t *d_order = nullptr;
	std::vector<int32_t> h_order;   // what d_order holds: pair ids, grouped by size class, longest first inside a class
	bool coop_grouped = false;      // the last align ran several pairs side by side on the whole-device kernel
	std::vector<int8_t> h_class;    // size class of every pair in the last align (0 generic, then band kernels: 1 wide, 2 small, 3 tiny, 4 micro)
	// outputs
	int32_t *d_s = nullptr, *d_ncig = nullptr, *d_status = nullptr, *d_dbg4 = nullptr;
	int64_t *d_iter = nullptr, *d_cigoff = nullptr, *d_cells1 = nullptr;
	uint32_t *d_cig_pool = nullptr;
	int64_t cig_pool_words = 0;
	unsigned long long *d_cig_head = nullptr;
	// state of the last align
	bool aligned = false, finalized = false;
	mwf_opt_t opt{};
	std::vector<int32_t> h_s, h_ncig, h_status;
	std::vector<int64_t> h_iter, h_cigoff, h_cells1;
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

int ensure(mwf_gpu_t *g, DevBuf &b, size_t bytes)
{
	if (bytes <= b.bytes) return 0;
	if (b.p) {
		HIP_TRY(g, hipStreamSynchronize(g->stream));
		HIP_TRY(g, hipFree(b.p));
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
	return 0;
}

void release(DevBuf &b)
{
	if (b.p) (void)hipFree(b.p);
	b.p = nullptr, b.bytes = 0;
}

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
	if (std::max(o.x, std::max(o.o1 + o.e1, o.o2 + o.e2)) + 1 > kMaxRing) return "max(x, o1+e1, o2+e2) must be < 256";
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

struct Plan {
	int kind = 0;              // 0: generic kernel, 2: band kernel
	BandGeom band{0, 0, 0, 0};
	int block = 256, grid = 1;
	int32_t W = 0, GW = 0;
	int64_t ring_slot_ints = 0, rows_slot = 0, tb_slot_bytes = 0, cig_scratch_slot = 0;
	int64_t snap_slot_ints = 0, snap_meta_slot = 0, seg_slot = 0;
	bool low_mem = false, cigar = false;
};

// Run the one-workgroup-per-pair kernel over `n_items` pairs given by d_order (device) on `slots` workgroups.
// Which kernel serves a set of pairs.  The band kernel keeps E/F in registers and therefore only holds windows up to
// its span; it has no low-memory first pass.  kind: -1 automatic, 0 generic, 2 band.
void choose_kernel(const mwf_gpu_t *g, const mwf_opt_t &opt, const Penalty &P, int64_t max_len, int64_t max_bound,
                   int64_t max_seq_lds, int64_t max_tl, int want_kind, Plan &pl, int geom_block = 0)
{
	pl.kind = 0;
	const bool low_mem = (opt.flag & MWF_F_CIGAR) && opt.step > 0;
	if (want_kind == 0 || low_mem || !band_supported(P)) return;
	const int64_t max_window = std::min<int64_t>(max_len + 1, 2 * max_bound + 3);
	BandGeom bg;
	bg.packed = 0;
	// Packed variants (E/F registers as int16 pairs): valid when no offset (a target index, plus at most one per penalty for
	// offsets that ran past the matrix) and no penalty count can reach 32767.  They halve the state registers, which is
	// what lets several workgroups share a CU — one pair's barrier phase then overlaps another's compute:
	//   window <=  448:  64 threads x 3 chunks, sixteen pairs per CU (twelve with traceback): short reads
	//   window <= 1216: 128 threads x 3 chunks, eight pairs per CU (six with traceback)
	//   window <= 2752: 256 threads x 3 chunks, four (three)
	//   wider:          512 x 3, two per CU, score-only; with traceback 768 x 2 packed, one per CU (the 512-thread variant
	//                   needs more than its 128 VGPRs then: 52.9 ms against 60.3 ms unpacked on the 1024 x 10 kb batch)
	// Unpacked (long targets): 256 x 2 up to 1728 columns, 768 x 2 beyond.
	// (measured alternatives on the 1024 x 10 kb batch: 1024 threads x 2 chunks spills and runs 50 ms, 512 x 3 unpacked 49 ms, 768 x 2 42 ms)
	const bool range_ok = max_tl + max_bound < 32767 && g->band_pack != 0;
	const bool cigar = (opt.flag & MWF_F_CIGAR) != 0;
	if (range_ok) {
		bg.packed = 1;
		bg.block = max_window <= kBandMicroWindow ? 64 : max_window <= kBandTinyWindow ? 128 : max_window <= kBandSmallWindow ? 256 : cigar ? 768 : 512;
	} else bg.block = max_window <= 8 * 256 - 256 - 64 ? 256 : 768;
	// forced geometry (tests, tuning): 256 and 768 mean the unpacked variants unless packing is asked for as well
	if ((g->block == 64 || g->block == 128) && range_ok) bg.block = g->block, bg.packed = 1;
	if (g->block == 256) bg.block = 256, bg.packed = range_ok && g->band_pack == 1;
	if (g->block == 768) bg.block = 768, bg.packed = range_ok && cigar;
	if (g->block == 512 && range_ok) bg.block = 512, bg.packed = 1;
	// geometry picked by the caller for a size class (pairs short enough that their window should stay inside a small span)
	if (g->block == 0 && (geom_block == 64 || geom_block == 128) && range_ok) bg.block = geom_block, bg.packed = 1;
	if (g->block == 0 && geom_block == 256) bg.block = 256, bg.packed = range_ok;
	bg.span = bg.block / 64 * (bg.packed && bg.block != 768 ? 3 : 2) * 256;
	if (want_kind != 2 && max_len + 1 > 4 * (int64_t)bg.span) return; // windows will mostly outgrow the span: go generic at once
	const int64_t lds_cap = bg.block >= 768 ? 140 * 1024 : bg.block == 512 ? 70 * 1024 : bg.block == 256 ? 36 * 1024 : bg.block == 128 ? 18 * 1024 : 9 * 1024;
	bg.lds_bytes = max_seq_lds <= lds_cap ? (int)((max_seq_lds + 15) / 16 * 16) : 0;
	pl.kind = 2, pl.band = bg;
}

int run_batch_kernel(mwf_gpu_t *g, mwf_gpu_batch_t *b, const mwf_opt_t &opt, const int32_t *d_order, int32_t n_items,
                     int slots, int64_t max_len, int64_t max_bound, int64_t max_bound1, int64_t tb_total_budget, bool timed,
                     int want_kind = -1, int64_t max_tl = -1, int64_t max_seq_lds = -1, int timed_end = -1, int geom_block = 0)
{
	if (max_tl < 0) max_tl = b->max_tl;
	if (max_seq_lds < 0) max_seq_lds = b->max_seq_lds;
	const Penalty P = make_penalty(opt);
	Plan pl;
	pl.cigar = (opt.flag & MWF_F_CIGAR) != 0;
	pl.low_mem = pl.cigar && opt.step > 0;
	// generic kernel: four waves per pair, eight once the windows are wide (measured on 1250 x 50 kb: 708 ms against 782 ms)
	pl.block = g->block > 0 && g->block != 768 ? g->block : (std::min<int64_t>(max_len + 1, 2 * max_bound + 3) >= 8192 ? 512 : 256);
	choose_kernel(g, opt, P, max_len, max_bound, max_seq_lds, max_tl, want_kind >= 0 ? want_kind : g->force_kind, pl, geom_block);
	// `slots` is an upper bound from the caller (retries ask for fewer, larger slots); the chosen kernel's own residency
	// bounds it as well
	int per_cu, lds_e2_cols = 0;
	if (pl.kind == 2) {
		pl.block = pl.band.block;
		per_cu = g->slots_per_cu > 0 ? g->slots_per_cu : band_kernel_occupancy(P, pl.band, pl.cigar);
	} else {
		// wide windows (the 512-thread choice above), default gap extension: E2/F2 stay in LDS while the window fits 16 k columns
		if (g->lds_e2 && pl.block == 512 && g->block == 0 && P.e2 == 1 && !g->scalar_generic && !pl.low_mem) {
			lds_e2_cols = 16384;
			pl.block = 768; // one workgroup per CU either way (128 KB of LDS): twelve waves fit its 168-VGPR budget, 490 ms against 519 ms with eight
		}
		per_cu = g->slots_per_cu > 0 ? g->slots_per_cu : batch_kernel_occupancy(pl.block, !g->scalar_generic && !pl.low_mem, lds_e2_cols);
	}
	slots = std::max(1, std::min(slots, g->n_cu * std::max(1, per_cu)));
	if (getenv("MWF_DEBUG"))
		fprintf(stderr, "[libmwf_hip] kernel kind %d: block %d packed %d lds %d B, %d workgroup(s) per CU, %d slots\n", pl.kind, pl.block,
		        pl.band.packed, pl.band.lds_bytes, per_cu, slots);
	pl.grid = std::max(1, std::min<int>(slots, n_items));
	// row stride: whole 256-column chunks plus room for the band kernel's neighbour loads past the last chunk
	pl.W = (int32_t)((max_len + 3 + 255) / 256 * 256 + 512);
	pl.GW = pl.W / 64 + 2;
	pl.ring_slot_ints = (int64_t)(P.nH + 2 * P.n1 + 2 * P.n2) * pl.W;
	const size_t S = (size_t)pl.grid;

	if (ensure(g, g->ring, S * pl.ring_slot_ints * 4)) return -1;
	if (ensure(g, g->good, S * (size_t)P.nH * pl.GW * 8)) return -1;
	if (ensure(g, g->queue, 64)) return -1;
	if (pl.cigar) {
		pl.rows_slot = max_bound + 2;
		pl.cig_scratch_slot = max_len + 2;
		int64_t per = tb_total_budget / (int64_t)S;
		const int64_t worst = (max_bound + 1) * (max_len + 1); // every penalty as wide as the whole matrix
		if (pl.low_mem && opt.step > 2 * P.nH) {
			// the second pass collapses the band to one diagonal at every checkpoint (miniwfa.c:413-416) and consecutive
			// checkpoints are at most step+nH penalties apart, so a row is never wider than about 2*(step+nH)
			const int64_t seg_worst = (max_bound + 1) * std::min<int64_t>(max_len + 1, 2 * (int64_t)(opt.step + 2 * P.nH) + 8);
			per = std::min(per, seg_worst);
		}
		pl.tb_slot_bytes = std::max<int64_t>(4096, std::min(per, worst + 8 * (max_bound + 2))) / 4 * 4; // rows are padded to dwords
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
		const int64_t budget = (int64_t)(tb_total_budget / 4 / (int64_t)S);
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
	a.queue = (int32_t*)g->queue.p;
	a.scalar_generic = g->scalar_generic;
	a.lds_e2_cols = lds_e2_cols;
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
	a.snap = pl.low_mem ? (int32_t*)g->snap.p : nullptr, a.snap_slot_ints = pl.snap_slot_ints;
	a.snap_meta = pl.low_mem ? (int32_t*)g->snap_meta.p : nullptr, a.snap_meta_slot = pl.snap_meta_slot;
	a.seg = pl.low_mem ? (int32_t*)g->seg.p : nullptr, a.seg_slot = pl.seg_slot;
	a.out_s = b->d_s, a.out_iter = b->d_iter, a.out_ncig = b->d_ncig, a.out_cigoff = b->d_cigoff;
	a.out_status = b->d_status, a.out_cells1 = b->d_cells1, a.out_dbg = b->d_dbg4;
	a.dbg = b->debug_pair >= 0 ? (int32_t*)g->dbg.p : nullptr;
	a.dbg_cap = b->debug_pair >= 0 ? (int32_t)(g->dbg.bytes / 8) : 0;

	HIP_TRY(g, hipMemsetAsync(g->queue.p, 0, 64, g->stream));
	// HIP events bracket the kernel only: every workspace allocation above is already done
	if (timed) HIP_TRY(g, hipEventRecord(g->ev0, g->stream));
	const int lrc = pl.kind == 2 ? launch_band(a, pl.grid, pl.band, g->stream) : launch_batch(a, pl.grid, pl.block, g->stream);
	if (lrc != 0) {
		g->err = "kernel launch failed";
		return -1;
	}
	if (timed_end < 0 ? timed : timed_end != 0) { // the events bracket all launches of an align call, not the retries
		HIP_TRY(g, hipEventRecord(g->ev1, g->stream));
		g->ev_pending = true;
	}
	g->stats.n_launches += 1;
	g->stats.grid = std::max(g->stats.grid, pl.grid), g->stats.block = pl.block, g->stats.kernel_kind = pl.kind;
	return 0;
}


// One pair across the whole device (mwf_coop.hip).  Everything is enqueued on the stream: first pass, and in low-memory
// mode the checkpoint walk over its traceback matrix and the second pass, then traceback + outputs.
// Workgroups per pair when `n` pairs of at most `len` columns share the device: as many as the widest possible window can
// use when the pair is alone (it can then never outgrow them); when several pairs run side by side, as many as a window of
// a third of tl+ql needs (windows stay near a quarter at 3-5 % divergence) — a pair that does outgrow its group is re-run
// alone by finalize().  The per-penalty latency does not depend on the group size (C4-like 150 kb pair: 151 ms on 256
// workgroups, 141 ms on 64), so pairs side by side multiply the throughput.
int coop_group_size(int n_cu, int64_t len, bool alone)
{
	int G = n_cu;
	const int64_t chunks = alone ? (len >> 8) + 3 : (len >> 8) / 3 + 8;
	while (G > (alone ? 64 : 16) && chunks <= coop_chunk_slots(G / 2)) G /= 2;
	return G;
}

// Up to n_cu / group size pairs side by side on the whole-device kernel, each on its own group of workgroups (mwf_coop.hip).
// Everything is enqueued on the stream: first pass, and in low-memory mode the checkpoint walk over its traceback matrix
// and the second pass, then traceback + outputs.
int run_coop_group(mwf_gpu_t *g, mwf_gpu_batch_t *b, const mwf_opt_t &opt, const std::vector<int32_t> &pairs, int Gs, bool first, bool last)
{
	const Penalty P = make_penalty(opt);
	const bool cigar = (opt.flag & MWF_F_CIGAR) != 0, low_mem = cigar && opt.step > 0;
	const int n_groups = (int)pairs.size();
	if (n_groups < 1 || Gs < 1 || (int64_t)Gs * n_groups > std::min(coop_max_grid(cigar), g->n_cu)) { g->err = "whole-device kernel cannot be made resident"; return -1; }
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
	if (ensure(g, g->ring, NG * (size_t)P.nH * W * 4 + 4096)) return -1;
	if (ensure(g, g->good, NG * (size_t)P.nH * GW * 8)) return -1;
	// granules crossing waves: [nH][TC][2 sides][4] x 8 bytes; misc: flags, barrier words, pass state, then the flag ring
	const size_t gran_bytes = (size_t)P.nH * TC * 2 * 4 * 8;
	const size_t flag_ring_bytes = (size_t)64 * 32 * 128; // mwf_coop.hip: kFlagRing x kFlagCopies lines of 128 bytes
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
		const int64_t worst = NG * (rows_slot + 1) * (len + 8);
		int64_t want = std::min(worst, g->tb_budget_mb > 0 ? (g->tb_budget_mb << 20) : g->coop_tb_cap);
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
	if (traced) {
		if (ensure(g, g->dbg, (size_t)8 * (bound + 2))) return -1;
		HIP_TRY(g, hipMemsetAsync(g->dbg.p, 0, g->dbg.bytes, g->stream));
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
	a.out_s = b->d_s, a.out_iter = b->d_iter, a.out_ncig = b->d_ncig, a.out_cigoff = b->d_cigoff;
	a.out_status = b->d_status, a.out_cells1 = b->d_cells1, a.out_dbg = b->d_dbg4;
	a.dbg = traced && n_groups == 1 ? (int32_t*)g->dbg.p : nullptr; // the band trace is a single-pair diagnostic
	a.dbg_cap = a.dbg ? (int32_t)(g->dbg.bytes / 8) : 0;
	a.coop_pair = pairs[0];
	a.coop_spin_limit = (uint32_t)g->coop_spin_limit;
	a.coop_groups = n_groups, a.coop_group_size = Gs;
	a.coop_edge = (int32_t*)g->coop_edge.p, a.coop_edge_stride = (int64_t)(gran_bytes / 4);
	a.coop_flags = (int32_t*)g->coop_misc.p, a.coop_misc_stride = (int64_t)misc_bytes; // per group: flags | +1024 barrier words | +2048 pass state | +4096 flag ring
	a.coop_sync = (unsigned int*)((char*)g->coop_misc.p + 1024);
	a.coop_state = (int32_t*)((char*)g->coop_misc.p + 2048);
	int32_t *d_ids = (int32_t*)((char*)g->coop_misc.p + NG * misc_bytes);
	HIP_TRY(g, hipMemcpyAsync(d_ids, pairs.data(), NG * 4, hipMemcpyHostToDevice, g->stream));
	a.coop_pair_ids = d_ids;

	auto reset_sync = [&](bool all) -> int { // granule tags and the flag ring start out as "no penalty" (-1), counters at zero
		for (size_t q = 0; q < NG; ++q) {
			char *m = (char*)g->coop_misc.p + q * misc_bytes;
			if (all) { HIP_TRY(g, hipMemsetAsync(m, 0, 4096, g->stream)); }
			else HIP_TRY(g, hipMemsetAsync(m + 1024, 0, 1024, g->stream));
			HIP_TRY(g, hipMemsetAsync(m + 4096, 0xff, flag_ring_bytes, g->stream));
		}
		HIP_TRY(g, hipMemsetAsync(g->coop_edge.p, 0xff, NG * gran_bytes, g->stream));
		return 0;
	};
	if (reset_sync(true)) return -1;
	if (first) HIP_TRY(g, hipEventRecord(g->ev0, g->stream));
	a.coop_pass = low_mem ? 1 : 0;
	if (launch_coop_pass(a, Gs * n_groups, g->stream)) { g->err = "kernel launch failed (whole-device pass)"; return -1; }
	g->stats.n_launches += 1;
	if (low_mem) {
		if (launch_coop_walk(a, g->stream)) { g->err = "kernel launch failed (checkpoint walk)"; return -1; }
		if (reset_sync(false)) return -1; // barrier counters, flag ring and granules of the second pass
		a.coop_pass = 2;
		// the second pass is not traced: the band trace of a low-memory run is that of its second pass, traced below
		if (launch_coop_pass(a, Gs * n_groups, g->stream)) { g->err = "kernel launch failed (second pass)"; return -1; }
		g->stats.n_launches += 2;
	}
	if (launch_coop_finish(a, g->stream)) { g->err = "kernel launch failed (traceback)"; return -1; }
	g->stats.n_launches += 1;
	if (last) {
		HIP_TRY(g, hipEventRecord(g->ev1, g->stream));
		g->ev_pending = true;
	}
	g->stats.grid = Gs * n_groups, g->stats.block = 512, g->stats.kernel_kind = 1;
	return 0;
}

// one pair with the device to itself
int run_coop_pair(mwf_gpu_t *g, mwf_gpu_batch_t *b, const mwf_opt_t &opt, int32_t pair, bool first, bool last)
{
	const bool cigar = (opt.flag & MWF_F_CIGAR) != 0;
	const int G = coop_group_size(std::min(coop_max_grid(cigar), g->n_cu), (int64_t)b->h_tl[pair] + b->h_ql[pair], true);
	return run_coop_group(g, b, opt, std::vector<int32_t>{pair}, G, first, last);
}

// can the whole-device traceback arena still grow? (free memory beyond what it already holds)
bool coop_can_grow(mwf_gpu_t *g)
{
	size_t fr = 0, tot = 0;
	if (hipMemGetInfo(&fr, &tot) != hipSuccess) return false;
	return (int64_t)(fr / 10 * 9) + (int64_t)g->tb.bytes > (int64_t)g->tb.bytes + ((int64_t)1 << 30) && (int64_t)g->tb.bytes >= g->coop_tb_cap / 4 * 3;
}

int64_t tb_budget_bytes(mwf_gpu_t *g)
{
	if (g->tb_budget_mb > 0) return g->tb_budget_mb << 20;
	size_t fr = 0, tot = 0;
	if (hipMemGetInfo(&fr, &tot) != hipSuccess) fr = (size_t)8 << 30;
	// leave a fifth of what is free right now alone; the arena is kept between calls
	int64_t b = (int64_t)(fr / 5 * 4) + (int64_t)g->tb.bytes;
	return std::min<int64_t>(b, (int64_t)64 << 30);
}

int finalize(mwf_gpu_t *g, mwf_gpu_batch_t *b);

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
	return g;
}

void mwf_gpu_destroy(mwf_gpu_t *g)
{
	if (!g) return;
	(void)hipSetDevice(g->device);
	(void)hipStreamSynchronize(g->stream);
	for (DevBuf *b : {&g->ring, &g->sring, &g->good, &g->tb, &g->row_off, &g->row_lo, &g->cig_scratch, &g->snap, &g->snap_meta, &g->seg, &g->queue, &g->dbg, &g->coop_edge, &g->coop_misc})
		release(*b);
	if (g->ev0) (void)hipEventDestroy(g->ev0);
	if (g->ev1) (void)hipEventDestroy(g->ev1);
	if (g->own_stream) (void)hipStreamDestroy(g->stream);
	delete g;
}

const char *mwf_gpu_last_error(const mwf_gpu_t *g) { return g ? g->err.c_str() : "no engine (no gfx950 device could be opened)"; }

int mwf_gpu_set(mwf_gpu_t *g, const char *name, int64_t value)
{
	if (!g || !name) return -1;
	if (!strcmp(name, "block")) {
		if (value != 0 && value != 64 && value != 128 && value != 256 && value != 384 && value != 512 && value != 768 && value != 1024) return -1;
		g->block = (int)value;
	} else if (!strcmp(name, "slots_per_cu")) g->slots_per_cu = (int)value;
	else if (!strcmp(name, "coop_min_len")) g->coop_min_len = value;
	else if (!strcmp(name, "tb_budget_mb")) g->tb_budget_mb = value;
	else if (!strcmp(name, "force_kind")) g->force_kind = (int)value;
	else if (!strcmp(name, "band_pack")) g->band_pack = (int)value;
	else if (!strcmp(name, "lds_e2")) g->lds_e2 = value != 0;
	else if (!strcmp(name, "scalar_generic")) g->scalar_generic = value != 0;
	else if (!strcmp(name, "coop_spin_limit")) g->coop_spin_limit = std::max<int64_t>(0, std::min<int64_t>(value, 0x7fffffff));
	else if (!strcmp(name, "coop_tb_cap_mb")) g->coop_tb_cap = std::max<int64_t>(1, value) << 20;
	else return -1;
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
	*st = m->stats;
}

/* ------------------------------------------------------------------ batches */

static mwf_gpu_batch_t *batch_common(mwf_gpu_t *g, int32_t n, const int32_t *h_tl, const int32_t *h_ql)
{
	mwf_gpu_batch_t *b = new mwf_gpu_batch_t();
	b->g = g, b->n = n;
	b->h_tl.assign(h_tl, h_tl + n);
	b->h_ql.assign(h_ql, h_ql + n);
	const size_t N = (size_t)std::max(n, 1);
	int64_t words = 0;
	for (int32_t i = 0; i < n; ++i) {
		words += (int64_t)h_tl[i] + h_ql[i] + 1;
		b->max_tl = std::max<int64_t>(b->max_tl, h_tl[i]);
		b->max_seq_lds = std::max<int64_t>(b->max_seq_lds, (((int64_t)h_tl[i] + 3) & ~3LL) + 8 + (((int64_t)h_ql[i] + 3) & ~3LL) + 16);
	}
	b->cig_pool_words = std::max<int64_t>(words, 1);
	bool ok = hipMalloc(&b->d_s, N * 4) == hipSuccess && hipMalloc(&b->d_ncig, N * 4) == hipSuccess &&
	          hipMalloc(&b->d_status, N * 4) == hipSuccess && hipMalloc(&b->d_dbg4, N * 16) == hipSuccess &&
	          hipMalloc(&b->d_iter, N * 8) == hipSuccess && hipMalloc(&b->d_cigoff, N * 8) == hipSuccess &&
	          hipMalloc(&b->d_cells1, N * 8) == hipSuccess && hipMalloc(&b->d_order, N * 4) == hipSuccess &&
	          hipMalloc(&b->d_cig_head, 64) == hipSuccess;
	if (!ok) {
		g->err = "hipMalloc of result arrays failed";
		mwf_gpu_batch_free(b);
		return nullptr;
	}
	// longest pairs first, so the persistent workgroups finish together
	std::vector<int32_t> order(n);
	std::iota(order.begin(), order.end(), 0);
	std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) {
		return (int64_t)h_tl[x] + h_ql[x] > (int64_t)h_tl[y] + h_ql[y];
	});
	b->h_order = order;
	if (n > 0 && hipMemcpy(b->d_order, order.data(), (size_t)n * 4, hipMemcpyHostToDevice) != hipSuccess) {
		g->err = "upload of the processing order failed";
		mwf_gpu_batch_free(b);
		return nullptr;
	}
	return b;
}

mwf_gpu_batch_t *mwf_gpu_batch_upload(mwf_gpu_t *g, int32_t n, const char *seqs, int64_t seq_bytes,
                                      const int64_t *t_off, const int32_t *tl, const int64_t *q_off, const int32_t *ql)
{
	if (!g || n < 0) return nullptr;
	(void)hipSetDevice(g->device);
	mwf_gpu_batch_t *b = batch_common(g, n, tl, ql);
	if (!b) return nullptr;
	b->owns_inputs = true;
	b->seq_bytes = seq_bytes;
	const size_t N = (size_t)std::max(n, 1);
	void *ds = nullptr, *dto = nullptr, *dqo = nullptr, *dtl = nullptr, *dql = nullptr;
	bool ok = hipMalloc(&ds, (size_t)seq_bytes + 64) == hipSuccess && hipMalloc(&dto, N * 8) == hipSuccess &&
	          hipMalloc(&dqo, N * 8) == hipSuccess && hipMalloc(&dtl, N * 4) == hipSuccess && hipMalloc(&dql, N * 4) == hipSuccess;
	b->d_seqs = (const uint8_t*)ds, b->d_t_off = (const int64_t*)dto, b->d_q_off = (const int64_t*)dqo;
	b->d_tl = (const int32_t*)dtl, b->d_ql = (const int32_t*)dql;
	if (ok) ok = hipMemsetAsync(ds, 0, (size_t)seq_bytes + 64, g->stream) == hipSuccess;
	if (ok && seq_bytes > 0) ok = hipMemcpyAsync(ds, seqs, (size_t)seq_bytes, hipMemcpyHostToDevice, g->stream) == hipSuccess;
	if (ok && n > 0)
		ok = hipMemcpyAsync(dto, t_off, (size_t)n * 8, hipMemcpyHostToDevice, g->stream) == hipSuccess &&
		     hipMemcpyAsync(dqo, q_off, (size_t)n * 8, hipMemcpyHostToDevice, g->stream) == hipSuccess &&
		     hipMemcpyAsync(dtl, tl, (size_t)n * 4, hipMemcpyHostToDevice, g->stream) == hipSuccess &&
		     hipMemcpyAsync(dql, ql, (size_t)n * 4, hipMemcpyHostToDevice, g->stream) == hipSuccess;
	if (ok) ok = hipStreamSynchronize(g->stream) == hipSuccess; // the host buffers are borrowed only for this call
	if (!ok) {
		g->err = "upload of the batch failed";
		mwf_gpu_batch_free(b);
		return nullptr;
	}
	return b;
}

mwf_gpu_batch_t *mwf_gpu_batch_wrap(mwf_gpu_t *g, int32_t n, const void *d_seqs, int64_t seq_bytes,
                                    const int64_t *d_t_off, const int32_t *d_tl, const int64_t *d_q_off, const int32_t *d_ql,
                                    const int32_t *h_tl, const int32_t *h_ql)
{
	if (!g || n < 0) return nullptr;
	(void)hipSetDevice(g->device);
	mwf_gpu_batch_t *b = batch_common(g, n, h_tl, h_ql);
	if (!b) return nullptr;
	b->owns_inputs = false;
	b->d_seqs = (const uint8_t*)d_seqs, b->seq_bytes = seq_bytes;
	b->d_t_off = d_t_off, b->d_q_off = d_q_off, b->d_tl = d_tl, b->d_ql = d_ql;
	return b;
}

void mwf_gpu_batch_free(mwf_gpu_batch_t *b)
{
	if (!b) return;
	(void)hipSetDevice(b->g->device);
	(void)hipStreamSynchronize(b->g->stream);
	if (b->owns_inputs)
		for (const void *p : {(const void*)b->d_seqs, (const void*)b->d_t_off, (const void*)b->d_q_off, (const void*)b->d_tl, (const void*)b->d_ql})
			if (p) (void)hipFree(const_cast<void*>(p));
	for (void *p : {(void*)b->d_s, (void*)b->d_ncig, (void*)b->d_status, (void*)b->d_dbg4, (void*)b->d_iter, (void*)b->d_cigoff,
	                (void*)b->d_cells1, (void*)b->d_order, (void*)b->d_cig_pool, (void*)b->d_cig_head})
		if (p) (void)hipFree(p);
	delete b;
}

int mwf_gpu_batch_align(mwf_gpu_t *g, mwf_gpu_batch_t *b, const mwf_opt_t *opt)
{
	if (!g || !b || !opt) return -1;
	if (const char *why = validate(*opt)) { g->err = why; return -2; }
	(void)hipSetDevice(g->device);
	b->opt = *opt;
	b->aligned = false, b->finalized = false;
	g->stats = mwf_gpu_stats_t{};
	if (b->n == 0) { b->aligned = b->finalized = true; return 0; }
	const bool cigar = (opt->flag & MWF_F_CIGAR) != 0;
	if (cigar && !b->d_cig_pool) {
		if (hipMalloc(&b->d_cig_pool, (size_t)b->cig_pool_words * 4) != hipSuccess) { g->err = "hipMalloc of the CIGAR pool failed"; return -1; }
	}
	int64_t max_len = 0, max_bound = 0, max_bound1 = 0;
	for (int32_t i = 0; i < b->n; ++i) {
		max_len = std::max<int64_t>(max_len, (int64_t)b->h_tl[i] + b->h_ql[i]);
		max_bound = std::max(max_bound, penalty_bound(*opt, b->h_tl[i], b->h_ql[i], true));
		max_bound1 = std::max(max_bound1, penalty_bound(*opt, b->h_tl[i], b->h_ql[i], false));
	}
	if (max_len + 4 >= ((int64_t)1 << 31)) { g->err = "tl+ql must be below 2^31-4"; return -2; }
	const int slots = 1 << 30; // as many as the chosen kernel can keep resident (run_batch_kernel bounds it)
	if (b->debug_pair >= 0 && ensure(g, g->dbg, (size_t)8 * (max_bound + 2))) return -1;
	HIP_TRY(g, hipMemsetAsync(b->d_cig_head, 0, 64, g->stream));
	HIP_TRY(g, hipMemsetAsync(b->d_status, 0xff, (size_t)b->n * 4, g->stream));
	// a few long pairs: each one gets the whole device in turn
	const Penalty P0 = make_penalty(*opt);
	const int64_t coop_len = g->coop_min_len > 0 ? g->coop_min_len : 65536;
	// The whole-device kernel takes time ~ (tl+ql) per pair — and several pairs fit side by side, each on its own group of
	// workgroups; the generic kernel runs up to 256 pairs side by side in time ~ (tl+ql)^2.  Measured at 3 % divergence
	// (profiles/few_long_pairs.py): 100 kb pairs 88 ms each against 250 ms for any number of them, 150 kb pairs 128 ms
	// against 550 ms — the whole-device kernel wins while the batch has fewer than about (tl+ql)/70000 pairs per group.
	const int n_cu_coop = std::min(coop_max_grid(cigar), g->n_cu);
	const int coop_side_by_side = n_cu_coop > 0 ? std::max(1, n_cu_coop / coop_group_size(n_cu_coop, max_len, false)) : 1;
	const int64_t coop_max_pairs = std::max<int64_t>(1, std::min<int64_t>(256, max_len / 70000 * coop_side_by_side));
	const bool coop = g->force_kind == 1 || (g->force_kind < 0 && coop_supported(P0) && b->n <= coop_max_pairs && max_len >= coop_len);
	b->coop_grouped = false;
	if (coop) {
		if (!coop_supported(P0)) { g->err = "whole-device kernel does not support these penalties"; return -2; }
		if (n_cu_coop < 1) { g->err = "whole-device kernel cannot be made resident"; return -1; }
		std::vector<int32_t> idx(b->h_order.begin(), b->h_order.end()); // longest first
		std::stable_sort(idx.begin(), idx.end(), [&](int32_t x, int32_t y) { return (int64_t)b->h_tl[x] + b->h_ql[x] > (int64_t)b->h_tl[y] + b->h_ql[y]; });
		for (size_t at = 0; at < idx.size();) {
			const int64_t len0 = (int64_t)b->h_tl[idx[at]] + b->h_ql[idx[at]];
			const int Gs = coop_group_size(n_cu_coop, len0, false);
			const size_t n_side = std::min<size_t>(idx.size() - at, (size_t)std::max(1, n_cu_coop / Gs));
			const bool last = at + n_side == idx.size();
			if (n_side <= 1 || b->debug_pair >= 0) { // alone (also: band traces are single-pair diagnostics)
				if (run_coop_pair(g, b, *opt, idx[at], at == 0, at + 1 == idx.size())) return -1;
				at += 1;
				continue;
			}
			b->coop_grouped = true;
			if (run_coop_group(g, b, *opt, std::vector<int32_t>(idx.begin() + at, idx.begin() + at + n_side), Gs, at == 0, last)) return -1;
			at += n_side;
		}
		b->aligned = true;
		return 0;
	}
	const int64_t budget = cigar ? tb_budget_bytes(g) : 0;
	// Size classes.  One long pair must not push a thousand short ones onto the slow kernel (mwf_wfa_chain's gap fills are
	// exactly such a mix): pairs are grouped by what their window can grow to, and every group runs on the kernel that suits
	// it — generic (largest workspace) first, so that later groups never have to grow a buffer.  Kernel and block size forced
	// by the caller (tests, tuning) keep the whole batch in one group.
	const bool low_mem = cigar && opt->step > 0;
	const bool classes = g->force_kind < 0 && g->block == 0 && !low_mem && band_supported(P0);
	struct Group { std::vector<int32_t> ids; int64_t max_len = 0, max_bound = 0, max_bound1 = 0, max_tl = 0, max_seq_lds = 0; } grp[5];
	b->h_class.assign((size_t)b->n, 0);
	for (int32_t i = 0; i < b->n; ++i) {
		const int64_t len = (int64_t)b->h_tl[i] + b->h_ql[i], bound = penalty_bound(*opt, b->h_tl[i], b->h_ql[i], true);
		int c = 0;
		if (classes) {
			const int64_t window = std::min<int64_t>(len + 1, 2 * bound + 3);
			const bool packable = (int64_t)b->h_tl[i] + bound < 32767 && g->band_pack != 0;
			// A window cannot outgrow min(tl+ql+1, 2 x penalty bound + 3); in practice it stays far below tl+ql (a quarter of
			// it at 5 % divergence), so a pair is also given to a small kernel when it is merely short — if its window does
			// outgrow that span, finalize() moves it to the wide band kernel, and from there to the generic one.
			if (packable && (window <= kBandMicroWindow || len + 1 <= 2 * (int64_t)(1 * 3 * 256))) c = 4;
			else if (packable && (window <= kBandTinyWindow || len + 1 <= 3 * (int64_t)(2 * 3 * 256))) c = 3;
			else if (packable ? (window <= kBandSmallWindow || len + 1 <= 3 * (int64_t)(4 * 3 * 256)) : window <= 8 * 256 - 256 - 64) c = 2;
			else if (len + 1 <= 4 * (int64_t)(8 * 3 * 256)) c = 1;
		}
		b->h_class[i] = (int8_t)c;
		Group &G = grp[c];
		G.ids.push_back(i);
		G.max_len = std::max(G.max_len, len), G.max_bound = std::max(G.max_bound, bound);
		G.max_bound1 = std::max(G.max_bound1, penalty_bound(*opt, b->h_tl[i], b->h_ql[i], false));
		G.max_tl = std::max<int64_t>(G.max_tl, b->h_tl[i]);
		G.max_seq_lds = std::max<int64_t>(G.max_seq_lds, (((int64_t)b->h_tl[i] + 3) & ~3LL) + 8 + (((int64_t)b->h_ql[i] + 3) & ~3LL) + 16);
	}
	std::vector<int32_t> order;
	order.reserve((size_t)b->n);
	for (Group &G : grp) {
		std::stable_sort(G.ids.begin(), G.ids.end(), [&](int32_t x, int32_t y) { // longest first: the persistent workgroups finish together
			return (int64_t)b->h_tl[x] + b->h_ql[x] > (int64_t)b->h_tl[y] + b->h_ql[y];
		});
		order.insert(order.end(), G.ids.begin(), G.ids.end());
	}
	if (order != b->h_order) {
		HIP_TRY(g, hipStreamSynchronize(g->stream)); // an earlier align of this batch may still be reading the old order
		HIP_TRY(g, hipMemcpy(b->d_order, order.data(), order.size() * 4, hipMemcpyHostToDevice));
		b->h_order = order;
	}
	int n_groups = 0, done_groups = 0;
	for (const Group &G : grp) n_groups += !G.ids.empty();
	size_t at = 0;
	for (int c = 0; c < 5; ++c) {
		const Group &G = grp[c];
		if (G.ids.empty()) continue;
		++done_groups;
		if (run_batch_kernel(g, b, *opt, b->d_order + at, (int32_t)G.ids.size(), slots, G.max_len, G.max_bound, G.max_bound1, budget,
		                     done_groups == 1, classes ? (c == 0 ? 0 : 2) : -1, G.max_tl, G.max_seq_lds, done_groups == n_groups,
		                     c == 4 ? 64 : c == 3 ? 128 : c == 2 ? 256 : 0)) return -1;
		at += G.ids.size();
	}
	b->aligned = true;
	return 0;
}

} // extern "C"

namespace {

// Wait for the batch, re-run pairs whose traceback arena overflowed with fewer, larger slots.
int finalize(mwf_gpu_t *g, mwf_gpu_batch_t *b)
{
	if (b->finalized) return 0;
	if (!b->aligned) { g->err = "batch was not aligned"; return -1; }
	const size_t n = (size_t)b->n;
	b->h_s.resize(n), b->h_ncig.resize(n), b->h_status.resize(n), b->h_iter.resize(n), b->h_cigoff.resize(n), b->h_cells1.resize(n);
	auto fetch = [&]() -> int {
		HIP_TRY(g, hipStreamSynchronize(g->stream));
		if (n == 0) return 0;
		HIP_TRY(g, hipMemcpy(b->h_status.data(), b->d_status, n * 4, hipMemcpyDeviceToHost));
		HIP_TRY(g, hipMemcpy(b->h_s.data(), b->d_s, n * 4, hipMemcpyDeviceToHost));
		HIP_TRY(g, hipMemcpy(b->h_iter.data(), b->d_iter, n * 8, hipMemcpyDeviceToHost));
		HIP_TRY(g, hipMemcpy(b->h_ncig.data(), b->d_ncig, n * 4, hipMemcpyDeviceToHost));
		HIP_TRY(g, hipMemcpy(b->h_cigoff.data(), b->d_cigoff, n * 8, hipMemcpyDeviceToHost));
		HIP_TRY(g, hipMemcpy(b->h_cells1.data(), b->d_cells1, n * 8, hipMemcpyDeviceToHost));
		return 0;
	};
	if (fetch()) return -1;
	int slots = g->stats.grid, redo_kind = g->stats.kernel_kind == 2 ? 2 : 0;
	bool coop_fell_back = false, coop_gave_up = false, coop_gave_up_now = false;
	for (int round = 0; round < 14; ++round) {
		std::vector<int32_t> redo;
		bool band_overflow = false;
		for (size_t i = 0; i < n; ++i) {
			const int32_t st = b->h_status[i];
			if (st == ST_BAND_OVERFLOW) band_overflow = true, redo.push_back((int32_t)i);
			else if (st == ST_INTERNAL && g->stats.kernel_kind == 1 && round <= 1 && !coop_gave_up) {
				// a wait between workgroups of the whole-device kernel ran into its spin limit (they were not all resident, e.g.
				// the device is shared): the one-workgroup kernel needs no such thing
				fprintf(stderr, "[libmwf_hip] warning: whole-device kernel gave up waiting for a workgroup on pair %d; re-running it on one workgroup (slow)\n", (int)i);
				band_overflow = true, redo.push_back((int32_t)i);
				coop_gave_up_now = true;
			}
			else if (st == ST_TB_OVERFLOW || st == ST_SNAP_OVERFLOW) redo.push_back((int32_t)i);
			else if (st != ST_OK && st != ST_STOPPED) {
				g->err = "pair " + std::to_string(i) + " failed on the device with status " + std::to_string(st);
				return -3;
			}
		}
		if (redo.empty()) break;
		if (g->stats.kernel_kind == 1 && b->coop_grouped && !coop_gave_up_now) {
			// pairs that ran side by side had a share of the workgroups and of the traceback arena: whatever did not fit gets
			// the device to itself (and from there the usual remedies)
			b->coop_grouped = false;
			for (int32_t i : redo)
				if (run_coop_pair(g, b, b->opt, i, false, false)) return -1;
			g->stats.n_retries += (int32_t)redo.size();
			if (fetch()) return -1;
			continue;
		}
		if (round == 0 && g->stats.kernel_kind != 1 && !b->h_class.empty()) {
			// the batch ran in size classes: stay on the band kernel only if every pair to redo came from one
			bool any_class = false, all_band = true;
			for (int8_t c : b->h_class) any_class |= c != 0;
			for (int32_t i : redo) all_band &= b->h_class[i] != 0;
			if (any_class) redo_kind = all_band ? 2 : 0;
		}
		if (coop_gave_up_now) coop_gave_up = true, coop_gave_up_now = false, b->coop_grouped = false;
		if (band_overflow) {
			redo_kind = 0; // the window outgrew the register-resident span: generic kernel, same slots
			// ... unless the pair sat in one of the small size classes: then the wide band kernel first (the rest waits a round)
			std::vector<int32_t> promote;
			if (g->stats.kernel_kind != 1)
				for (int32_t i : redo)
					if (b->h_status[i] == ST_BAND_OVERFLOW && (size_t)i < b->h_class.size() && b->h_class[i] >= 2) promote.push_back(i);
			if (!promote.empty()) {
				for (int32_t i : promote) b->h_class[i] = 1;
				redo.swap(promote);
				redo_kind = 2;
			}
			if (g->stats.kernel_kind == 1 && b->h_status[redo[0]] == ST_BAND_OVERFLOW)
				fprintf(stderr, "[libmwf_hip] warning: wavefront of pair %d outgrew the whole-device kernel's span; re-running it on one workgroup (slow)\n", redo[0]);
		}
		else if (g->stats.kernel_kind == 1 && g->tb_budget_mb == 0 && g->coop_tb_cap < ((int64_t)1 << 40) && coop_can_grow(g)) {
			g->coop_tb_cap *= 2;
			for (int32_t i : redo)
				if (run_coop_pair(g, b, b->opt, i, false, false)) return -1;
			g->stats.n_retries += (int32_t)redo.size();
			if (fetch()) return -1;
			continue;
		} else if (g->stats.kernel_kind == 1 && b->opt.step > 0 && !coop_fell_back) {
			coop_fell_back = true;
			// whole-device low-memory run whose first-pass traceback does not fit: the generic kernel's true two-pass mode
			redo_kind = 0, slots = (int)redo.size(), band_overflow = true;
		} else if (slots == 1) {
			g->err = std::string(b->h_status[redo[0]] == ST_TB_OVERFLOW ? "traceback" : "low-memory snapshots") + " of pair " + std::to_string(redo[0]) +
			         " (tl=" + std::to_string(b->h_tl[redo[0]]) + ", ql=" + std::to_string(b->h_ql[redo[0]]) + ") do not fit in device memory" +
			         (b->opt.step > 0 ? "" : "; set opt.step > 0 (low-memory mode)");
			return -4;
		}
		if (!band_overflow) slots = std::max(1, std::min<int>(slots / 8, (int)redo.size()));
		std::stable_sort(redo.begin(), redo.end(), [&](int32_t x, int32_t y) {
			return (int64_t)b->h_tl[x] + b->h_ql[x] > (int64_t)b->h_tl[y] + b->h_ql[y];
		});
		int64_t max_len = 0, max_bound = 0, max_bound1 = 0;
		for (int32_t i : redo) {
			max_len = std::max<int64_t>(max_len, (int64_t)b->h_tl[i] + b->h_ql[i]);
			max_bound = std::max(max_bound, penalty_bound(b->opt, b->h_tl[i], b->h_ql[i], true));
			max_bound1 = std::max(max_bound1, penalty_bound(b->opt, b->h_tl[i], b->h_ql[i], false));
		}
		int32_t *d_redo = nullptr;
		HIP_TRY(g, hipMalloc(&d_redo, redo.size() * 4));
		HIP_TRY(g, hipMemcpy(d_redo, redo.data(), redo.size() * 4, hipMemcpyHostToDevice));
		g->stats.n_retries += (int32_t)redo.size();
		const int rc = run_batch_kernel(g, b, b->opt, d_redo, (int32_t)redo.size(), slots, max_len, max_bound, max_bound1, tb_budget_bytes(g), false, redo_kind);
		if (rc == 0 && fetch()) { (void)hipFree(d_redo); return -1; }
		(void)hipFree(d_redo);
		if (rc) return -1;
	}
	g->stats.cells = 0, g->stats.cells_pass1 = 0;
	for (size_t i = 0; i < n; ++i) g->stats.cells += b->h_iter[i], g->stats.cells_pass1 += b->h_cells1[i];
	b->finalized = true;
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

int32_t mwf_gpu_batch_cigar(mwf_gpu_t *g, mwf_gpu_batch_t *b, int32_t i, uint32_t *dst, int32_t cap)
{
	if (!g || !b || i < 0 || i >= b->n) return -1;
	(void)hipSetDevice(g->device);
	if (int rc = finalize(g, b)) return rc;
	const int32_t nc = b->h_ncig[i];
	if (nc > cap) return -5;
	if (nc > 0) HIP_TRY(g, hipMemcpy(dst, b->d_cig_pool + b->h_cigoff[i], (size_t)nc * 4, hipMemcpyDeviceToHost));
	return nc;
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

static mwf_gpu_t *thread_engine()
{
	// One engine (stream + pool) per host thread keeps the reference's re-entrancy: no shared
	// mutable state between threads beyond the HIP runtime itself.
	// (deliberately never destroyed: tearing a stream down from a thread_local destructor can run after
	// the HIP runtime's own static teardown)
	struct Holder { mwf_gpu_t *g = nullptr; };
	static thread_local Holder h;
	if (!h.g) {
		int dev = 0;
		if (const char *e = getenv("MWF_DEVICE")) dev = atoi(e);
		h.g = mwf_gpu_create(dev, nullptr);
		if (!h.g) fatal("cannot open a HIP device; this library has no CPU path", nullptr);
	}
	return h.g;
}

void mwf_wfa_batch(void *km, const mwf_opt_t *opt, int32_t n, const int32_t *tl, const char *const *ts,
                   const int32_t *ql, const char *const *qs, mwf_rst_t *r)
{
	mwf_gpu_t *g = thread_engine();
	if (n <= 0) return;
	std::vector<int64_t> t_off(n), q_off(n);
	int64_t total = 0;
	for (int32_t i = 0; i < n; ++i) {
		t_off[i] = total, total += tl[i];
		q_off[i] = total, total += ql[i];
	}
	std::vector<char> packed((size_t)total + 16, 0);
	for (int32_t i = 0; i < n; ++i) {
		if (tl[i]) memcpy(&packed[t_off[i]], ts[i], tl[i]);
		if (ql[i]) memcpy(&packed[q_off[i]], qs[i], ql[i]);
	}
	mwf_gpu_batch_t *b = mwf_gpu_batch_upload(g, n, packed.data(), total, t_off.data(), tl, q_off.data(), ql);
	if (!b) fatal("batch upload failed", mwf_gpu_last_error(g));
	if (mwf_gpu_batch_align(g, b, opt)) fatal("alignment failed", mwf_gpu_last_error(g));
	std::vector<int32_t> s(n), nc(n);
	std::vector<int64_t> it(n);
	if (mwf_gpu_batch_results(g, b, s.data(), it.data(), nc.data())) fatal("alignment failed", mwf_gpu_last_error(g));
	for (int32_t i = 0; i < n; ++i) {
		memset(&r[i], 0, sizeof(mwf_rst_t)); // reference miniwfa.c:387
		r[i].s = s[i], r[i].n_iter = it[i];
		if ((opt->flag & MWF_F_CIGAR) && s[i] >= 0) {
			r[i].n_cigar = nc[i];
			// reference krelocate()s the CIGAR into the caller's arena (miniwfa.c:434); a zero-length one stays NULL
			r[i].cigar = nc[i] > 0 ? (uint32_t*)kmalloc(km, (size_t)nc[i] * 4) : nullptr;
			if (nc[i] > 0 && mwf_gpu_batch_cigar(g, b, i, r[i].cigar, nc[i]) != nc[i]) fatal("CIGAR download failed", mwf_gpu_last_error(g));
		}
	}
	if (opt->flag & MWF_F_DEBUG) { // reference miniwfa.c:367 prints the traceback end state
		std::vector<int32_t> d4((size_t)n * 4);
		if ((opt->flag & MWF_F_CIGAR) && hipMemcpy(d4.data(), b->d_dbg4, (size_t)n * 16, hipMemcpyDeviceToHost) == hipSuccess)
			for (int32_t i = 0; i < n; ++i)
				if (s[i] >= 0) fprintf(stderr, "s0=%d, s=%d, i=%d, k=%d\n", s[i] - 1, d4[4 * i], d4[4 * i + 1], d4[4 * i + 2]);
	}
	mwf_gpu_batch_free(b);
}

void mwf_wfa_exact(void *km, const mwf_opt_t *opt, int32_t tl, const char *ts, int32_t ql, const char *qs, mwf_rst_t *r)
{
	mwf_wfa_batch(km, opt, 1, &tl, &ts, &ql, &qs, r);
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
