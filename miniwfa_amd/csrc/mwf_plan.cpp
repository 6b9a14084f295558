// mwf_plan.cpp — the plan / launch / retry layer of libmwf_hip.so's host side: penalties, which kernel and geometry serves a set of pairs
// (choose_kernel), the size classes of a batch (mwf_gpu_batch_align), one launch of the one-workgroup-per-pair kernels (run_batch_kernel), the
// passes of the whole-device kernel (run_coop_group) and the re-runs of what did not fit where it ran (finalize).  Split from mwf_engine.cpp in
// round 5 (no behaviour change).
#include "mwf_engine.h"

namespace mwf {
namespace host {

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
			// (window_hint: the widest window the class's pairs are expected to reach — a pair of very different lengths opens a gap its length does not tell)
			const int64_t by_len = window_hint > 0 ? std::max<int64_t>(max_len, window_hint * 100 / 28) : max_len;
			const int block = g->mid_block ? g->mid_block : (by_len <= 1000 ? 256 : by_len <= 2500 ? 512 : 1024); // (4 x 400 bp: 131 us on four waves, 145 on eight)
			const int seq2 = g->seq2bit != 0 && !g->acgt_off_once;
			pl.kind = 2, pl.band = BandGeom{block, 1, 64 * groups, lds, seq2, 2};
			return;
		}
	}
	if (want_kind == 0 || low_mem || !can_packed) return;
	if (geom_block == 514) { // the 512-thread geometry with four chunk slots on biased offsets (the caller checked the lengths: kBandSpanMaxSeq); 2-bit copies only
		const int64_t need_lds = ((max_len >> 4) + 4) * 4;
		if (g->seq2bit == 0 || g->acgt_off_once || need_lds > 70 * 1024 || !band2_biased512_supported(P)) return;
		// (five chunk slots per wave while target + query stay below 3.5 of that span, else six)
		// (... or when the class was admitted on a forecast window — divergence known, window_hint = the class's largest — that five slots cannot hold: the
		// admission test is against the SIX-slot window, a 17 kb pair at 10 % must not start on five and overflow)
		const bool six = max_len + 1 > 7 * (int64_t)(band2_biased512_chunks() * 256) / 2 || (window_hint > 0 && window_hint + 768 > ((int64_t)band2_biased512_chunks() - 1) * 256 - 64);
		const int chunks = band2_biased512_chunks() + (six ? 8 : 0);
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
	// Round 6: the byte-wise copy exists in ONE geometry, 768 x 2 (24 chunks like 512 x 3, one workgroup per CU) — whatever the window class or a forced block:
	// pairs outside plain A/C/G/T are rare (reads with an N take the lane kernel's byte-wise copy), and a byte-wise twin of every geometry was 40 % of this
	// kernel family's code.  The 768-thread geometry in turn has no 2-bit form.
	if (bg.block == 768) bg.seq2 = 0, bg.lds_bytes = max_seq_lds <= 140 * 1024 ? (int)((max_seq_lds + 15) / 16 * 16) : 0;
	else if (!bg.seq2) {
		bg.block = 768, bg.span = 768 / 64 * 2 * 256;
		bg.lds_bytes = max_seq_lds <= 140 * 1024 ? (int)((max_seq_lds + 15) / 16 * 16) : 0;
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
		per_cu = cached_occupancy(g, P, pl, 0, false);
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
			pl.block = ring16 ? (pl.cigar ? 768 : 512) : 768;
		}
		if (P.nH > kMaxRing) pl.block = 256; // the big-ring form of the generic kernel (launch_batch): one column per lane, 256 threads
		per_cu = cached_occupancy(g, P, pl, lds_e2_cols, !g->scalar_generic, ring16);
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
	// re-runs without the host (BatchArgs::retry_ids): this launch fills the batch's list / takes its pairs from it
	if (g->retry_mode == 1) a.retry_ids = b->d_retry_ids + g->retry_slot * kRetryCap, a.retry_cap = kRetryCap, a.retry_count = (unsigned int*)(b->d_cig_head + 2 + g->retry_slot);
	if (g->retry_mode == 2) a.n_pairs_dev = (const unsigned int*)(b->d_cig_head + 2 + g->retry_slot), a.queue = nullptr, a.queue_parts = 0;
	a.scalar_generic = g->scalar_generic;
	a.band_fold = g->band_fold ? 1 : 0;
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
	const int sysP = g->sys_p, sysP2 = g->sys_p;
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
		as.sys_coop_launch = 1;
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

} // namespace host
} // namespace mwf

extern "C" {

int mwf_gpu_batch_align(mwf_gpu_t *g, mwf_gpu_batch_t *b, const mwf_opt_t *opt)
{
	if (!g || !b || !opt) return -1;
	if (const char *why = validate(*opt)) { g->err = why; return -2; }
	(void)hipSetDevice(g->device);
	b->opt = *opt;
	b->aligned = false, b->finalized = false, b->h_cig_valid = false;
	b->last_grid = 0, b->n_retries = 0, b->dev_retry_used = false;
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
		// (low-memory mode: the alternative is the generic kernel's two-pass form, three times its high-memory time — 17 x 30 kb @ 5 % 186 ms against 50 ms per round of
		// sixteen here, 48 x 100 kb @ 3 % 694 ms against 555 ms in six rounds of eight: at least four rounds, where the rule above allowed one round of 30 kb pairs;
		// profiles/r06/routing_survey_long_before.txt, _after.txt)
		const bool lowmem_call = (opt->flag & MWF_F_CIGAR) && opt->step > 0;
		const int64_t coop_rounds = lowmem_call ? std::max<int64_t>(4, max_len * 32 / 1000000) : std::max<int64_t>(1, max_len * 27 / 1000000);
		int64_t coop_max_pairs = std::max<int64_t>(1, std::min<int64_t>(256, coop_rounds * coop_side_by_side));
		if (max_len < 65536) coop_max_pairs = std::min<int64_t>(coop_max_pairs, lowmem_call ? 48 : 16); // (measured up to sixteen)
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
	// Penalties the packed band kernel is not instantiated for (gap extensions other than (2,1), (2,2), (1,1): minimap2's asm5 has 3 and 1) leave the generic kernel for
	// everything above — but the lane and mid kernels read their penalties at run time: read batches and a few mid-size pairs still get them (20 000 x 150 bp with
	// e1 = 3: 4.8 ms on the generic kernel, a thirteenth of that on the lane kernel).  What outgrows them goes to the generic kernel.
	const bool small_only = g->force_kind < 0 && g->block == 0 && !band2_supported(P0) && g->band_pack != 0;
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
		// the mid kernel (a workgroup per pair, one per CU) serves a FEW pairs: a batch of at most mid_cap pairs, or the at most mid_cap pairs of a
		// larger batch that are neither short reads (lane kernel) nor long (classes 0-2 ... see below) — mwf_wfa_chain's gap fills are hundreds of
		// tiny pairs and a handful of longer ones, and the handful sets the call's time
		const bool mid_batch = g->force_kind < 0 && g->block == 0 && mid_cap > 0 && mid_supported(P0);
		const bool mid_ok = mid_batch && b->n <= mid_cap;
		const bool know_acgt = !b->h_acgt.empty() && g->seq2bit != 0;
		const bool pack_pen = g->band_pack != 0 && band2_supported(P0);
		std::vector<int8_t> cls((size_t)b->n); // group of every pair
		int32_t count[15] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
		const bool span_pen = pack_pen && g->band_span != 0 && g->seq2bit != 0;
		// The classes' length limits stand for "the window stays inside the span at ~5 % divergence".  Where the batch's own divergence is known
		// (estimate_divergence: batches built from host memory) the lengths are weighed by it: three times as diverged = as if three times as long.
		// Only upwards of the prior, and a little downwards: a class too narrow costs a second run, one too wide a few per cent.
		const double div_r = b->div_est > 0 && g->div_aware ? std::min(8.0, std::max(0.7, (double)b->div_est / 0.05)) : 1.0;
		const double div_lane = b->div_est > 0 && g->div_aware ? std::min(8.0, std::max(0.45, (double)b->div_est / 0.05)) : 1.0;
		// A batch of reads whose lengths straddle the limit is not split for a few pairs: where most of the pairs up to 10 % beyond the limit lie below it, the rest follow
		// them (2000 x 300 bp with 28 pairs beyond: 0.37 ms as one lane launch, 0.50 with a launch of their own)
		int64_t lane_limit = mid_ok ? std::min(g->lane_max_len, 320) : g->lane_max_len;
		if (lane_ok && !mid_ok && (classes || small_only)) {
			int64_t n_lo = 0, n_hi = 0;
			for (int32_t i = 0; i < b->n; ++i) {
				const int64_t w = (int64_t)((double)std::max(b->h_tl[i], b->h_ql[i]) * div_lane);
				if (std::abs(b->h_tl[i] - b->h_ql[i]) > 24) continue;
				n_lo += w <= lane_limit, n_hi += w > lane_limit && w <= lane_limit * 11 / 10;
			}
			if (n_hi > 0 && n_lo >= n_hi) lane_limit = lane_limit * 11 / 10;
		}
		struct PairInfo { int64_t bound, bound1, exp_win; };
		std::vector<PairInfo> info((size_t)b->n);
		std::vector<int8_t> mid_cand;   // large batches: pairs the mid kernel would take if they are few
		int32_t n_cand = 0;
		if (mid_batch && !mid_ok) mid_cand.assign((size_t)b->n, 0);
		// class of pair i; allow_mid: the mid kernel may take it
		auto classify = [&](int32_t i, bool allow_mid) -> int {
			const int64_t tl = b->h_tl[i], ql = b->h_ql[i], len = tl + ql;
			// A pair of very different lengths must open a gap of |tl - ql|: its window reaches that diagonal whatever its divergence (two columns per
			// penalty of the gap, until the matrix ends), so the length limits see it as if it were longer — beyond what indels of a related pair add up
			// to.  (Found on mwf_wfa_chain's gap fills: a 54 x 740 fill sat in the 64-thread class and was run twice, every call.)
			const int64_t skew = std::abs(tl - ql), skew_len = 6 * std::max<int64_t>(0, skew - len / 16);
			const int64_t lenw = (int64_t)((double)len * div_r) + skew_len; // the pair's length as the classes' limits should see it
			// the window the pair is expected to reach (0.27 (tl+ql) at 5 %), plus 15 %, where the divergence is known: the long classes' length limits
			// were drawn for ~3 % (configs[4]) and sent 20 kb pairs at 15 % and 50 kb pairs at 5 % through the span geometry for nothing
			const int64_t exp_win = b->div_est > 0 && g->div_aware ? std::min<int64_t>(len + 1, (int64_t)(6.2 * b->div_est * (double)len) + 64) : 0;
			const int64_t bound1 = penalty_bound(*opt, tl, ql, false);
			const int64_t bound = opt->max_s > 0 ? std::min<int64_t>(bound1, (int64_t)opt->max_s + 1) : bound1; // (= penalty_bound(..., true))
			info[(size_t)i] = PairInfo{bound, bound1, exp_win};
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
				// (... at 5 %.  A batch the sketch puts at 8 % and more outgrows the 64-thread class earlier than its weight says — the sketch reads low up there, and a pair
				// whose penalty stays below 256 is never shrunk, its window is 2 s + 1: 2000 x 350 bp @ 10 % lost 876 pairs, 20 000 x 380 bp 18 473, to a second run)
				else if (packable && (window <= kBandMicroWindow || lenw + 1 <= (b->div_est >= 0.06f && g->div_aware ? 1000 : 1400))) c = 4;
				else if (packable && (window <= kBandTinyWindow || lenw + 1 <= 3600)) c = 3;
				else if (packable && (window <= kBandSmallWindow || lenw + 1 <= 8200)) c = 2;
				else if (packable && (lenw + 1 <= 4 * (int64_t)(8 * 3 * 256) || window <= kBandWideWindow)) c = 1;
				// (tl + ql up to 3.5 of its spans: a 12 kb pair at 5 % needs ~6000 of the 7872 columns; 512 x 15 kb @ 5 % — windows of ~7500 — lost 44 pairs to late
				// overflows, 30.7 against 24.9 ms on the span geometry from the start)
				else if (span_ok && band2_biased512_supported(P0) && g->wide_slots != 3 && (exp_win ? exp_win + 768 <= ((int64_t)band2_biased512_chunks() + 8 - 1) * 256 - 64 : len + 1 <= 7 * (int64_t)((band2_biased512_chunks() + 8) * 256) / 2)) c = 14;
				else if (span_ok && ((exp_win ? exp_win + 768 <= band_span_window() : len + 1 <= 7 * band2_span_chunks() * 256) || window <= band_span_window())) c = 13;
			}
			b->h_class[i] = (int8_t)(c == 5 ? 0 : c == 13 ? 5 : c == 14 ? 1 : c);
			b->h_flags[i] = (int8_t)(step0 ? 1 : 0);
			// the class the pair's lengths would give it, for the admission to the lane and mid kernels where the band kernel itself cannot serve (their offsets are 16 bits too)
			int c_adm = c;
			if (small_only && c == 0 && tl + bound < 32767) {
				const int64_t window = std::min<int64_t>(len + 1, 2 * bound + 3);
				c_adm = (window <= kBandMicroWindow || lenw + 1 <= (b->div_est >= 0.06f && g->div_aware ? 1000 : 1400)) ? 4 : (window <= kBandTinyWindow || lenw + 1 <= 3600) ? 3 : 0;
			}
			// short pairs: a window of 64 diagonals holds them while the penalty stays below ~45 (a 200 bp pair at 5 %)
			// (in a batch small enough for the mid kernel the lane kernel keeps the pairs of up to 320 bases: 16 x 400 bp 0.31 ms on the lane
			// kernel — pairs that outgrow its chunks are re-run — against 0.13 on the mid kernel, 1 x 300 bp 56 against 68 us; profiles/r04/lane_vs_mid.txt)
			// (the lane kernel's limit is a window — its chunks — and a window is proportional to divergence x length: the weight goes further down for it than for the
			// classes, 2000 x 450 bp @ 2 % 0.14 against 0.27 ms on the 64-thread geometry; profiles/r06/lane_crossover.txt)
			const bool to_lane = (classes || small_only) && lane_ok && c_adm >= 1 && c_adm <= 4 && (int64_t)((double)std::max(tl, ql) * div_lane) <= lane_limit && skew <= 24;
			if (to_lane) c = (know_acgt && !b->h_acgt[i]) ? 12 : 10, b->h_class[i] = (int8_t)(classes ? 4 : 2); // (where an overflow goes: the re-run of "band" pairs takes the mid kernel for a handful, else whatever choose_kernel has for the penalties)
			// a few mid-size pairs: a workgroup each, rings in LDS (a penalty then costs a fraction of what it costs the band kernels).  Admitted
			// when the span the LDS can hold beside the sequences covers the window of a pair at ~6 % divergence (about 0.3 (tl+ql)) and the gap its lengths
			// force — or every column the pair can ever reach; 16-bit offsets.
			if (mid_batch && !to_lane && c <= 4 && (!low_mem || step0) && tl + bound < 32760) {
				const int64_t seq_lds = ((tl + 7) & ~7LL) + 16 + ((ql + 7) & ~7LL) + 32;
				const int64_t window = std::min<int64_t>(len + 1, 2 * bound + 3);
				const int64_t want = std::min<int64_t>(window, lenw * 34 / 100 + 128) + 2 * P0.nH;
				int groups = (int)std::min<int64_t>((window + 2 * P0.nH + 63) / 64, 128);
				while (groups > 1 && mid_lds_bytes(P0, groups, seq_lds) > 158 * 1024) --groups;
				// (the span lies around the middle of diagonals 0 and ql - tl, mwf_mid.hip: all the columns a window can reach — the matrix, or what the penalty
				// bound allows either side of diagonal 0 — plus the dead margins)
				const int64_t C = (int64_t)groups * 64, left = tl + 1 + (ql - tl) / 2 - C / 2, right = left + C - 1;
				const bool holds_all = std::max<int64_t>(1, tl + 1 - (bound + 1)) - P0.nH >= left && std::min<int64_t>(len + 1, tl + 1 + bound + 1) + P0.nH <= right;
				if (mid_lds_bytes(P0, groups, seq_lds) <= 158 * 1024 && (holds_all || (C >= want && skew < groups * 32))) {
					if (!allow_mid) {
						// (a large batch: counted; the pairs move to the mid kernel afterwards if they are few, and only the band kernels' SMALL classes give
						// pairs away — wide windows are the 512-thread geometry's work whatever their number)
						if (!mid_cand.empty() && c_adm >= 3 && c_adm <= 4 && !mid_cand[(size_t)i]) mid_cand[(size_t)i] = 1, ++n_cand;
					} else {
						b->h_class[i] = (int8_t)((c >= 1 && c <= 4) || (classes && packable) ? 2 : 0); // where an overflow goes: the wide packed band kernel, else generic
						c = 11;
						mid_bytes |= know_acgt && !b->h_acgt[i]; // a pair the host knows not to be plain A/C/G/T: the (few) pairs of this class all take the byte-wise copy
					}
				}
			}
			if (c >= 1 && c <= 4 && know_acgt && !b->h_acgt[i] && packable) c += 5;
			return c;
		};
		for (int32_t i = 0; i < b->n; ++i) cls[(size_t)i] = (int8_t)classify(i, mid_ok);
		if (getenv("MWF_DEBUG")) fprintf(stderr, "[libmwf_hip] classes of %d pairs: divergence estimate %.4f (weight %.2f)\n", b->n, (double)b->div_est, div_r);
		if (n_cand > 0 && n_cand <= mid_cap)
			for (int32_t i = 0; i < b->n; ++i)
				if (mid_cand[(size_t)i]) cls[(size_t)i] = (int8_t)classify(i, true);
		for (int32_t i = 0; i < b->n; ++i) {
			const int64_t tl = b->h_tl[i], ql = b->h_ql[i], len = tl + ql;
			const int c = cls[(size_t)i];
			const PairInfo &pi = info[(size_t)i];
			GroupInfo &G = gi[c];
			++count[c];
			G.max_len = std::max(G.max_len, len), G.max_bound = std::max(G.max_bound, pi.bound);
			G.max_bound1 = std::max(G.max_bound1, pi.bound1);
			// (the mid class: the window its pairs are expected to reach, forced gap included — its launch picks the workgroup size by it)
			const int64_t exp_win = c == 11 ? std::min<int64_t>(len + 1, (int64_t)((double)len * div_r * 0.28) + 2 * std::abs(tl - ql)) : pi.exp_win;
			G.max_tl = std::max<int64_t>(G.max_tl, tl), G.max_exp_win = std::max(G.max_exp_win, exp_win);
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
		// (the lane class of a batch of reads hands its overflows to a follow-up launch on the device, below)
		const bool lane_retry = (c == 10 || c == 12) && g->dev_retry && G.n >= 1024 && !preset && g->force_kind < 0 && g->block == 0 && mid_supported(P0) && G.max_len <= 1200 &&
		                        G.max_tl + G.max_bound < 32760 && !low_mem;
		if (lane_retry) g->retry_mode = 1, g->retry_slot = c == 12 ? 1 : 0;
		g->acgt_off_once = (c > 5 && c < 10) || (c == 11 && mid_bytes) || c == 12;
		const int rc = run_batch_kernel(g, b, c == 5 ? *opt : opt_hi, b->d_order + at, G.n, slots, G.max_len, G.max_bound, G.max_bound1,
		                                done_groups == 1, (classes || c >= 11) ? (c == 0 || c == 5 ? 0 : 2) : -1, G.max_tl, G.max_seq_lds, lane_retry ? 0 : done_groups == n_groups,
		                                cc == 8 ? 514 : cc == 7 ? 1024 : cc == 6 ? 33 : cc == 5 ? 32 : cc == 4 ? 64 : cc == 3 ? 128 : cc == 2 ? 256 : 0, &ran,
		                                (c == 1 && (g->wide_slots == 4 || (g->wide_slots == 0 && PC.wide_state != 1 && g->queue_clean))) ? kBandWide4Window : (c == 14 || c == 11) ? G.max_exp_win : 0);
		g->retry_mode = 0;
		if (c == 1 && g->wide_slots == 0 && PC.wide_state == 0 && g->queue_clean && ran == 2 && g->stats.block == 512) PC.wide_measured = true;
		// Batches of reads: what the lane kernel hands back (a window that left its chunks: one read in tens of thousands) is re-run by a follow-up launch of
		// the mid kernel, whose span holds the widest window such a pair can have — from a list the lane kernel filled ON THE DEVICE.  Round 4 read the status
		// words back, launched, waited and read them again: ~0.2 ms behind a 0.5 ms launch.  (An empty list costs the launch of a few idle workgroups.)
		if (rc == 0 && lane_retry) {
			const mwf_gpu_stats_t keep = g->stats;
			g->retry_mode = 2;
			g->acgt_off_once = c == 12;
			int ran2 = 0;
			// (every entry of the list — the mid kernel strides over it by its grid —, on at most 32 workgroups: an empty list costs the launch of a few idle ones)
			const int rc2 = run_batch_kernel(g, b, opt_hi, b->d_retry_ids + g->retry_slot * kRetryCap, kRetryCap, std::min(slots, 32), G.max_len, G.max_bound, G.max_bound1, false, 2, G.max_tl, G.max_seq_lds,
			                                 done_groups == n_groups, 33, &ran2, 0);
			g->retry_mode = 0, g->acgt_off_once = false;
			const int32_t launches = g->stats.n_launches;
			g->stats = keep, g->stats.n_launches = launches; // (the statistics describe the class's own launch)
			if (rc2) return -1;
			b->dev_retry_used = true;
		}
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

namespace mwf {
namespace host {


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
	if (b->dev_retry_used) { // pairs a follow-up launch re-ran from the device-side list count as re-runs too (the third word of the head: BatchArgs::retry_count)
		for (int slot = 0; slot < kRetrySlots; ++slot) { // (what a list could not hold stayed ST_BAND_OVERFLOW and is counted by the host path below)
			uint32_t k = 0;
			memcpy(&k, b->host_out.data() + 16 + 8 * slot, 4);
			b->n_retries += (int32_t)std::min<uint32_t>(k, (uint32_t)kRetryCap);
		}
	}

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

} // namespace host
} // namespace mwf
