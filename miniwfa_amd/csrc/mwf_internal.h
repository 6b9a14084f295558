// mwf_internal.h — structures shared by the host side (mwf_engine.cpp / mwf_memory.cpp / mwf_plan.cpp) and the HIP kernels
// (mwf_kernels.hip).  Nothing here is part of the public ABI (include/miniwfa.h).
#pragma once
#include <stdint.h>

namespace mwf {

constexpr int32_t kNegInf = -0x40000000;   // reference miniwfa.c:67
constexpr int32_t kMaxRing = 256;          // ring slots the fast kernels' LDS tables hold: max(x, o1+e1, o2+e2) + 1 <= 256
constexpr int32_t kBigRing = 4096;         // ... and the generic kernel's big-ring form (any penalties with max(x, o1+e1, o2+e2) < 4096)

// per-pair status written by the kernels
enum : int32_t {
	ST_OK = 0,
	ST_STOPPED = 1,        // max_s / max_iter hit: s = -1 (reference miniwfa.c:422-428)
	ST_TB_OVERFLOW = 2,    // traceback arena of this slot too small: host re-runs the pair with a larger one
	ST_ROWS_OVERFLOW = 3,  // more penalties than row-metadata entries (cannot happen with the host's bound)
	ST_CIGAR_OVERFLOW = 4, // CIGAR pool exhausted: host grows it and re-runs
	ST_SNAP_OVERFLOW = 5,  // low-memory snapshot arena too small
	ST_INTERNAL = 6,       // an invariant the reference asserts on failed
	ST_PENDING = 7,        // not produced yet
	ST_BAND_OVERFLOW = 8,  // band kernel: window outgrew the register-resident span; host re-runs on the generic kernel
	ST_ALPHABET = 9        // band kernels with a 2-bit sequence copy: a base other than A/C/G/T; host re-runs the pair on the byte-wise copy
};

// Penalties in the form the recurrence uses them (reference miniwfa.c:252-256): lags into the ring.
struct Penalty {
	int32_t x;      // mismatch: H comes from the slice x penalties back
	int32_t oe1;    // o1+e1: gap-open source slice for piece 1
	int32_t e1;     // gap-extend source slice for piece 1
	int32_t oe2;    // o2+e2
	int32_t e2;
	int32_t o1, o2; // only traceback needs the opens separately
	int32_t nH;     // H ring depth  = max(x, oe1, oe2) + 1
	int32_t n1;     // E1/F1 depth   = e1 + 1
	int32_t n2;     // E2/F2 depth   = e2 + 1
};

// One workgroup ("slot") owns one of each of these regions; pairs are pulled from `queue`.
struct BatchArgs {
	// ---- input: packed sequences and per-pair geometry (device pointers)
	const uint8_t *seqs;
	const int64_t *t_off, *q_off;
	const int32_t *tl, *ql;
	const int32_t *order;      // optional processing order (longest first); may be null
	int32_t n_pairs;
	int32_t *queue;            // [0]: next position in `order`
	int32_t queue_parts;       // > 0: `queue` is a set of that many counters, 32 ints apart; counter c deals positions c, c + parts, ... to the workgroups with
	                           // blockIdx % parts == c (launches of thousands of short pairs: atomics on ONE address cost ~12.7 ns each)
	// ---- options
	Penalty pen;
	int32_t want_cigar;        // MWF_F_CIGAR
	int32_t step;              // low-memory checkpoint distance (0: high-memory)
	int32_t max_s;
	int64_t max_iter;
	int32_t debug_pair;        // pair whose per-penalty band is traced into dbg (or -1)
	// ---- per-slot workspace
	int32_t *ring;             // [slot][H:nH | E1:n1 | F1:n1 | E2:n2 | F2:n2][W]   offsets
	int32_t *sring;            // same shape, provenance (low-memory first pass only)
	int64_t ring_slot_ints;
	int32_t W;                 // row stride in ints; diagonal d of a pair lives at column d+tl+1
	unsigned long long *good;  // [slot][nH][GW] one bit per diagonal: some array holds an in-matrix offset
	int32_t GW;
	uint8_t *tb;               // [slot][tb_slot_bytes] traceback bytes, rows back to back
	int64_t tb_slot_bytes;
	int64_t *row_off;          // [slot][rows_slot] start of the row of penalty r+1 inside tb
	int32_t *row_lo;           // [slot][rows_slot] its first column
	int64_t rows_slot;
	uint32_t *cig_scratch;     // [slot][cig_scratch_slot] CIGAR is built backwards from the end
	int64_t cig_scratch_slot;
	uint32_t *cig_pool;        // all CIGARs of the batch, bump-allocated
	unsigned long long *cig_head;
	int64_t cig_pool_words;
	// ---- low-memory first pass
	int32_t *snap;             // [slot][snap_slot_ints] provenance snapshots, back to back
	int64_t snap_slot_ints;
	int32_t *snap_meta;        // [slot][snap_meta_slot] per snapshot: see mwf_kernels.hip
	int64_t snap_meta_slot;
	int32_t *seg;              // [slot][2*seg_slot] checkpoints (s, column)
	int64_t seg_slot;
	// ---- output, one entry per pair
	int32_t *out_s;
	int64_t *out_iter;
	int32_t *out_ncig;
	int64_t *out_cigoff;
	int32_t *out_status;
	int64_t *out_cells1;
	int32_t *out_dbg;          // [pair][4]: traceback end state (row, i, k) and last_state, for MWF_F_DEBUG
	int32_t *dbg;              // optional band trace of debug_pair: [2*dbg_cap] lo,hi per penalty (columns)
	int32_t dbg_cap;
	// ---- whole-device (cooperative) kernel only: cross-workgroup state, all accessed at agent scope
	int32_t coop_pair;         // the one pair this launch aligns
	int32_t lds_e2_cols;       // generic kernel, 512 threads, e2 == 1: columns of E2/F2 kept in LDS (power of two; 0 = all in HBM)
	int32_t ring16;            // generic kernel with E2/F2 in LDS: the ring rows in HBM hold 16-bit codes (half the traffic; offsets up to 65532)
	int32_t scalar_generic;    // generic kernel: 1 = one column per lane (forward_pass) also where the four-columns-per-lane pass applies
	int32_t band_lds_seq;      // band kernel with the sequences in LDS: bytes of the sequence copy (bookkeeping words and edge table sit behind it)
	int32_t band_fold;         // packed band kernel, score-only, o1 == x: the folded form (mwf_band2.hip: FOLD) — no loads of the row the first gap piece opens from
	int32_t coop_groups, coop_group_size; // pairs side by side on the whole-device kernel and workgroups per pair (grid = product)
	int64_t coop_edge_stride;  // ints between two groups' granule arrays
	int64_t coop_sedge_off;    // ints from a group's granule array to its array of provenance granules (true low-memory first pass)
	int64_t coop_misc_stride;  // bytes between two groups' flags / barrier words / pass state / flag ring
	const int32_t *coop_pair_ids; // [coop_groups] pair of every group (null: coop_pair)
	uint32_t coop_spin_limit;  // polls after which a wait for another workgroup gives up (ST_INTERNAL)
	// Re-runs without the host (round 5): a launch whose kernel hands pairs back (ST_BAND_OVERFLOW) appends their ids to retry_ids (at most retry_cap;
	// retry_count counts every one); a follow-up launch of a wider kernel in the same stream takes its pairs from that list — n_pairs_dev — instead
	// of the host reading the status words, re-launching and reading them again (~0.2 ms behind a 0.5 ms launch of 40 000 reads).
	int32_t *retry_ids;
	unsigned int *retry_count;
	int32_t retry_cap;
	const unsigned int *n_pairs_dev; // consumer: the pairs are order[0 .. min(*n_pairs_dev, n_pairs)), dealt blockIdx.x + k gridDim.x
	int32_t coop_pass;         // 0: plain pass; 1: pass whose traceback feeds the checkpoint walk; 2: second pass (uses seg); 3: first pass with provenance and snapshots
	int32_t *coop_edge;        // granules [nH][waves*2][2][4] x 8 B: E/F/H of every chunk's outer columns, tagged with their penalty
	int32_t *coop_flags;       // [12..14] origin offset and shrink reduction; [1024 + 4*(penalty mod 64) ..] edge-live / end-cell flag ring
	unsigned int *coop_sync;   // [0..1]: arrivals of the full barrier, [16+8g]: per-group arrivals, [96+8g]: per-group generation, [200..201]: workgroup-penalties finished
	int32_t *coop_state;       // results of a pass handed to the next launch: [0]=status [1]=s [2]=info [3]=n_seg [4..5]=cells
	// ---- systolic whole-device kernel (mwf_sys.hip): every chunk slot has a private H ring (`ring`: [group][slot][nH][256]) and
	// hands its outer columns to its two neighbours once per block of `sys_p` penalties
	int32_t sys_p;             // penalties per hand-off block (4, 8 or 16); a slot owns 64*sys_c - 2*sys_p columns, the rest is halo
	int32_t sys_c;             // columns per lane (4 or 1): a slot computes 64*sys_c columns
	int32_t sys_spread;        // 1: consecutive chunks on consecutive WORKGROUPS (a window of n chunks keeps n/grid waves per CU busy); 0: on consecutive waves of a workgroup
	int32_t *sys_box;          // [group][slot][2 parities][box ints]: outer columns' H rows of the block, E/F state at its end, window views
	int64_t sys_box_stride;    // ints between two groups' boxes
	unsigned long long *sys_prog; // [group][slot] x 64 bytes: blocks published
	int64_t sys_prog_stride;   // 8-byte words between two groups' progress words
	int32_t *sys_log;          // [group][2][rows_slot + 1]: wf_lo / wf_hi after every penalty, written by the owner of the edge column
	int64_t sys_log_stride;    // ints between two groups' logs
	int32_t *sys_park;         // [group][slot][E/F arrays][64 lanes][4]: registers of a wave's slot that is not the one it works on
	int64_t sys_park_stride;   // ints between two groups' parking areas
	int64_t *sys_ep;           // [group][epochs][2]: traceback layout per epoch of 256 penalties: base offset, first chunk | chunks << 32
	int64_t sys_ep_stride;     // int64 words between two groups' tables
	int32_t report_wide;       // packed band kernel, 512 threads x 4 chunk slots: 1 = note in the word behind cig_head whether three slots would have held every pair
	int32_t cig_block;         // > 0: workgroups take the CIGAR pool in blocks of this many words and place their pairs' CIGARs in them (batches of thousands of pairs:
	                           // one atomic on cig_head per pair is ~12.7 ns on a single address); 0: one allocation per pair, no holes
	int32_t lane_chunks;       // one-diagonal-per-lane kernels: 64-column chunks of their LDS rows (mwf_lane.hip: 1-4; mwf_mid.hip: its span / 64)
	int32_t sys_coop_launch;   // host side only: 1 = launch through hipLaunchCooperativeKernel (the runtime then guarantees that every workgroup is resident)
};

// launch wrappers implemented in mwf_kernels.hip (generic kernel: any penalties, any band, low-memory mode)
int launch_reset(int32_t *status, int32_t *s, int32_t n, unsigned long long *cig_head, int32_t *queue, int32_t n_queue, void *stream);
int launch_batch(const BatchArgs &a, int grid, int block, void *stream);
// 8-mer sketch of `samples` (<= 16) pairs of a device-resident batch: out[k] = 8-mers of pair (k n / samples)'s query prefix that occur in its target prefix
int launch_sketch(const uint8_t *seqs, const int64_t *t_off, const int32_t *tl, const int64_t *q_off, const int32_t *ql, int32_t n, int32_t samples, int32_t *out, void *stream);
int  bigring_kernel_occupancy();                      // ... of the big-ring form (penalty sets with max(x, o1+e1, o2+e2) >= 256)
int  batch_kernel_occupancy(int block, bool stream_pass, int lds_e2_cols, bool ring16);   // resident workgroups per CU for that block size

// geometry of a launch of the band family (mwf_band2.hip packed band kernel, mwf_lane.hip, mwf_mid.hip)
struct BandGeom {
	int block;        // threads per workgroup: 256 (x2 chunks), 768 (x2 chunks) or 512 (x3 chunks, packed state)
	int packed;       // 1: the packed band kernel (mwf_band2.hip: E/F register state as int16 pairs, 16-bit H rows); 2: its 512 x 4 geometry's copy on biased offsets
	int span;         // columns the workgroup can hold: waves * chunks * 256 (balanced kernel: columns of its LDS state ring)
	int lds_bytes;    // dynamic LDS for the sequence copy (0: read sequences from global memory)
	int seq2;         // packed kernel: the sequence copy holds 2 bits per base (pairs of plain A/C/G/T; others come back as ST_ALPHABET)
	int lane;         // 1: the one-wave, one-diagonal-per-lane kernel for short pairs (mwf_lane.hip): block 64, span 64, lds_bytes = rings + sequences;
	                  // 2: the one-workgroup, one-diagonal-per-lane kernel for a few mid-size pairs (mwf_mid.hip): span = columns of its LDS rows
};
// one pair (or a few) across the whole device: mwf_sys.hip (the per-penalty hand-off kernel of rounds 1-2, mwf_coop.hip, was removed in round 5)
bool coop_supported(const Penalty &p);
int64_t coop_chunk_slots(int grid);                  // 256-column chunks a launch of `grid` workgroups holds (window capacity + 1)                      // co-resident workgroups the kernel may be launched with
// launch wrappers implemented in mwf_sys.hip (the systolic whole-device kernel: every pass, the provenance pass of the two-pass low-memory mode included)
int64_t sys_chunk_slots(int grid);                   // chunk slots a launch of `grid` workgroups holds
int  sys_owned_cols(int p, int c);                   // columns a chunk slot owns: 64c - 2p
bool sys_c_supported(int c);                         // columns per lane the build has kernels for (1 and 4; 2 only with -DMWF_SYS_C2)
bool sys_p_supported(int p);                         // block lengths (penalties per hand-off) the build has kernels for
int64_t sys_box_ints(int p, bool seg = false);                         // ints of one hand-off box
int  sys_max_grid();                                 // co-resident workgroups the kernel may be launched with (one per CU)
int  launch_sys_pass(const BatchArgs &a, int grid, void *stream);        // forward pass (score / traceback bytes / second pass with band resets)
int  launch_sys_walk(const BatchArgs &a, void *stream);
int  launch_sys_trace(const BatchArgs &a, void *stream);                 // checkpoints from the snapshots of the systolic kernel's provenance pass                  // checkpoints from the traceback matrix of a first pass
int  launch_sys_finish(const BatchArgs &a, void *stream);                // traceback + per-pair outputs

// launch wrappers implemented in mwf_lane.hip (one wave per pair, one diagonal per lane, rings and sequences in LDS: short pairs)
bool lane_supported(const Penalty &p);
int  lane_lds_bytes(const Penalty &p, int chunks, int64_t seq_bytes); // seq_bytes >= tl + ql + 24 for every pair of the launch
int  launch_lane(const BatchArgs &a, int grid, int lds, bool seq2, void *stream); // seq2: 2-bit sequence copies (a pair outside plain A/C/G/T comes back as ST_ALPHABET)
int  lane_kernel_occupancy(int lds, bool cigar);

// launch wrappers implemented in mwf_mid.hip (one workgroup per pair, one diagonal per lane, rings and sequences in LDS: a few mid-size pairs)
bool mid_supported(const Penalty &p);
int  mid_lds_bytes(const Penalty &p, int groups, int64_t seq_bytes); // seq_bytes >= (tl up to 8) + 16 + (ql up to 8) + 32 for every pair of the launch
int  launch_mid(const BatchArgs &a, int grid, int block, int lds, bool seq2, void *stream); // seq2: 2-bit sequence copies (a pair outside plain A/C/G/T comes back as ST_ALPHABET)

// launch wrappers implemented in mwf_band2.hip (packed band kernel: 16-bit offsets, sequences in LDS; BandGeom::packed)
bool band2_supported(const Penalty &p);
int  launch_band2(const BatchArgs &a, int grid, const BandGeom &g, void *stream);
int  band2_kernel_occupancy(const Penalty &p, const BandGeom &g, bool cigar);
bool band2_biased512_supported(const Penalty &p);  // its five / six-slot copies on biased offsets exist for gap extensions (2, 1) only
int  band2_biased512_chunks();                       // ... and the 512-thread geometry's copy on biased offsets (8 waves x slots per wave)
int  band2_span_chunks();                            // 256-column chunks its 1024-thread geometry holds (16 waves x slots per wave)

} // namespace mwf
