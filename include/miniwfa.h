/*
 * miniwfa.h — C ABI of libmwf_hip.so, the MI355X (gfx950) implementation of miniwfa's exact
 * score-and-CIGAR path.
 *
 * PART 1 is the drop-in surface: the option/result structs and entry points of lh3/miniwfa
 * (reference miniwfa.h:32-89) with identical names, field order, sizes (56 B / 24 B), argument
 * meaning, ownership and error behaviour, so a program written against the reference header
 * links against this library unchanged.  Each declaration cites the reference line it replaces.
 *
 * PART 2 is new: a batch entry point over host buffers and a device-resident engine API
 * (plain pointers and sizes only) that the Python/torch host side and bench.py drive.
 *
 * There is no CPU fallback anywhere behind this header: every alignment runs in the HIP
 * kernels under miniwfa_amd/csrc/ (mwf_kernels.hip generic, mwf_band2.hip packed band,
 * mwf_lane.hip short pairs, mwf_mid.hip the mid-size pairs of small batches, mwf_sys.hip whole device), and every entry point aborts with a message
 * if no gfx950 device can be opened.
 */
#ifndef MWF_HIP_MINIWFA_H
#define MWF_HIP_MINIWFA_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------
 * PART 1 — drop-in surface
 * ---------------------------------------------------------------------------------------- */

/* flag bits, reference miniwfa.h:32-34 */
#define MWF_F_CIGAR      0x1      /* produce the CIGAR, not only the penalty */
#define MWF_F_NO_KALLOC  0x2      /* reference: bypass kalloc for scratch; here: accepted, no effect (scratch is device memory) */
#define MWF_F_DEBUG      0x10000  /* one line on stderr at the end of traceback (reference miniwfa.c:367) */

/* reference miniwfa.h:36-44; offsets flag@0 x@4 o1@8 e1@12 o2@16 e2@20 step@24 max_s@28 max_iter@32 max_occ@40 kmer@44 min_len@48 */
typedef struct {
	int32_t flag;
	int32_t x, o1, e1, o2, e2; /* mismatch; gap open/extend of the two affine pieces: a gap of length L costs min(o1+L*e1, o2+L*e2) */
	int32_t step;              /* >0: low-memory two-pass mode, checkpoint every `step` penalties */
	int32_t max_s;             /* >0: give up (s=-1) once the penalty exceeds this */
	int64_t max_iter;          /* >0: give up (s=-1) once more than this many (penalty,diagonal) cells were computed */
	int32_t max_occ, kmer, min_len; /* chaining heuristic only (mwf_wfa_chain) */
} mwf_opt_t;

/* reference miniwfa.h:46-51; offsets s@0 n_cigar@4 n_iter@8 cigar@16 */
typedef struct {
	int32_t s;        /* alignment penalty; -1 if stopped by max_s / max_iter */
	int32_t n_cigar;
	int64_t n_iter;   /* cells computed by the (second-pass) core loop, reference miniwfa.c:421 */
	uint32_t *cigar;  /* len<<4|op, op in {1:I, 2:D, 7:=, 8:X}; allocated from the caller's km (kfree(km,.) or free() if km==NULL) */
} mwf_rst_t;

/* reference miniwfa.h:62 / miniwfa.c:11-18: x=4 o1=4 e1=2 o2=15 e2=1 kmer=13 max_occ=2 min_len=30, rest 0 */
void mwf_opt_init(mwf_opt_t *opt);

/* reference miniwfa.h:83 / miniwfa.c:603-615: optimal global alignment of ts[0,tl) vs qs[0,ql).
 * Sequences are length-delimited arbitrary bytes compared verbatim.  *r is fully overwritten.
 * Limits of this implementation (the reference has neither; both end in a message and abort(), like the reference's
 * own assert/panic paths): max(x, o1+e1, o2+e2) < 4096 (below 256 every kernel applies; from 256 on — gap-open costs in the
 * hundreds — the pairs run on the generic kernel's big-ring form, one column per lane), and tl+ql < 2^31-4 (columns
 * are 32-bit).  x, e1, e2 >= 1 and o1, o2 >= 0, as the reference requires implicitly (miniwfa.c:390-392).
 * Device: MWF_DEVICE=<ordinal> (default 0).  Any number of host threads may call concurrently. */
void mwf_wfa_exact(void *km, const mwf_opt_t *opt, int32_t tl, const char *ts, int32_t ql, const char *qs, mwf_rst_t *r);

/* reference miniwfa.h:85 / miniwfa.c:898-908: exact with step=0 and max_iter=1e8; if that stops, mwf_wfa_chain. */
void mwf_wfa_auto(void *km, const mwf_opt_t *opt, int32_t tl, const char *ts, int32_t ql, const char *qs, mwf_rst_t *r);

/* reference miniwfa.h:84 / miniwfa.c:850-896: k-mer chaining heuristic whose gap fills are exact alignments.
 * The chaining itself is host code; every gap fill runs on the device as one batch. */
void mwf_wfa_chain(void *km, const mwf_opt_t *opt, int32_t tl, const char *ts, int32_t ql, const char *qs, mwf_rst_t *r);

/* reference miniwfa.h:88-89 / mwf-dbg.c:6-31 */
int32_t mwf_cigar2score(const mwf_opt_t *opt, int32_t n_cigar, const uint32_t *cigar, int32_t *tl, int32_t *ql);
void mwf_assert_cigar(const mwf_opt_t *opt, int32_t n_cigar, const uint32_t *cigar, int32_t tl0, int32_t ql0, int32_t s0);

/* ------------------------------------------------------------------------------------------
 * PART 2 — batch and device-resident API (new; SURVEY.md §8b "New")
 * ---------------------------------------------------------------------------------------- */

/* n independent pairs from host buffers; r[i] is filled exactly as mwf_wfa_exact would fill it.  Replaces the serial
 * loop over pairs of reference main.c:67-72.  Runs on the device MWF_DEVICE names, or — when the environment sets
 * MWF_DEVICES to "all" or a count — dealt over that many devices as mwf_wfa_batch_multi does. */
void mwf_wfa_batch(void *km, const mwf_opt_t *opt, int32_t n, const int32_t *tl, const char *const *ts,
                   const int32_t *ql, const char *const *qs, mwf_rst_t *r);

/* The same over several devices of one node: pairs are dealt longest first to the device with the least
 * work so far, every device runs its share on a host thread with its own engine, nothing crosses devices while they
 * work, and the results are merged back in the caller's order.  devices: n_dev HIP ordinals (an ordinal may repeat),
 * or NULL for ordinals 0..n_dev-1 (n_dev <= 0: every visible device).  CIGARs are allocated from `km` on the calling
 * thread only. */
void mwf_wfa_batch_multi(void *km, const mwf_opt_t *opt, int32_t n, const int32_t *tl, const char *const *ts,
                         const int32_t *ql, const char *const *qs, mwf_rst_t *r, int32_t n_dev, const int32_t *devices);

/* n pairs in chain mode: r[i] is what mwf_wfa_chain would give for pair i (reference miniwfa.c:850-896 once per record, main.c:67-72).  The k-mer chaining of
 * the pairs runs on a few host threads, the gaps between the anchors of ALL pairs are aligned exactly in one device batch. */
void mwf_wfa_chain_batch(void *km, const mwf_opt_t *opt, int32_t n, const int32_t *tl, const char *const *ts,
                         const int32_t *ql, const char *const *qs, mwf_rst_t *r);

/* ... and mwf_wfa_auto of n pairs (reference miniwfa.c:898-908 once per record): the exact branch as one batch, the pairs it gives up on through mwf_wfa_chain_batch. */
void mwf_wfa_auto_batch(void *km, const mwf_opt_t *opt, int32_t n, const int32_t *tl, const char *const *ts,
                        const int32_t *ql, const char *const *qs, mwf_rst_t *r);

/* Batch throughput for callers that keep the reference's one-pair-per-call loop (reference main.c:67-72).
 * mwf_wfa_submit() queues one pair (ts / qs / *opt are read later: they must stay valid until the job has been waited for) and returns at once;
 * a dispatcher thread aligns everything submitted so far as ONE mwf_wfa_batch call per option set — while it runs, the caller may keep submitting.
 * mwf_wfa_wait() blocks until the job is done, fills *r exactly as mwf_wfa_exact would (CIGAR allocated from `km` on the waiting thread) and
 * frees the job.  A pair waits at most MWF_COALESCE_US microseconds (default 100) for company unless a wait arrives first.
 * MWF_COALESCE_US=n (n > 0) in the environment also routes plain mwf_wfa_exact calls through the same dispatcher: calls from different host
 * threads that arrive within n microseconds share one launch (opt-in: a lone caller pays up to n microseconds per call). */
typedef struct mwf_job_s mwf_job_t;
mwf_job_t *mwf_wfa_submit(const mwf_opt_t *opt, int32_t tl, const char *ts, int32_t ql, const char *qs);
void mwf_wfa_wait(void *km, mwf_job_t *job, mwf_rst_t *r);
/* diagnostics: batches the dispatcher has run so far, pairs in them */
void mwf_wfa_async_stats(int64_t *n_batches, int64_t *n_jobs);

typedef struct mwf_gpu_s mwf_gpu_t;             /* engine: one device, one stream, one memory pool */
typedef struct mwf_gpu_batch_s mwf_gpu_batch_t; /* a set of pairs resident in that device's HBM */

int         mwf_gpu_device_count(void);
/* device: HIP ordinal.  stream: a hipStream_t to enqueue on, or NULL to let the engine create its own. */
mwf_gpu_t  *mwf_gpu_create(int device, void *stream);
void        mwf_gpu_destroy(mwf_gpu_t *g);
const char *mwf_gpu_last_error(const mwf_gpu_t *g);

/* Packed layout: all sequences back to back in `seqs` (seq_bytes bytes, +>=16 readable bytes of slack);
 * pair i is target seqs[t_off[i], t_off[i]+tl[i]) and query seqs[q_off[i], q_off[i]+ql[i]).
 * _upload copies from host memory; _wrap adopts buffers that already live in this device's HBM
 * (e.g. torch tensors) without copying — they must outlive the batch object. */
mwf_gpu_batch_t *mwf_gpu_batch_upload(mwf_gpu_t *g, int32_t n, const char *seqs, int64_t seq_bytes,
                                      const int64_t *t_off, const int32_t *tl, const int64_t *q_off, const int32_t *ql);
mwf_gpu_batch_t *mwf_gpu_batch_wrap(mwf_gpu_t *g, int32_t n, const void *d_seqs, int64_t seq_bytes,
                                    const int64_t *d_t_off, const int32_t *d_tl, const int64_t *d_q_off, const int32_t *d_ql,
                                    const int32_t *h_tl, const int32_t *h_ql);
void mwf_gpu_batch_free(mwf_gpu_batch_t *b);

/* Align every pair of the batch with `opt`.  Kernels are enqueued on the engine's stream; the call
 * returns after they are enqueued — except on the whole-device kernel (a few long pairs), whose launches are serialised
 * per device and waited for.  Pairs that need a re-run get it in mwf_gpu_batch_results().  Returns 0, or a negative
 * error (see mwf_gpu_last_error). */
int mwf_gpu_batch_align(mwf_gpu_t *g, mwf_gpu_batch_t *b, const mwf_opt_t *opt);

/* Wait for the batch and copy the fixed-size results to host arrays of length n (any may be NULL). */
int mwf_gpu_batch_results(mwf_gpu_t *g, mwf_gpu_batch_t *b, int32_t *s, int64_t *n_iter, int32_t *n_cigar);
/* Device pointers to the same fixed-size results (int32 s[n], int64 n_iter[n], int32 status[n]) for zero-copy consumers.
 * They are FINAL only after mwf_gpu_batch_results() returned: a pair whose window or traceback outgrew the workspace of
 * the kernel it ran on is re-run by that call.  Until then such a pair reads s == -2 (never a valid answer; -1 means
 * "stopped by max_s / max_iter", as in the reference) and status != 0 (0: done, 1: stopped). */
const int32_t *mwf_gpu_batch_dev_scores(const mwf_gpu_batch_t *b);
const int64_t *mwf_gpu_batch_dev_iters(const mwf_gpu_batch_t *b);
const int32_t *mwf_gpu_batch_dev_status(const mwf_gpu_batch_t *b);
/* CIGAR of pair i into dst (capacity cap words); returns n_cigar or a negative error. */
int32_t mwf_gpu_batch_cigar(mwf_gpu_t *g, mwf_gpu_batch_t *b, int32_t i, uint32_t *dst, int32_t cap);
/* Optional: bring every CIGAR of the batch to the host in one copy; mwf_gpu_batch_cigar then serves from that copy. */
int mwf_gpu_batch_fetch_cigars(mwf_gpu_t *g, mwf_gpu_batch_t *b);

/* Timing and counters of the most recent mwf_gpu_batch_align on this engine. */
typedef struct {
	double  kernel_ms;     /* HIP-event time around the alignment kernels on the engine's stream */
	int64_t cells;         /* (penalty,diagonal) cells computed by core passes (= sum of n_iter) */
	int64_t cells_pass1;   /* cells computed by low-memory first passes (not part of n_iter) */
	int32_t n_launches;    /* kernel launches issued */
	int32_t n_retries;     /* pairs re-run with a larger traceback arena */
	int32_t grid, block;   /* geometry of the dominant launch */
	int32_t kernel_kind;   /* 0: one workgroup per pair (generic); 1: one pair across the whole device; 2: one workgroup per pair (band) */
	int64_t dev_bytes;     /* device memory the engine holds now: workspace pools + recycled batch allocations (live batches hold their own) */
	int64_t dev_bytes_peak;/* ... and the most it held since creation or the last mwf_gpu_set(g, "trim", 0) */
	int32_t packed;        /* band kernels: 1 = the packed 16-bit variant (mwf_band2.hip; with block 1024: its span geometry for pairs of up to ~60 kb), 32 = the one-wave-per-pair lane kernel (mwf_lane.hip), 33 = the one-workgroup-per-pair mid kernel (mwf_mid.hip); generic kernel: 16 = 16-bit ring rows */
	int32_t lowmem_two_pass; /* low-memory mode: 1 = the first pass stored no traceback (provenance + snapshots), 0 = checkpoints walked off a full traceback */
} mwf_gpu_stats_t;
void mwf_gpu_get_stats(const mwf_gpu_t *g, mwf_gpu_stats_t *st);

/* Diagnostics: align with `opt` and report the column window [lo,hi] (column = diagonal + tl + 1) of every
 * wavefront slice the core pass opened for `pair`; returns the number of penalties written (<= cap). */
int32_t mwf_gpu_debug_band(mwf_gpu_t *g, mwf_gpu_batch_t *b, const mwf_opt_t *opt, int32_t pair, int32_t *lohi, int32_t cap);

/* Tunables (call before align; every call invalidates the cached plans of the engine's batches).  Thirteen names and one action:
 *   "tb_budget_mb"      traceback arena of the one-workgroup-per-pair kernels (0 = automatic: four fifths of what is free, at most a quarter of the device)
 *   "lowmem_budget_mb"  whole-device kernel, opt.step > 0: a first-pass traceback above this many MB switches to the two-pass form whose first pass stores none (0 = a quarter of the device)
 *   "coop_min_len"      tl + ql from which a batch of at most sixteen pairs runs on the whole-device kernel (0 = 20 000 score-only, 15 000 with CIGAR)
 *   "seq2bit"           1 (default): pairs of plain A/C/G/T are held at 2 bits per base in LDS, any other pair runs on the byte-wise copy; 0: always bytes
 *   "ring16"            generic kernel: 1 (default) = 16-bit ring rows while target length + penalty fits 16 bits (half the HBM traffic); 0: always 32-bit rows
 *   "band_span"         1 (default): pairs beyond the 512-thread geometry (to 62 000 bases per sequence, windows to ~20 000 columns) run on the packed kernel's
 *                       1024-thread span geometry instead of the generic kernel; 0: never; 2: every pair it can take
 *   "wide_slots"        chunk slots per wave of the 512-thread geometry: 0 (default) = four on a batch's first align, three afterwards if that align showed they hold every pair; 3; 4
 *   "lane_max_len"      pairs whose longer sequence has at most this many bases try the one-wave-per-pair kernel first (default 325, weighed by the batch's divergence where it is known; 0: never)
 *   "mid_max_pairs"     a batch of at most this many pairs runs its mid-size pairs on the one-workgroup-per-pair kernel with every ring in LDS (default -1: one pair per CU; 0: never)
 *   "host_results"      1 (default): a score-only batch of up to 64 pairs gets its result arrays in pinned host memory (the mwf_gpu_batch_dev_*() pointers then point there); 0: device memory
 *   "div_aware"         1 (default): the size classes follow the batch's divergence (an 8-mer sketch of a few pairs: on the host while a batch is packed, on the device when one is wrapped); 0: lengths only
 *   "dev_retry"         1 (default): what the short-pair kernel hands back is re-run from a device-side list by a follow-up launch; 0: through the host
 *   "band_fold"         1 (default): score-only with o1 == x the packed kernel folds the gap-open row into its E1 / F1 registers (one row load less per chunk); 0: never
 *   "trim"              (action) free the engine's workspace pools; they grow back on demand.
 * (The hooks tests and profiling scripts force kernels, geometries and failure paths with — "force_kind", "block", "sys_c" ... — are a separate, undeclared entry point,
 * mwf_gpu_test_hook in csrc/mwf_engine.cpp.) */
int mwf_gpu_set(mwf_gpu_t *g, const char *name, int64_t value);

#ifdef __cplusplus
}
#endif

#endif
