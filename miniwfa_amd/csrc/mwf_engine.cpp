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
// (Round 5: the memory helpers moved to mwf_memory.cpp, the plan / launch / retry layer to mwf_plan.cpp; this file is the C ABI.)
#include "mwf_engine.h"

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

// The tunables a user of the library may want (documented in include/miniwfa.h) ...
int mwf_gpu_set(mwf_gpu_t *g, const char *name, int64_t value)
{
	if (!g || !name) return -1;
	if (!strcmp(name, "tb_budget_mb")) g->tb_budget_mb = value;
	else if (!strcmp(name, "lowmem_budget_mb")) g->lowmem_budget_mb = std::max<int64_t>(0, value);
	else if (!strcmp(name, "coop_min_len")) g->coop_min_len = value;
	else if (!strcmp(name, "seq2bit")) g->seq2bit = (int)value;
	else if (!strcmp(name, "ring16")) g->ring16 = (int)value;
	else if (!strcmp(name, "band_span") && value >= 0 && value <= 2) g->band_span = (int)value;
	else if (!strcmp(name, "wide_slots") && (value == 0 || value == 3 || value == 4)) g->wide_slots = (int)value;
	else if (!strcmp(name, "lane_max_len")) g->lane_max_len = (int)std::max<int64_t>(0, std::min<int64_t>(value, 8000));
	else if (!strcmp(name, "mid_max_pairs")) g->mid_max_pairs = (int)std::max<int64_t>(-1, std::min<int64_t>(value, 1 << 20));
	else if (!strcmp(name, "host_results")) g->res_pin_on = value != 0;
	else if (!strcmp(name, "div_aware")) g->div_aware = value != 0;
	else if (!strcmp(name, "dev_retry")) g->dev_retry = value != 0;
	else if (!strcmp(name, "band_fold")) g->band_fold = value != 0;
	else if (!strcmp(name, "trim")) { (void)hipSetDevice(g->device); trim(g); }
	else return -1;
	++g->tun_gen; // (whatever the tunable: no hand-kept list of "the ones that classify" to forget an entry of)
	return 0;
}

// ... and the hooks the tests force kernels, geometries and failure paths with (round 6: split off mwf_gpu_set — 30 names had grown there; four that no
// test or document needed are gone: coop_launch, sys_p2, ring16_block, slots_per_cu).  Exported for tests/ and profiles/, not declared in include/miniwfa.h.
int mwf_gpu_test_hook(mwf_gpu_t *g, const char *name, int64_t value)
{
	if (!g || !name) return -1;
	if (!strcmp(name, "block")) {
		if (value != 0 && value != 64 && value != 128 && value != 256 && value != 512 && value != 768 && value != 1024) return -1;
		g->block = (int)value;
	} else if (!strcmp(name, "force_kind")) g->force_kind = (int)value;
	else if (!strcmp(name, "band_pack")) g->band_pack = (int)value;
	else if (!strcmp(name, "lane_chunks") && value >= 0 && value <= 4) g->lane_chunks = (int)value;
	else if (!strcmp(name, "mid_block") && (value == 0 || value == 256 || value == 512 || value == 1024)) g->mid_block = (int)value;
	else if (!strcmp(name, "lds_e2")) g->lds_e2 = value != 0;
	else if (!strcmp(name, "scalar_generic")) g->scalar_generic = value != 0;
	else if (!strcmp(name, "coop_spin_limit")) g->coop_spin_limit = std::max<int64_t>(0, std::min<int64_t>(value, 0x7fffffff));
	else if (!strcmp(name, "coop_tb_cap_mb")) g->coop_tb_cap = std::max<int64_t>(1, value) << 20;
	else if (!strcmp(name, "coop_grid")) g->coop_grid_cap = (int)std::max<int64_t>(0, value);
	else if (!strcmp(name, "sys_p") && sys_p_supported((int)value)) g->sys_p = (int)value; // (8; 4 and 16 only in builds with -DMWF_SYS_ALL_P)
	else if (!strcmp(name, "sys_c") && (value == 0 || sys_c_supported((int)value))) g->sys_c = (int)value; // (2: builds with -DMWF_SYS_C2 only)
	else return -1;
	++g->tun_gen;
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
	// the batch's divergence from an 8-mer sketch of a few of its pairs, like a batch built from host memory gets while it is packed (the
	// classes of a resident 15 % or 30 % batch were drawn from its lengths alone and ran ~every pair twice: VERDICT r5)
	if (g->div_aware) b->div_est = estimate_divergence_device(g, b);
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

} // extern "C"


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
	static const bool timing = getenv("MWF_SHARE_TIMING") != nullptr; // (diagnostics: where a drop-in batch call's time goes)
	const auto t_0 = std::chrono::steady_clock::now();
	mwf_gpu_batch_t *b = batch_from_host(g, m, ltl.data(), lts.data(), lql.data(), lqs.data(), nullptr, 0, nullptr, nullptr);
	const bool cigar = (opt->flag & MWF_F_CIGAR) != 0;
	if (!b) R.err = std::string("batch upload failed: ") + mwf_gpu_last_error(g);
	else {
		R.s.resize((size_t)m), R.ncig.resize((size_t)m), R.iter.resize((size_t)m);
		const auto t_1 = std::chrono::steady_clock::now();
		int rc = mwf_gpu_batch_align(g, b, opt);
		const auto t_2 = std::chrono::steady_clock::now();
		rc = rc || mwf_gpu_batch_results(g, b, R.s.data(), R.iter.data(), R.ncig.data());
		const auto t_3 = std::chrono::steady_clock::now();
		rc = rc || (cigar && fetch_cigars(g, b));
		if (timing) {
			auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
			int32_t longest = 0;
			for (int32_t j = 0; j < m; ++j) longest = std::max(longest, std::max(ltl[j], lql[j]));
			mwf_gpu_stats_t st;
			mwf_gpu_get_stats(g, &st);
			fprintf(stderr, "[libmwf_hip] share of %d pairs (longest %d): upload %.3f ms, align %.3f ms (kernels %.3f ms, %d launch(es), %lld re-run), results %.3f ms, cigars %.3f ms\n", m, longest,
			        ms(t_0, t_1), ms(t_1, t_2), st.kernel_ms, (int)st.n_launches, (long long)st.n_retries, ms(t_2, t_3), ms(t_3, std::chrono::steady_clock::now()));
		}
		if (rc)
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
	if (coalesce_window_us() > 0) { exact_coalesced(km, opt, tl, ts, ql, qs, r); return; } // MWF_COALESCE_US: calls of different host threads share a launch (mwf_async.cpp)
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

// mwf_wfa_auto of n records (reference main.c:67-72 loops it): the exact branch of every pair as one batch, and the pairs it gave up on (1e8 cells, miniwfa.c:900-903)
// through mwf_wfa_chain_batch — r[i].n_iter stays the exact branch's count, as in the reference, which never writes it in chain mode.
void mwf_wfa_auto_batch(void *km, const mwf_opt_t *opt0, int32_t n, const int32_t *tl, const char *const *ts, const int32_t *ql, const char *const *qs, mwf_rst_t *r)
{
	if (n <= 0) return;
	mwf_opt_t opt = *opt0;
	opt.step = 0, opt.max_iter = 100000000;
	mwf_wfa_batch(km, &opt, n, tl, ts, ql, qs, r);
	std::vector<int32_t> idx;
	for (int32_t i = 0; i < n; ++i)
		if (r[i].s < 0) idx.push_back(i);
	if (idx.empty()) return;
	if (opt.flag & MWF_F_CIGAR) opt.step = 5000;
	opt.max_iter = -1;
	const int32_t m = (int32_t)idx.size();
	std::vector<int32_t> ctl((size_t)m), cql((size_t)m);
	std::vector<const char*> cts((size_t)m), cqs((size_t)m);
	std::vector<mwf_rst_t> cr((size_t)m);
	for (int32_t j = 0; j < m; ++j) ctl[j] = tl[idx[j]], cql[j] = ql[idx[j]], cts[j] = ts[idx[j]], cqs[j] = qs[idx[j]], cr[j] = r[idx[j]];
	mwf_wfa_chain_batch(km, &opt, m, ctl.data(), cts.data(), cql.data(), cqs.data(), cr.data());
	for (int32_t j = 0; j < m; ++j) r[idx[j]] = cr[j];
}

} // extern "C"

