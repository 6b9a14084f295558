// mwf_async.cpp — batch throughput for callers that keep the reference's one-pair-per-call shape (reference main.c:67-72 loops
// mwf_wfa_exact over its pairs; miniwfa.c:603-615 aligns one pair per call).
//
// A single short pair is latency-, not throughput-work on a GPU (one kernel launch, one wave or workgroup busy: a 1 kb pair costs
// ~300 us per call against ~150 us on one host core), and the batch entry point needs the caller rewritten around arrays.  Two ways
// to get the batch kernels without that:
//   * mwf_wfa_submit() / mwf_wfa_wait(): the loop body becomes "submit", the results are collected afterwards (or a few iterations
//     later).  Submitted pairs are gathered by one dispatcher thread and aligned as ONE mwf_wfa_batch call per option set — while it
//     runs, the caller keeps submitting the next batch;
//   * MWF_COALESCE_US=n in the environment: plain mwf_wfa_exact calls arriving from DIFFERENT host threads within n microseconds of
//     each other share one batch launch (opt-in: a lone caller pays up to n us of waiting per call).
// Nothing here touches a device: the dispatcher calls mwf_wfa_batch (mwf_engine.cpp), which takes a pooled engine like any caller.
#include "mwf_engine.h"

#include <condition_variable>

struct mwf_job_s {
	mwf_opt_t opt;
	int32_t tl, ql;
	const char *ts, *qs;
	mwf_rst_t res;        // (CIGAR from libc malloc until mwf_wfa_wait moves it into the caller's arena)
	bool done = false;
	std::chrono::steady_clock::time_point t_submit;
};

namespace {

constexpr size_t kMaxBatch = 16384; // pairs per dispatcher batch

struct Dispatcher {
	std::mutex mu;
	std::condition_variable cv_work, cv_done;
	std::vector<mwf_job_t*> pending;
	bool flush_now = false, started = false;
	int64_t window_us = 100;      // how long the oldest submitted pair may wait for company
	int64_t n_batches = 0, n_jobs = 0;
};
Dispatcher &disp() { static Dispatcher *d = new Dispatcher(); return *d; } // (never destroyed: its thread may outlive static destructors)

void run_jobs(std::vector<mwf_job_t*> &jobs)
{
	// one mwf_wfa_batch per option set (byte-equal mwf_opt_t), in submission order inside a set
	std::vector<char> used(jobs.size(), 0);
	for (size_t a = 0; a < jobs.size(); ++a) {
		if (used[a]) continue;
		std::vector<size_t> idx;
		for (size_t b = a; b < jobs.size(); ++b)
			if (!used[b] && memcmp(&jobs[b]->opt, &jobs[a]->opt, sizeof(mwf_opt_t)) == 0) idx.push_back(b), used[b] = 1;
		const int32_t n = (int32_t)idx.size();
		std::vector<int32_t> tl((size_t)n), ql((size_t)n);
		std::vector<const char*> ts((size_t)n), qs((size_t)n);
		std::vector<mwf_rst_t> r((size_t)n);
		for (int32_t j = 0; j < n; ++j) tl[j] = jobs[idx[j]]->tl, ql[j] = jobs[idx[j]]->ql, ts[j] = jobs[idx[j]]->ts, qs[j] = jobs[idx[j]]->qs;
		mwf_wfa_batch(nullptr, &jobs[a]->opt, n, tl.data(), ts.data(), ql.data(), qs.data(), r.data()); // km == NULL: CIGARs from libc
		for (int32_t j = 0; j < n; ++j) jobs[idx[j]]->res = r[j];
	}
}

void dispatcher_main()
{
	Dispatcher &D = disp();
	std::unique_lock<std::mutex> lock(D.mu);
	for (;;) {
		D.cv_work.wait(lock, [&] { return !D.pending.empty(); });
		// the oldest pair waits at most window_us for company; a waiter (explicit API) or a full batch flushes at once
		const auto deadline = D.pending.front()->t_submit + std::chrono::microseconds(D.window_us);
		while (!D.flush_now && D.pending.size() < kMaxBatch && std::chrono::steady_clock::now() < deadline) D.cv_work.wait_until(lock, deadline);
		std::vector<mwf_job_t*> jobs;
		jobs.swap(D.pending);
		D.flush_now = false;
		lock.unlock();
		run_jobs(jobs);
		lock.lock();
		for (mwf_job_t *j : jobs) j->done = true;
		D.n_batches += 1, D.n_jobs += (int64_t)jobs.size();
		D.cv_done.notify_all();
	}
}

mwf_job_t *submit(const mwf_opt_t *opt, int32_t tl, const char *ts, int32_t ql, const char *qs, int64_t window_us)
{
	mwf_job_t *j = new mwf_job_t();
	j->opt = *opt, j->tl = tl, j->ql = ql, j->ts = ts, j->qs = qs;
	memset(&j->res, 0, sizeof(j->res));
	j->t_submit = std::chrono::steady_clock::now();
	Dispatcher &D = disp();
	std::lock_guard<std::mutex> lock(D.mu);
	if (!D.started) {
		D.started = true;
		std::thread(dispatcher_main).detach();
	}
	if (window_us > 0) D.window_us = window_us;
	D.pending.push_back(j);
	if (D.pending.size() == 1 || D.pending.size() >= kMaxBatch) D.cv_work.notify_one();
	return j;
}

void wait(void *km, mwf_job_t *j, mwf_rst_t *r, bool flush)
{
	Dispatcher &D = disp();
	{
		std::unique_lock<std::mutex> lock(D.mu);
		if (!j->done && flush) D.flush_now = true, D.cv_work.notify_one();
		D.cv_done.wait(lock, [&] { return j->done; });
	}
	*r = j->res;
	if (km && r->cigar) { // reference krelocate()s the CIGAR into the caller's arena (miniwfa.c:434): on the caller's thread — kalloc has no locks
		uint32_t *c = (uint32_t*)kmalloc(km, (size_t)r->n_cigar * 4);
		memcpy(c, r->cigar, (size_t)r->n_cigar * 4);
		free(r->cigar);
		r->cigar = c;
	}
	delete j;
}

} // namespace

namespace mwf {
namespace host {

// MWF_COALESCE_US (read once): > 0 = mwf_wfa_exact goes through the dispatcher with that window
int64_t coalesce_window_us()
{
	static const int64_t v = [] {
		const char *e = getenv("MWF_COALESCE_US");
		if (!e || !*e) return (int64_t)0;
		char *end = nullptr;
		const long long x = strtoll(e, &end, 10);
		if (*end != 0 || x < 0 || x > 1000000) fatal("MWF_COALESCE_US must be a number of microseconds (0 ... 1000000)", e);
		return (int64_t)x;
	}();
	return v;
}

void exact_coalesced(void *km, const mwf_opt_t *opt, int32_t tl, const char *ts, int32_t ql, const char *qs, mwf_rst_t *r)
{
	wait(km, submit(opt, tl, ts, ql, qs, coalesce_window_us()), r, false); // (no flush: the window is what lets other threads' calls join)
}

} // namespace host
} // namespace mwf

extern "C" {

mwf_job_t *mwf_wfa_submit(const mwf_opt_t *opt, int32_t tl, const char *ts, int32_t ql, const char *qs)
{
	if (!opt || tl < 0 || ql < 0) return nullptr;
	if (const char *why = validate(*opt)) fatal("mwf_wfa_submit", why);
	return submit(opt, tl, ts, ql, qs, coalesce_window_us());
}

void mwf_wfa_wait(void *km, mwf_job_t *job, mwf_rst_t *r)
{
	if (!job || !r) return;
	wait(km, job, r, true);
}

void mwf_wfa_async_stats(int64_t *n_batches, int64_t *n_jobs)
{
	Dispatcher &D = disp();
	std::lock_guard<std::mutex> lock(D.mu);
	if (n_batches) *n_batches = D.n_batches;
	if (n_jobs) *n_jobs = D.n_jobs;
}

} // extern "C"
