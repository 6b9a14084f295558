// mwf_memory.cpp — device memory of libmwf_hip.so's host side: buffers that only grow, the batch blocks the engine recycles, the pinned
// staging buffer uploads and downloads go through, and the construction of a batch from host memory (lengths, processing order,
// alphabet classification, divergence sketch, ONE upload).  Split from mwf_engine.cpp in round 5 (no behaviour change).
#include "mwf_engine.h"

namespace mwf {
namespace host {

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

// ---- batches ------------------------------------------------------------------------------------------------------------

// Carve the batch's one allocation.  [order | t_off q_off tl ql seqs (owned inputs) | results]; the input part is laid
// out exactly as upload_segments() streams it.

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
	L.retry = at, at += (size_t)kRetrySlots * kRetryCap * 4;
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
	b->d_retry_ids = (int32_t*)(base + L.retry);
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
// One sampled pair's estimate from its hit count: hits of `tot` query 8-mers in a target prefix of lt bases.
double sketch_divergence(int32_t hit, int32_t tot, int32_t lt)
{
	constexpr int K = 8;
	const double fp = 1.0 - std::exp(-(double)(lt - K + 1) / 65536.0); // chance hits
	double f = ((double)hit / std::max(tot, 1) - fp) / (1.0 - fp);
	f = std::min(1.0, std::max(f, 1e-3));
	return 1.0 - std::pow(f, 1.0 / K);
}

// The batch's figure from the sampled pairs': the MEDIAN (a mean lets a few unrelated pairs among the samples — chain-mode gap
// fills are such a mix — push every pair of the batch into wider, slower classes, and a low mean under-sizes the diverged ones), and "unknown"
// (0: the length-only classes) when the samples disagree strongly — quartiles more than a factor of three apart: no one figure describes the batch.
float combine_divergence(std::vector<double> &d)
{
	if (d.empty()) return 0.f;
	std::sort(d.begin(), d.end());
	const size_t n = d.size();
	const double med = n & 1 ? d[n / 2] : 0.5 * (d[n / 2 - 1] + d[n / 2]);
	if (n >= 4) {
		const double q1 = d[n / 4], q3 = d[(3 * n) / 4];
		if (q3 > 3.0 * std::max(q1, 0.02)) return 0.f;
	}
	return (float)med;
}

float estimate_divergence(int32_t n, const int32_t *tl, const int32_t *ql, const std::function<const uint8_t*(int32_t, bool)> &seq)
{
	constexpr int K = 8;
	constexpr uint32_t MASK = (1u << (2 * K)) - 1;
	std::vector<uint64_t> bits((size_t)1 << (2 * K - 6));
	std::vector<double> est;
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
		est.push_back(sketch_divergence(hit, tot, lt));
	}
	return combine_divergence(est);
}

// The same for a batch whose sequences live in device memory (mwf_gpu_batch_wrap): one small kernel over the sampled pairs' prefixes, its
// sixteen counts read back (a launch and a 64-byte copy: ~30 us per wrapped batch, once).  0 (unknown) when anything fails — an estimate only.
float estimate_divergence_device(mwf_gpu_t *g, mwf_gpu_batch_t *b)
{
	const int32_t n = b->n, samples = std::min(16, n);
	if (n <= 0 || !b->d_seqs || !b->d_t_off || !b->d_q_off || !b->d_tl || !b->d_ql) return 0.f;
	int32_t *d_out = b->d_dbg4; // (16 bytes per pair of scratch behind the results: nothing uses it before an align)
	if (launch_sketch(b->d_seqs, b->d_t_off, b->d_tl, b->d_q_off, b->d_ql, n, samples, d_out, g->stream)) { (void)hipGetLastError(); return 0.f; }
	int32_t hits[16];
	if (hipMemcpyAsync(hits, d_out, (size_t)samples * 4, hipMemcpyDeviceToHost, g->stream) != hipSuccess || hipStreamSynchronize(g->stream) != hipSuccess) { (void)hipGetLastError(); return 0.f; }
	std::vector<double> est;
	for (int32_t k = 0; k < samples; ++k) {
		const int32_t i = (int32_t)((int64_t)k * n / samples);
		const int32_t lt = std::min(b->h_tl[i], 1500), lq = std::min(b->h_ql[i], 1500);
		if (lt < 32 || lq < 32) continue;
		est.push_back(sketch_divergence(hits[k], lq - 7, lt));
	}
	return combine_divergence(est);
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

} // namespace host
} // namespace mwf
