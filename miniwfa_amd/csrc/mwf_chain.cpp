// mwf_chain.cpp — mwf_wfa_chain(): the chaining heuristic of miniwfa (reference miniwfa.c:617-896) on top of the
// GPU exact path.
//
// The heuristic itself is small host-side integer work and stays on the host: collect every forward k-mer of both
// sequences, pair the ones that occur at most max_occ times on each side, keep a longest colinear subset, drop anchors
// sitting on gapless runs shorter than min_len, then walk the anchors.  What the reference does one at a time inside
// that walk — an exact alignment of every gap between anchors (miniwfa.c:877) — is collected here and run as ONE device
// batch (mwf_wfa_batch), which is the only expensive part.  Results (penalty and CIGAR) are the reference's:
// tests/test_gpu_parity.py checks them against vectors produced by the reference's own chain mode.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <atomic>
#include <chrono>
#include <climits>
#include <thread>
#include "miniwfa.h"
#include "kalloc.h"

namespace {

// A/a C/c G/g T/t U/u -> 0..3, anything else breaks the k-mer (reference seq_nt4_table, miniwfa.c:699-716)
struct BaseCodes {
	uint8_t v[256];
	BaseCodes()
	{
		memset(v, 4, sizeof(v));
		v['A'] = v['a'] = 0, v['C'] = v['c'] = 1, v['G'] = v['g'] = 2, v['T'] = v['t'] = v['U'] = v['u'] = 3;
	}
};
const BaseCodes kBaseCodes;
inline int base_code(unsigned char c) { return kBaseCodes.v[c]; }

// Ascending sort of 64-bit keys: least-significant-digit radix sort on the bytes that differ between keys (the reference sorts its k-mers and
// anchor pairs with a radix sort too, miniwfa.c:699-716 — tens of thousands of keys per call, where std::sort was a third of a call's host time);
// any correct sort gives the reference's order: the keys are compared whole.
void sort_u64(std::vector<uint64_t> &a)
{
	const size_t n = a.size();
	if (n < 512) { std::sort(a.begin(), a.end()); return; }
	uint64_t all_or = 0, all_and = ~0ULL;
	for (uint64_t v : a) all_or |= v, all_and &= v;
	const uint64_t varies = all_or ^ all_and; // bits that are not the same in every key
	std::vector<uint64_t> tmp(n);
	uint64_t *src = a.data(), *dst = tmp.data();
	for (int shift = 0; shift < 64; shift += 8) {
		if (((varies >> shift) & 0xff) == 0) continue;
		size_t count[256] = {0};
		for (size_t i = 0; i < n; ++i) ++count[(src[i] >> shift) & 0xff];
		size_t at = 0;
		for (int d = 0; d < 256; ++d) { const size_t c = count[d]; count[d] = at, at += c; }
		for (size_t i = 0; i < n; ++i) dst[count[(src[i] >> shift) & 0xff]++] = src[i];
		std::swap(src, dst);
	}
	if (src != a.data()) std::copy(src, src + n, a.data());
}

// Stable least-significant-digit radix sort of 64-bit keys on the bit field [lo_bit, lo_bit + n_bits) only, digits of up to 13 bits (two passes for the 26 bits
// of a 13-mer).  Round 6: the k-mer list is generated target first, then query, positions ascending — i.e. already in the order the full key
// (kmer << 1 | rid) << 32 | pos breaks ties in — so a STABLE sort on the k-mer bits alone yields exactly the fully sorted list, in two passes instead of six.
void radix_sort_field(std::vector<uint64_t> &a, int lo_bit, int n_bits)
{
	const size_t n = a.size();
	if (n < 2 || n_bits <= 0) return;
	std::vector<uint64_t> tmp(n);
	uint64_t *src = a.data(), *dst = tmp.data();
	std::vector<uint32_t> count;
	for (int done = 0; done < n_bits;) {
		const int w = std::min(13, n_bits - done), shift = lo_bit + done;
		const uint64_t m = (1ULL << w) - 1;
		count.assign((size_t)1 << w, 0);
		for (size_t i = 0; i < n; ++i) ++count[(src[i] >> shift) & m];
		uint32_t at = 0;
		for (size_t d = 0; d < count.size(); ++d) { const uint32_t c = count[d]; count[d] = at, at += c; }
		for (size_t i = 0; i < n; ++i) dst[count[(src[i] >> shift) & m]++] = src[i];
		std::swap(src, dst);
		done += w;
	}
	if (src != a.data()) std::copy(src, src + n, a.data());
}

// every k-mer of seq as (kmer << 1 | rid) << 32 | end position  (reference mg_fc_kmer, miniwfa.c:718-730)
void collect_kmers(int32_t len, const char *seq, int rid, int k, std::vector<uint64_t> &out)
{
	const uint64_t mask = (1ULL << (2 * k)) - 1;
	uint64_t x = 0;
	int32_t run = 0;
	for (int32_t i = 0; i < len; ++i) {
		const int c = base_code((unsigned char)seq[i]);
		if (c < 4) {
			x = (x << 2 | (uint64_t)c) & mask;
			if (++run >= k) out.push_back((x << 1 | (uint64_t)rid) << 32 | (uint32_t)i);
		} else run = 0, x = 0;
	}
}

// Longest strictly increasing subsequence, patience sorting with predecessor links (reference mg_lis_64,
// miniwfa.c:678-697).  Among equally long answers the one this construction yields is the one the reference returns.
std::vector<int32_t> longest_increasing(const std::vector<uint64_t> &a)
{
	const int32_t n = (int32_t)a.size();
	std::vector<int32_t> tail(n + 1, 0), pred(n, 0);
	std::vector<uint64_t> tailv(n + 1, 0); // a[tail[.]]: the binary search then walks one contiguous array instead of chasing indices into a
	int32_t L = 0;
	for (int32_t i = 0; i < n; ++i) {
		const uint64_t v = a[i];
		int32_t lo = 1, hi = L;
		if (L > 0 && tailv[L] < v) lo = L + 1; // (colinear anchors: nearly every element extends the longest run)
		else while (lo <= hi) {
			const int32_t mid = (lo + hi + 1) >> 1;
			if (tailv[mid] < v) lo = mid + 1;
			else hi = mid - 1;
		}
		pred[i] = tail[lo - 1];
		tail[lo] = i, tailv[lo] = v;
		if (lo > L) L = lo;
	}
	std::vector<int32_t> out(L);
	for (int32_t i = L - 1, k = L ? tail[L] : 0; i >= 0; --i) out[i] = k, k = pred[k];
	return out;
}

// Colinear anchors (target end << 32 | query end), reference mg_chain, miniwfa.c:732-784
std::vector<uint64_t> chain_anchors(int32_t tl, const char *ts, int32_t ql, const char *qs, int k, int max_occ)
{
	std::vector<uint64_t> anchors;
	if (tl < k || ql < k) return anchors;
	if (k < 2 || k > 15) { fprintf(stderr, "[libmwf_hip] mwf_wfa_chain: kmer must be in [2,15]\n"); abort(); }
	std::vector<uint64_t> a;
	a.reserve((size_t)tl + ql);
	collect_kmers(tl, ts, 0, k, a);
	collect_kmers(ql, qs, 1, k, a);
	radix_sort_field(a, 33, 2 * k); // (keys are unique: the reference's fully sorted order — ties of the k-mer broken by rid, then position — is the generation order)
	std::vector<uint64_t> b;
	for (size_t i0 = 0, i = 1; i <= a.size(); ++i) {
		if (i == a.size() || (a[i0] >> 33) != (a[i] >> 33)) {
			size_t j = i0;
			while (j < i && ((a[j] >> 32) & 1) == 0) ++j; // target occurrences sort first
			if (j > i0 && j < i && (int64_t)(j - i0) <= max_occ && (int64_t)(i - j) <= max_occ)
				for (size_t s = i0; s < j; ++s)
					for (size_t t = j; t < i; ++t)
						b.push_back(a[s] << 32 | (uint32_t)a[t]);
			i0 = i;
		}
	}
	// b must be ordered by target position, then query position (the reference radix-sorts it whole, miniwfa.c:760).  Every pair of one target position
	// comes out of ONE k-mer group with its query positions ascending, so a stable counting sort on the target position alone is that order: one pass.
	{
		std::vector<uint32_t> at((size_t)tl + 1, 0);
		for (uint64_t v : b) ++at[(size_t)(v >> 32)];
		uint32_t sum = 0;
		for (uint32_t &c : at) { const uint32_t x = c; c = sum, sum += x; }
		std::vector<uint64_t> sorted(b.size());
		for (uint64_t v : b) sorted[at[(size_t)(v >> 32)]++] = v;
		b.swap(sorted);
	}
	for (uint64_t &v : b) v = v >> 32 | v << 32; // order by target position, compare by query position
	const std::vector<int32_t> lis = longest_increasing(b);
	anchors.reserve(lis.size());
	for (int32_t idx : lis) anchors.push_back(b[idx] >> 32 | b[idx] << 32);
	return anchors;
}

// Drop anchors that sit on a gapless run shorter than min_len (reference wf_anchor_filter, miniwfa.c:829-848)
void filter_anchors(std::vector<uint64_t> &a, int32_t tl, int32_t ql, int32_t k, int32_t min_len)
{
	const int32_t n = (int32_t)a.size();
	int32_t x0 = 0, y0 = 0, x1 = 0, start = -1, run = 0;
	for (int32_t i = 0; i <= n; ++i) {
		int32_t x, y;
		if (i == n) x = tl, y = ql;
		else x = (int32_t)(a[i] >> 32) + 1, y = (int32_t)a[i] + 1;
		if (x - x0 != y - y0) {
			if (run < min_len)
				for (int32_t j = start > 0 ? start : 0; j < i; ++j) a[j] = 0;
			x0 = x, y0 = y, start = i, run = k;
		} else run += x - x1;
		x1 = x;
	}
	a.erase(std::remove(a.begin(), a.end(), (uint64_t)0), a.end());
}

// fraction of shared k-mers, the larger of the two directions (reference mwf_ksim, miniwfa.c:786-812)
double kmer_similarity(int32_t l1, const char *s1, int32_t l2, const char *s2, int k)
{
	if (l1 < k || l2 < k) return 0;
	std::vector<uint64_t> a;
	collect_kmers(l1, s1, 0, k, a);
	collect_kmers(l2, s2, 1, k, a);
	sort_u64(a);
	int64_t n1 = 0, n2 = 0, shared = 0;
	for (size_t i0 = 0, i = 1; i <= a.size(); ++i) {
		if (i == a.size() || (a[i0] >> 33) != (a[i] >> 33)) {
			size_t j = i0;
			while (j < i && ((a[j] >> 32) & 1) == 0) ++j;
			const int64_t m1 = (int64_t)(j - i0), m2 = (int64_t)(i - j);
			n1 += m1, n2 += m2;
			if (m1 > 0 && m2 > 0) shared += std::min(m1, m2);
			i0 = i;
		}
	}
	const double p1 = (double)shared / n1, p2 = (double)shared / n2;
	return p1 > p2 ? p1 : p2;
}

struct CigarBuf {
	std::vector<uint32_t> w;
	void push(uint32_t op, int32_t len) // reference wf_cigar_push1, miniwfa.c:51-62
	{
		if (!w.empty() && (w.back() & 0xf) == op) w.back() += (uint32_t)len << 4;
		else w.push_back((uint32_t)len << 4 | op);
	}
	void append(int32_t n, const uint32_t *c) // reference wf_cigar_push, miniwfa.c:816-827: only the first op may merge
	{
		if (n == 0) return;
		push(c[0] & 0xf, (int32_t)(c[0] >> 4));
		w.insert(w.end(), c + 1, c + n);
	}
};

struct Segment {       // one step of the walk over the anchors
	int kind;          // 0 '=' run, 1 exact gap fill, 2 too-diverged block (D then I), 3 pure deletion, 4 pure insertion
	int32_t x0, y0, dx, dy;
	int32_t fill;      // index into the gap-fill batch (kind 1)
};

inline int32_t gap_cost(const mwf_opt_t *o, int32_t len)
{
	const int32_t a = o->o2 + len * o->e2, b = o->o1 + len * o->e1;
	return a < b ? a : b;
}

// what the walk over one pair's anchors decided (reference miniwfa.c:861-891), and the gaps it wants aligned exactly
struct ChainPlan {
	std::vector<Segment> segs;
	std::vector<int32_t> ftl, fql;
	std::vector<const char*> fts, fqs;
	int32_t n_anchors = 0;
	double ms_anchors = 0;
};

void plan_chain(const mwf_opt_t *opt, int32_t tl, const char *ts, int32_t ql, const char *qs, ChainPlan &P)
{
	const auto t_0 = std::chrono::steady_clock::now();
	std::vector<uint64_t> anchors = chain_anchors(tl, ts, ql, qs, opt->kmer, opt->max_occ);
	filter_anchors(anchors, tl, ql, opt->kmer, opt->min_len);
	P.ms_anchors = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_0).count();
	const int32_t n_a = P.n_anchors = (int32_t)anchors.size();
	int32_t x0 = 0, y0 = 0;
	for (int32_t i = 0; i <= n_a; ++i) {
		int32_t x1, y1;
		if (i == n_a) x1 = tl, y1 = ql;
		else x1 = (int32_t)(anchors[i] >> 32) + 1, y1 = (int32_t)anchors[i] + 1;
		Segment sg{-1, x0, y0, x1 - x0, y1 - y0, -1};
		if (i < n_a && x1 - x0 == y1 - y0 && x1 - x0 <= opt->kmer) sg.kind = 0;
		else if (x0 < x1 && y0 < y1) {
			if (x1 - x0 >= 10000 && y1 - y0 >= 10000 && kmer_similarity(x1 - x0, ts + x0, y1 - y0, qs + y0, opt->kmer) < 0.02) sg.kind = 2;
			else {
				sg.kind = 1, sg.fill = (int32_t)P.ftl.size();
				P.ftl.push_back(x1 - x0), P.fql.push_back(y1 - y0), P.fts.push_back(ts + x0), P.fqs.push_back(qs + y0);
			}
		} else if (x0 < x1) sg.kind = 3;
		else if (y0 < y1) sg.kind = 4;
		if (sg.kind >= 0) P.segs.push_back(sg);
		x0 = x1, y0 = y1;
	}
}

// the pair's penalty and CIGAR from its segments and the answers of its gap fills (fills[0] is the pair's first)
void stitch_chain(void *km, const mwf_opt_t *opt, const ChainPlan &P, const mwf_rst_t *fills, mwf_rst_t *r)
{
	const bool want_cigar = (opt->flag & MWF_F_CIGAR) != 0;
	CigarBuf c;
	int32_t score = 0;
	for (const Segment &sg : P.segs) {
		switch (sg.kind) {
		case 0: if (want_cigar) c.push(7, sg.dx); break;
		case 1:
			if (want_cigar) c.append(fills[sg.fill].n_cigar, fills[sg.fill].cigar);
			score += fills[sg.fill].s;
			break;
		case 2:
			if (want_cigar) c.push(2, sg.dx), c.push(1, sg.dy);
			score += opt->o2 * 2 + opt->e2 * (sg.dx + sg.dy);
			break;
		case 3: c.push(2, sg.dx), score += gap_cost(opt, sg.dx); break; // pushed even without MWF_F_CIGAR (miniwfa.c:883-885)
		case 4: c.push(1, sg.dy), score += gap_cost(opt, sg.dy); break;
		}
	}
	r->s = score; // n_iter is left as the caller had it (miniwfa.c:850-896 never writes it)
	r->n_cigar = (int32_t)c.w.size();
	r->cigar = nullptr;
	if (!c.w.empty()) {
		r->cigar = (uint32_t*)kmalloc(km, c.w.size() * sizeof(uint32_t));
		memcpy(r->cigar, c.w.data(), c.w.size() * sizeof(uint32_t));
	}
}

} // namespace

extern "C" void mwf_wfa_chain(void *km, const mwf_opt_t *opt, int32_t tl, const char *ts, int32_t ql, const char *qs, mwf_rst_t *r)
{
	static const bool timing = getenv("MWF_CHAIN_TIMING") != nullptr; // (diagnostics: where a call's time goes)
	const auto t_0 = std::chrono::steady_clock::now();
	ChainPlan P;
	plan_chain(opt, tl, ts, ql, qs, P);
	// ---- every gap fill in one device batch (the reference calls mwf_wfa_exact per gap, miniwfa.c:877)
	std::vector<mwf_rst_t> fills(P.ftl.size());
	const auto t_2 = std::chrono::steady_clock::now();
	if (!P.ftl.empty())
		mwf_wfa_batch(nullptr, opt, (int32_t)P.ftl.size(), P.ftl.data(), P.fts.data(), P.fql.data(), P.fqs.data(), fills.data());
	const auto t_3 = std::chrono::steady_clock::now();
	stitch_chain(km, opt, P, fills.data(), r);
	for (mwf_rst_t &f : fills) free(f.cigar);
	if (timing) {
		auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
		int64_t fill_bases = 0;
		int32_t longest = 0, n_100 = 0, n_1000 = 0;
		for (size_t i = 0; i < P.ftl.size(); ++i) {
			fill_bases += P.ftl[i] + P.fql[i];
			const int32_t l = std::max(P.ftl[i], P.fql[i]);
			longest = std::max(longest, l), n_100 += l > 100, n_1000 += l > 1000;
		}
		fprintf(stderr, "[libmwf_hip] chain fills: longest %d, %d above 100 bp, %d above 1000 bp\n", longest, n_100, n_1000);
		fprintf(stderr, "[libmwf_hip] chain %d x %d: anchors %.2f ms (%d kept), gaps %.2f ms, batch of %zu fills (%lld bases) %.2f ms, stitch %.2f ms\n", tl, ql, P.ms_anchors, P.n_anchors,
		        ms(t_0, t_2) - P.ms_anchors, P.ftl.size(), (long long)fill_bases, ms(t_2, t_3), ms(t_3, std::chrono::steady_clock::now()));
	}
}

// Many pairs in chain mode (the reference's test program loops mwf_wfa_chain over its records, main.c:67-72): the chaining of every pair on a few host threads,
// then the gap fills of ALL pairs in one device batch, then the stitching on the caller's thread (kalloc arenas are not thread-safe).  Pair by pair the answers of
// mwf_wfa_chain; a call per pair pays its own launch, copies and the ~800 penalties of its longest fill one after the other.
extern "C" void mwf_wfa_chain_batch(void *km, const mwf_opt_t *opt, int32_t n, const int32_t *tl, const char *const *ts, const int32_t *ql, const char *const *qs, mwf_rst_t *r)
{
	if (n <= 0) return;
	std::vector<ChainPlan> plans((size_t)n);
	{
		const int n_thr = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(n, 16), (int64_t)std::thread::hardware_concurrency()));
		std::atomic<int32_t> next{0};
		auto work = [&] {
			for (int32_t i; (i = next.fetch_add(1)) < n;) plan_chain(opt, tl[i], ts[i], ql[i], qs[i], plans[(size_t)i]);
		};
		std::vector<std::thread> pool;
		for (int t = 1; t < n_thr; ++t) pool.emplace_back(work);
		work();
		for (std::thread &t : pool) t.join();
	}
	std::vector<size_t> first((size_t)n + 1, 0);
	for (int32_t i = 0; i < n; ++i) first[(size_t)i + 1] = first[(size_t)i] + plans[(size_t)i].ftl.size();
	const size_t total = first[(size_t)n];
	if (total > (size_t)INT32_MAX) { fprintf(stderr, "[libmwf_hip] mwf_wfa_chain_batch: more than 2^31 gap fills in one call\n"); abort(); }
	std::vector<int32_t> ftl(total), fql(total);
	std::vector<const char*> fts(total), fqs(total);
	for (int32_t i = 0; i < n; ++i) {
		const ChainPlan &P = plans[(size_t)i];
		std::copy(P.ftl.begin(), P.ftl.end(), ftl.begin() + (ptrdiff_t)first[(size_t)i]), std::copy(P.fql.begin(), P.fql.end(), fql.begin() + (ptrdiff_t)first[(size_t)i]);
		std::copy(P.fts.begin(), P.fts.end(), fts.begin() + (ptrdiff_t)first[(size_t)i]), std::copy(P.fqs.begin(), P.fqs.end(), fqs.begin() + (ptrdiff_t)first[(size_t)i]);
	}
	std::vector<mwf_rst_t> fills(total);
	if (total > 0) mwf_wfa_batch(nullptr, opt, (int32_t)total, ftl.data(), fts.data(), fql.data(), fqs.data(), fills.data());
	for (int32_t i = 0; i < n; ++i) stitch_chain(km, opt, plans[(size_t)i], fills.data() + first[(size_t)i], &r[i]);
	for (mwf_rst_t &f : fills) free(f.cigar);
}
