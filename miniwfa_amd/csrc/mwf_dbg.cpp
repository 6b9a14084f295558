// mwf_dbg.cpp — CIGAR self-checks of the drop-in ABI (reference mwf-dbg.c:6-31).
#include <cstdio>
#include <cstdlib>
#include "miniwfa.h"

extern "C" {

// Penalty and consumed lengths implied by a CIGAR: '='/'X'/'M' advance both sequences, 'I' the
// query, 'D' the target; an indel run of length L costs min(o1+L*e1, o2+L*e2), 'X' costs x per base.
int32_t mwf_cigar2score(const mwf_opt_t *opt, int32_t n_cigar, const uint32_t *cigar, int32_t *tl, int32_t *ql)
{
	int64_t score = 0;
	int32_t on_t = 0, on_q = 0;
	for (int32_t i = 0; i < n_cigar; ++i) {
		const int32_t op = (int32_t)(cigar[i] & 0xf), len = (int32_t)(cigar[i] >> 4);
		switch (op) {
		case 1: case 2: {
			const int64_t p1 = opt->o1 + (int64_t)len * opt->e1, p2 = opt->o2 + (int64_t)len * opt->e2;
			score += p1 < p2 ? p1 : p2;
			if (op == 1) on_q += len; else on_t += len;
			break;
		}
		case 8: score += (int64_t)len * opt->x; /* fall through */
		case 0: case 7: on_t += len, on_q += len; break;
		default: break;
		}
	}
	if (tl) *tl = on_t;
	if (ql) *ql = on_q;
	return (int32_t)score;
}

// Lengths must match exactly (hard failure, like the reference's assert); a CIGAR that costs more
// than the reported penalty only draws a warning (reference mwf-dbg.c:30).
void mwf_assert_cigar(const mwf_opt_t *opt, int32_t n_cigar, const uint32_t *cigar, int32_t tl0, int32_t ql0, int32_t s0)
{
	int32_t tl = 0, ql = 0;
	const int32_t s = mwf_cigar2score(opt, n_cigar, cigar, &tl, &ql);
	if (tl != tl0 || ql != ql0) {
		fprintf(stderr, "[mwf_assert_cigar] CIGAR consumes (%d,%d) bases, sequences are (%d,%d)\n", tl, ql, tl0, ql0);
		abort();
	}
	if (s > s0) fprintf(stderr, "[mwf_assert_cigar] s0=%d, s=%d\n", s0, s);
}

} // extern "C"
