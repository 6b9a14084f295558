// test-mwf — command-line front end with the reference's options and output format (reference main.c:19-92),
// on top of libmwf_hip.so.
//
//   test-mwf [-c] [-p INT] [-u] [-t] [-l INT] [-f INT] [-a] [-e] [-K] [-d] <in1.fa> <in2.fa>
//
// Reads the two FASTA/FASTQ files in lock-step (record i of file 1 is aligned to record i of file 2, reference
// main.c:67) and prints, per pair, the reference's PAF-like line
//     name1  len1  0  len1  +  name2  len2  0  len2  penalty  [CIGAR]
// Differences from the reference, all on the input side: in exact mode the pairs of the whole file are aligned as ONE
// device batch (mwf_wfa_batch) instead of one call per pair; plain and gzip input are both accepted when built with
// -DMWF_HAVE_ZLIB -lz, plain only otherwise.  The "T" timing lines on stderr report wall time of the batch.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "miniwfa.h"
#ifdef MWF_HAVE_ZLIB
#include <zlib.h>
#endif

namespace {

struct Record { std::string name, seq; };

// Minimal FASTA/FASTQ reader (multi-line sequences, '>' and '@' headers, '+' quality blocks skipped by length).
class SeqReader {
public:
	explicit SeqReader(const char *path)
	{
#ifdef MWF_HAVE_ZLIB
		gz_ = gzopen(path, "r");
		ok_ = gz_ != nullptr;
#else
		fp_ = strcmp(path, "-") ? fopen(path, "r") : stdin;
		ok_ = fp_ != nullptr;
#endif
	}
	~SeqReader()
	{
#ifdef MWF_HAVE_ZLIB
		if (gz_) gzclose(gz_);
#else
		if (fp_ && fp_ != stdin) fclose(fp_);
#endif
	}
	bool ok() const { return ok_; }
	bool next(Record &r)
	{
		std::string line;
		if (pending_.empty()) {
			while (getline(line))
				if (!line.empty() && (line[0] == '>' || line[0] == '@')) { pending_ = line; break; }
			if (pending_.empty()) return false;
		}
		const bool fastq = pending_[0] == '@';
		size_t e = 1;
		while (e < pending_.size() && pending_[e] != ' ' && pending_[e] != '\t') ++e;
		r.name = pending_.substr(1, e - 1);
		r.seq.clear();
		pending_.clear();
		while (getline(line)) {
			if (!line.empty() && (line[0] == '>' || (line[0] == '@' && !fastq))) { pending_ = line; break; }
			if (fastq && !line.empty() && line[0] == '+') { // quality: as many characters as bases
				size_t q = 0;
				while (q < r.seq.size() && getline(line)) q += line.size();
				break;
			}
			for (char c : line)
				if (c != ' ' && c != '\t') r.seq.push_back(c);
		}
		return true;
	}

private:
	bool getline(std::string &out)
	{
		out.clear();
		char buf[65536];
		for (;;) {
#ifdef MWF_HAVE_ZLIB
			if (!gzgets(gz_, buf, sizeof(buf))) return !out.empty();
#else
			if (!fgets(buf, sizeof(buf), fp_)) return !out.empty();
#endif
			size_t n = strlen(buf);
			const bool eol = n && buf[n - 1] == '\n';
			while (n && (buf[n - 1] == '\n' || buf[n - 1] == '\r')) --n;
			out.append(buf, n);
			if (eol) return true;
		}
	}
#ifdef MWF_HAVE_ZLIB
	gzFile gz_ = nullptr;
#else
	FILE *fp_ = nullptr;
#endif
	bool ok_ = false;
	std::string pending_;
};

void usage(const mwf_opt_t &opt)
{
	fprintf(stderr, "Usage: test-mwf [options] <in1.fa> <in2.fa>\n");
	fprintf(stderr, "Options:\n");
	fprintf(stderr, "  -c       generate CIGAR\n");
	fprintf(stderr, "  -p INT   step size (force -c; 0 to disable) [%d]\n", opt.step);
	fprintf(stderr, "  -u       apply the chaining heuristic\n");
	fprintf(stderr, "  -t       automatically choose between the exact and the chaining mode\n");
	fprintf(stderr, "  -l INT   min gapless length for chain filtering [%d]\n", opt.min_len);
	fprintf(stderr, "  -f INT   max k-mer occurrence [%d]\n", opt.max_occ);
	fprintf(stderr, "  -a       mimic affine gap\n");
	fprintf(stderr, "  -e       mimic edit distance\n");
	fprintf(stderr, "  -K       accepted for compatibility (scratch memory lives on the device)\n");
}

} // namespace

int main(int argc, char *argv[])
{
	mwf_opt_t opt;
	int mode = 0, i = 1;
	mwf_opt_init(&opt);
	for (; i < argc && argv[i][0] == '-' && argv[i][1]; ++i) { // the reference's option letters (main.c:29-44), clustered or not
		for (const char *p = argv[i] + 1; *p; ++p) {
			auto arg = [&]() -> const char * {
				if (p[1]) { const char *a = p + 1; p += strlen(p) - 1; return a; }
				if (i + 1 < argc) return argv[++i];
				fprintf(stderr, "ERROR: option -%c needs an argument\n", *p);
				exit(1);
			};
			switch (*p) {
			case 'K': opt.flag |= MWF_F_NO_KALLOC; break;
			case 'c': opt.flag |= MWF_F_CIGAR; break;
			case 'd': opt.flag |= MWF_F_DEBUG; break;
			case 'p': opt.flag |= MWF_F_CIGAR, opt.step = atoi(arg()); break;
			case 'a': opt.o2 = opt.o1, opt.e2 = opt.e1; break;
			case 'e': opt.x = 1, opt.o1 = opt.o2 = 0, opt.e1 = opt.e2 = 1; break;
			case 'l': opt.min_len = atoi(arg()); break;
			case 'f': opt.max_occ = atoi(arg()); break;
			case 'u': mode = 1; break;
			case 't': mode = 2; break;
			default: fprintf(stderr, "ERROR: unknown option\n"); return 1;
			}
		}
	}
	if (argc - i < 2) { usage(opt); return 1; }
	SeqReader r1(argv[i]), r2(argv[i + 1]);
	if (!r1.ok() || !r2.ok()) { fprintf(stderr, "ERROR: cannot open the input files\n"); return 1; }
	std::vector<Record> a, b;
	for (Record x, y; r1.next(x) && r2.next(y);) a.push_back(x), b.push_back(y);
	const int32_t n = (int32_t)a.size();
	if (getenv("MWF_CLI_PARSE_ONLY")) { // reader self-test: no device needed
		for (int32_t k = 0; k < n; ++k) printf("%s\t%zu\t%s\t%zu\n", a[k].name.c_str(), a[k].seq.size(), b[k].name.c_str(), b[k].seq.size());
		return 0;
	}
	std::vector<mwf_rst_t> rst(n);
	std::vector<int32_t> tl(n), ql(n);
	std::vector<const char*> ts(n), qs(n);
	for (int32_t k = 0; k < n; ++k) tl[k] = (int32_t)a[k].seq.size(), ql[k] = (int32_t)b[k].seq.size(), ts[k] = a[k].seq.data(), qs[k] = b[k].seq.data();
	const auto t0 = std::chrono::steady_clock::now();
	if (mode == 0) mwf_wfa_batch(nullptr, &opt, n, tl.data(), ts.data(), ql.data(), qs.data(), rst.data());
	else if (mode == 1) { // chain mode over every record: their gap fills in one device batch
		for (int32_t k = 0; k < n; ++k) memset(&rst[k], 0, sizeof(mwf_rst_t));
		mwf_wfa_chain_batch(nullptr, &opt, n, tl.data(), ts.data(), ql.data(), qs.data(), rst.data());
	} else { // auto: the exact branch of every record as one batch, chain mode for those it gives up on
		for (int32_t k = 0; k < n; ++k) memset(&rst[k], 0, sizeof(mwf_rst_t));
		mwf_wfa_auto_batch(nullptr, &opt, n, tl.data(), ts.data(), ql.data(), qs.data(), rst.data());
	}
	const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	for (int32_t k = 0; k < n; ++k) {
		if (opt.flag & MWF_F_CIGAR) mwf_assert_cigar(&opt, rst[k].n_cigar, rst[k].cigar, tl[k], ql[k], rst[k].s);
		printf("%s\t%d\t0\t%d\t+\t%s\t%d\t0\t%d\t%d", a[k].name.c_str(), tl[k], tl[k], b[k].name.c_str(), ql[k], ql[k], rst[k].s);
		if (opt.flag & MWF_F_CIGAR) {
			putchar('\t');
			for (int32_t j = 0; j < rst[k].n_cigar; ++j) printf("%d%c", rst[k].cigar[j] >> 4, "MIDNSHP=XBid"[rst[k].cigar[j] & 0xf]);
		}
		putchar('\n');
		free(rst[k].cigar);
		fprintf(stderr, "T\t%s\t%s\t%.3f\n", a[k].name.c_str(), b[k].name.c_str(), sec / (n ? n : 1));
	}
	return 0;
}
