// host_format.cpp -- PAF text of a mapped batch, byte-identical to the reference's.
//
// Follows mp_write_paf (format.c:333-358), mp_write_cs (format.c:102-187) and the per-query output
// filters of worker_pipeline step 2 (map.c:298-311).  GFF3/GTF/--aln/--trans are not part of this round.
#include <cctype>
#include <cstring>
#include <string>
#include <thread>
#include <atomic>
#include <vector>
#include "host_core.h"

namespace mpa {

static void put_int(std::string &s, int64_t v)
{
	char buf[24];
	int n = 0;
	uint64_t x = v < 0 ? (uint64_t)(-v) : (uint64_t)v;
	do buf[n++] = (char)('0' + x % 10), x /= 10; while (x);
	if (v < 0) buf[n++] = '-';
	while (n) s.push_back(buf[--n]);
}

static inline uint8_t codon_aa3(const uint8_t *nt)
{
	return nt[0] > 3 || nt[1] > 3 || nt[2] > 3 ? 21 : tab_codon()[nt[0] << 4 | nt[1] << 2 | nt[2]];
}

// the cs:Z: difference string (format.c:102-187)
static void put_cs(std::string &s, const mpa_idx_s *mi, const char *aa, const mpa_hit_t &h, const uint32_t *cig)
{
	static const char lc[] = "acgtn";
	const uint8_t *aa20 = tab_aa20();
	std::vector<uint8_t> ntv((size_t)std::max<int64_t>(h.ve - h.vs, 1));
	fetch_nt(mi, (int32_t)h.vid, h.vs, h.ve, ntv.data());
	const uint8_t *nt = ntv.data();
	int32_t nl = 0, al = 0;
	s += "cs:Z:";
	for (int32_t k = 0; k < h.n_cigar; ++k) {
		const int32_t op = cig[k] & 0xf, len = (int32_t)(cig[k] >> 4);
		if (op == 0) {
			int32_t run = 0;
			for (int32_t l = 0; l < len; ++l) {
				const uint8_t *c = nt + nl + 3 * l;
				if (codon_aa3(c) != aa20[(uint8_t)aa[al + l]]) {
					if (run > 0) s.push_back(':'), put_int(s, run);
					s.push_back('*'), s.push_back(lc[c[0]]), s.push_back(lc[c[1]]), s.push_back(lc[c[2]]), s.push_back((char)toupper(aa[al + l]));
					run = 0;
				} else ++run;
			}
			if (run > 0) s.push_back(':'), put_int(s, run);
			nl += len * 3, al += len;
		} else if (op == 1) {
			s.push_back('+');
			for (int32_t j = 0; j < len; ++j) s.push_back((char)toupper(aa[al + j]));
			al += len;
		} else if (op == 2 || op == 10) {
			const int32_t n = op == 2 ? len * 3 : len;
			s.push_back('-');
			for (int32_t i = 0; i < n; ++i) s.push_back(lc[nt[nl + i]]);
			nl += n;
		} else if (op == 11) {
			s.push_back('*');
			for (int32_t i = 0; i < len; ++i) s.push_back(lc[nt[nl + i]]);
			s.push_back((char)toupper(aa[al]));
			nl += len, ++al;
		} else if (op == 3 || op == 12 || op == 13) {
			const int32_t lshift = op == 3 ? 0 : op == 12 ? 1 : 2, rshift = lshift == 0 ? 0 : 3 - lshift;
			if (lshift > 0) {
				s.push_back('*');
				for (int32_t i = 0; i < lshift; ++i) s.push_back(lc[nt[nl + i]]);
				s.push_back((char)toupper(aa[al]));
			}
			s.push_back('~'), s.push_back(lc[nt[nl + lshift]]), s.push_back(lc[nt[nl + lshift + 1]]);
			put_int(s, len - (lshift + rshift));
			s.push_back(lc[nt[nl + len - rshift - 2]]), s.push_back(lc[nt[nl + len - rshift - 1]]);
			if (rshift > 0) {
				s.push_back('-');
				for (int32_t i = 0; i < rshift; ++i) s.push_back(lc[nt[nl + len - rshift + i]]);
			}
			if (lshift) ++al;
			nl += len;
		}
	}
}

static void put_paf(std::string &s, const mpa_idx_s *mi, const mpa_mapopt_t &opt, const char *name, const char *seq, int32_t qlen,
                    const mpa_hit_t *h, const uint32_t *cig_pool)
{
	static const char ops[] = "MIDNSHP=XBFGUVE";
	s += name, s.push_back('\t'), put_int(s, qlen);
	if (!h) { s += "\t0\t0\t*\t*\t0\t0\t0\t0\t0\t0\n"; return; }
	const Contig &c = mi->ctg[h->vid >> 1];
	s.push_back('\t'), put_int(s, h->qs), s.push_back('\t'), put_int(s, h->qe), s.push_back('\t'), s.push_back("+-"[h->vid & 1]);
	s.push_back('\t'), s += c.name, s.push_back('\t'), put_int(s, c.len), s.push_back('\t');
	if (h->vid & 1) put_int(s, c.len - h->ve), s.push_back('\t'), put_int(s, c.len - h->vs);
	else put_int(s, h->vs), s.push_back('\t'), put_int(s, h->ve);
	s.push_back('\t');
	if (h->has_aln) {
		const uint32_t *cig = cig_pool + h->cigar_off;
		put_int(s, h->n_iden * 3), s.push_back('\t'), put_int(s, h->blen), s += "\t0\tAS:i:", put_int(s, h->dp_score);
		s += "\tms:i:", put_int(s, h->dp_max), s += "\tnp:i:", put_int(s, h->n_plus), s += "\tfs:i:", put_int(s, h->n_fs);
		s += "\tst:i:", put_int(s, h->n_stop), s += "\tda:i:", put_int(s, h->dist_start), s += "\tdo:i:", put_int(s, h->dist_stop);
		s += "\tcg:Z:";
		for (int32_t k = 0; k < h->n_cigar; ++k) put_int(s, cig[k] >> 4), s.push_back(ops[cig[k] & 0xf]);
		if (!(opt.flag & MPA_MF_NO_CS)) s.push_back('\t'), put_cs(s, mi, seq + h->qs, *h, cig);
	} else {
		put_int(s, h->chn_sc), s.push_back('\t'), put_int(s, h->chn_sc_ungap), s.push_back('\t'), put_int(s, h->cnt);
		if (!(opt.flag & MPA_MF_NO_CS)) s.push_back('\t');     // mp_write_cs() returns at once when there is no alignment
	}
	s.push_back('\n');
}

} // namespace mpa

using namespace mpa;

extern "C" int64_t mpa_format_paf(const mpa_idx_t *mi, const mpa_mapopt_t *opt, const mpa_qbatch_t *q, const char *const *names,
                                  const mpa_result_t *r, char **out)
{
	double t0 = now_ms();
	const mpa_hit_t *hits = mpa_result_hits(r);
	const int64_t *off = mpa_result_hit_off(r);
	const uint32_t *cig = mpa_result_cigars(r);
	// every query's lines are independent: format in parallel, concatenate in input order (map.c:293-313)
	const int32_t n_seq = q->n_seq;
	std::vector<std::string> part((size_t)n_seq);
	auto one = [&](int32_t i) {
		std::string &s = part[i];
		const char *seq = q->seqs + q->q_off[i];
		const int32_t qlen = (int32_t)(q->q_off[i + 1] - q->q_off[i]);
		const int64_t n_reg = off[i + 1] - off[i];
		const mpa_hit_t *h = hits + off[i];
		int32_t best_sc = -1, n_out = 0;
		if (n_reg > 0) best_sc = h[0].has_aln ? h[0].dp_max : h[0].chn_sc;
		for (int64_t j = 0; j < n_reg && j < opt->out_n; ++j) {
			const int32_t sc = h[j].has_aln ? h[j].dp_max : h[j].chn_sc;
			if (sc <= 0 || sc < (double)best_sc * opt->out_sim) continue;
			if (h[j].qe - h[j].qs < (double)qlen * opt->out_cov) continue;
			put_paf(s, mi, *opt, names[i], seq, qlen, &h[j], cig);
			++n_out;
		}
		if (n_out == 0 && (opt->flag & MPA_MF_SHOW_UNMAP)) put_paf(s, mi, *opt, names[i], seq, qlen, nullptr, cig);
	};
	{
		unsigned hw = std::thread::hardware_concurrency();
		const int nt = (int)std::max(1u, std::min(16u, hw / 4 ? hw / 4 : 1u));
		std::atomic<int32_t> next(0);
		auto work = [&]() { for (;;) { int32_t i = next.fetch_add(1); if (i >= n_seq) break; one(i); } };
		std::vector<std::thread> pool;
		for (int t = 1; t < nt && t < n_seq; ++t) pool.emplace_back(work);
		work();
		for (auto &th : pool) th.join();
	}
	size_t total = 0;
	for (const std::string &x : part) total += x.size();
	std::string s;
	s.reserve(total);
	for (const std::string &x : part) s += x;
	timing_note("format_paf", now_ms() - t0);
	char *buf = (char*)malloc(s.size() + 1);
	memcpy(buf, s.data(), s.size());
	buf[s.size()] = 0;
	*out = buf;
	return (int64_t)s.size();
}
