// host_format.cpp -- output text of a mapped batch, byte-identical to the reference's.
//
// Follows mp_write_output (format.c:453-473): mp_write_paf (format.c:333-358) with mp_write_cs (format.c:102-187), mp_write_gff
// (format.c:360-412), mp_write_gtf (format.c:414-451), the residue rows of --aln / --trans (format.c:189-331), and the per-query
// output filters and running hit id of worker_pipeline step 2 (map.c:298-311).
#include <cctype>
#include <cstdio>
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
	if (opt.flag & (MPA_MF_GFF | MPA_MF_GTF)) s += "##PAF\t";
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

// --aln / --trans (format.c:189-331): the alignment as four parallel rows -- target nucleotides (##ATN), their
// translation (##ATA), the match line (##AAS) and the query residues (##AQA) -- and the translated target alone (##STA).
// One text column per nucleotide; introns in lower case, abridged to max_intron_flank bases on either side when long.
static void put_residues(std::string &s, const mpa_idx_s *mi, const mpa_mapopt_t &opt, const char *aa, const mpa_hit_t &h, const uint32_t *cig)
{
	if (!h.has_aln) return;
	static const char UC[] = "ACGTN", LC[] = "acgtn";
	const char *i2c = "ARNDCQEGHILKMFPSTWYV*X";
	const uint8_t *aa20 = tab_aa20();
	const int32_t flank = opt.max_intron_flank;
	std::vector<uint8_t> ntv((size_t)(h.ve - h.vs + 3));
	const int64_t l_nt = fetch_nt(mi, (int32_t)h.vid, h.vs, h.ve + 3, ntv.data());   // three more bases when the contig has them
	const uint8_t *nt = ntv.data();
	std::string atn = "##ATN\t", ata = "##ATA\t", aas = "##AAS\t", aqa = "##AQA\t", sta = "##STA\t";
	auto col = [&](char a, char b, char c, char d) { atn.push_back(a), ata.push_back(b), aas.push_back(c), aqa.push_back(d); };
	auto match_char = [&](uint8_t t_aa, char q) {
		const uint8_t q_aa = aa20[(uint8_t)q];
		return t_aa == q_aa ? '|' : opt.mat[t_aa * opt.asize + q_aa] > 0 ? '+' : ' ';
	};
	int32_t nl = 0, al = h.qs;
	for (int32_t k = 0; k < h.n_cigar; ++k) {
		const int32_t op = cig[k] & 0xf, len = (int32_t)(cig[k] >> 4);
		if (op == 0) {                                       // M: codon against residue
			for (int32_t l = 0; l < len; ++l, nl += 3, ++al) {
				const uint8_t t = codon_aa3(nt + nl);
				sta.push_back(i2c[t]);
				col(UC[nt[nl]], i2c[t], match_char(t, aa[al]), (char)toupper(aa[al]));
				col(UC[nt[nl + 1]], '.', ' ', ' ');
				col(UC[nt[nl + 2]], '.', ' ', ' ');
			}
		} else if (op == 1) {                                // I: residues without target
			for (int32_t l = 0; l < len; ++l, ++al) col('-', '-', ' ', (char)toupper(aa[al])), col('-', '.', ' ', ' '), col('-', '.', ' ', ' ');
		} else if (op == 2) {                                // D: codons without residue
			for (int32_t l = 0; l < len; ++l, nl += 3) {
				const uint8_t t = codon_aa3(nt + nl);
				sta.push_back(i2c[t]);
				col(UC[nt[nl]], i2c[t], ' ', '-'), col(UC[nt[nl + 1]], '.', ' ', ' '), col(UC[nt[nl + 2]], '.', ' ', ' ');
			}
		} else if (op == 10) {                               // F: frameshift deletion
			for (int32_t l = 0; l < len; ++l, ++nl) col(UC[nt[nl]], '!', ' ', ' ');
		} else if (op == 11) {                               // G: frameshift match, one residue over `len` bases
			for (int32_t l = 0; l < len; ++l, ++nl) col(UC[nt[nl]], '$', ' ', l == 0 ? (char)toupper(aa[al]) : ' ');
			++al;
		} else if (op == 3 || op == 12 || op == 13) {        // N / U / V: intron in phase 0 / 1 / 2
			const int32_t intron = op == 3 ? len : len - 3;
			if (op != 3) {                                     // the split codon: 1 (U) or 2 (V) bases before the intron, the rest after
				const uint8_t c3[3] = { nt[nl], op == 12 ? nt[nl + len - 2] : nt[nl + 1], nt[nl + len - 1] };
				const uint8_t t = codon_aa3(c3);
				sta.push_back(i2c[t]);
				col(UC[nt[nl]], i2c[t], match_char(t, aa[al]), (char)toupper(aa[al]));
				++nl;
				if (op == 13) col(UC[nt[nl]], '.', ' ', ' '), ++nl;
				++al;
			}
			if (intron <= flank * 2) {
				for (int32_t l = 0; l < intron; ++l) col(LC[nt[nl + l]], ' ', ' ', ' ');
			} else {
				for (int32_t l = 0; l < flank; ++l) col(LC[nt[nl + l]], ' ', ' ', ' ');
				col('~', ' ', ' ', ' ');
				std::string num;
				put_int(num, intron);
				for (char ch : num) col(ch, ' ', ' ', ' ');
				col('~', ' ', ' ', ' ');
				for (int32_t l = 0; l < flank; ++l) col(LC[nt[nl + intron - flank + l]], ' ', ' ', ' ');
			}
			nl += intron;
			if (op != 3) {
				col(UC[nt[nl]], '.', ' ', ' '), ++nl;
				if (op == 12) col(UC[nt[nl]], '.', ' ', ' '), ++nl;
			}
		}
	}
	if (l_nt == h.ve - h.vs + 3 && sta.back() != '*') {       // one more codon: the stop, if the alignment ends right before it
		const uint8_t t = codon_aa3(nt + nl);
		sta.push_back(i2c[t]);
		col(UC[nt[nl]], i2c[t], ' ', ' '), col(UC[nt[nl + 1]], '.', ' ', ' '), col(UC[nt[nl + 2]], '.', ' ', ' ');
	}
	if (opt.flag & MPA_MF_SHOW_RESIDUE) s += atn, s += '\n', s += ata, s += '\n', s += aas, s += '\n', s += aqa, s += '\n';
	if (opt.flag & MPA_MF_SHOW_TRANS) s += sta, s += '\n';
}

static void put_id(std::string &s, const char *prefix, const char *mid, int64_t id)     // "%s<mid>%.6ld"
{
	char buf[32];
	snprintf(buf, sizeof(buf), "%.6ld", (long)id);
	s += prefix, s += mid, s += buf;
}

static void put_ratio(std::string &s, double x)                                          // "%.4f"
{
	char buf[32];
	snprintf(buf, sizeof(buf), "%.4f", x);
	s += buf;
}

// mp_write_gff (format.c:360-412)
static void put_gff(std::string &s, const mpa_idx_s *mi, const mpa_mapopt_t &opt, const char *name, int32_t qlen, const mpa_hit_t &h,
                    const mpa_feat_t *feat, int64_t id, int32_t hit_idx)
{
	if (!h.has_aln) return;
	const Contig &c = mi->ctg[h.vid >> 1];
	const bool rev = h.vid & 1, has_stop = h.qe == qlen && h.dist_stop == 0;
	const int64_t ve_mrna = has_stop ? h.ve + 3 : h.ve;
	std::string ids;
	if (opt.gff_delim >= 33 && opt.gff_delim <= 126 && hit_idx >= 0) ids = name, ids.push_back((char)opt.gff_delim), put_int(ids, hit_idx);
	else put_id(ids, opt.gff_prefix ? opt.gff_prefix : "MP", "", id);
	int64_t vs = rev ? c.len - ve_mrna : h.vs, ve = rev ? c.len - h.vs : ve_mrna;
	s += c.name, s += "\tminiprot\tmRNA\t", put_int(s, vs + 1), s.push_back('\t'), put_int(s, ve), s.push_back('\t'), put_int(s, h.dp_max);
	s.push_back('\t'), s.push_back("+-"[rev]), s += "\t.\tID=", s += ids, s += ";Rank=", put_int(s, hit_idx);
	s += ";Identity=", put_ratio(s, (double)h.n_iden * 3 / h.blen);
	s += ";Positive=", put_ratio(s, (double)h.n_plus * 3 / h.blen);
	if (h.n_fs > 0) s += ";Frameshift=", put_int(s, h.n_fs);
	if (h.n_stop > 0) s += ";StopCodon=", put_int(s, h.n_stop);
	s += ";Target=", s += name, s.push_back(' '), put_int(s, h.qs + 1), s.push_back(' '), put_int(s, h.qe), s.push_back('\n');
	for (int32_t j = 0; j < h.n_feat; ++j) {
		const mpa_feat_t &f = feat[j];
		int64_t fe = f.ve;
		if (has_stop && f.type == 0 && j + 1 < h.n_feat && feat[j + 1].type == 1) fe += 3;   // in GFF3 the last CDS includes the stop codon
		vs = rev ? c.len - fe : f.vs, ve = rev ? c.len - f.vs : fe;
		s += c.name, s += "\tminiprot\t", s += f.type == 1 ? "stop_codon" : "CDS";
		s.push_back('\t'), put_int(s, vs + 1), s.push_back('\t'), put_int(s, ve), s.push_back('\t'), put_int(s, f.score), s.push_back('\t'), s.push_back("+-"[rev]);
		s.push_back('\t'), put_int(s, f.phase), s += "\tParent=", s += ids, s += ";Rank=", put_int(s, hit_idx);
		if (f.type == 0) {
			s += ";Identity=", put_ratio(s, (double)f.n_iden * 3 / f.blen);
			if (f.acceptor[0] && strncmp(f.acceptor, "AG", 2) != 0) s += ";Acceptor=", s.push_back(f.acceptor[0]), s.push_back(f.acceptor[1]);
			if (f.donor[0] && strncmp(f.donor, "GT", 2) != 0) s += ";Donor=", s.push_back(f.donor[0]), s.push_back(f.donor[1]);
			if (f.n_fs > 0) s += ";Frameshift=", put_int(s, f.n_fs);
			if (f.n_stop > 0) s += ";StopCodon=", put_int(s, f.n_stop);
			s += ";Target=", s += name, s.push_back(' '), put_int(s, f.qs + 1), s.push_back(' '), put_int(s, f.qe);
		}
		s.push_back('\n');
	}
}

// mp_write_gtf (format.c:414-451)
static void put_gtf(std::string &s, const mpa_idx_s *mi, const mpa_mapopt_t &opt, int32_t qlen, const mpa_hit_t &h, const mpa_feat_t *feat, int64_t id)
{
	if (!h.has_aln) return;
	const Contig &c = mi->ctg[h.vid >> 1];
	const bool rev = h.vid & 1, has_stop = h.qe == qlen && h.dist_stop == 0;
	const int64_t ve_mrna = has_stop ? h.ve + 3 : h.ve;
	const char *prefix = opt.gff_prefix ? opt.gff_prefix : "MP";
	std::string gid, tid;
	put_id(gid, prefix, "G", id), put_id(tid, prefix, "T", id);
	const int64_t vs = rev ? c.len - ve_mrna : h.vs, ve = rev ? c.len - h.vs : ve_mrna;
	auto head = [&](const char *type, int64_t a, int64_t b, int32_t score) {
		s += c.name, s += "\tminiprot\t", s += type, s.push_back('\t'), put_int(s, a + 1), s.push_back('\t'), put_int(s, b), s.push_back('\t');
		put_int(s, score), s.push_back('\t'), s.push_back("+-"[rev]), s.push_back('\t');
	};
	head("gene", vs, ve, h.dp_max), s += ".\tgene_id \"", s += gid, s += "\";\n";
	head("transcript", vs, ve, h.dp_max), s += ".\ttranscript_id \"", s += tid, s += "\"; gene_id \"", s += gid, s += "\";\n";
	for (int32_t j = 0; j < h.n_feat; ++j) {
		const mpa_feat_t &f = feat[j];
		if (f.type != 0) continue;
		const int64_t a = rev ? c.len - f.ve : f.vs, b = rev ? c.len - f.vs : f.ve;
		int64_t a2 = a, b2 = b;
		if (f.ve == h.ve) { if (rev) a2 = c.len - ve_mrna; else b2 = ve_mrna; }           // the last exon also covers the stop codon
		head("exon", a2, b2, f.score), s += ".\ttranscript_id \"", s += tid, s += "\"; gene_id \"", s += gid, s += "\";\n";
		head("CDS", a, b, f.score), put_int(s, f.phase), s += "\ttranscript_id \"", s += tid, s += "\"; gene_id \"", s += gid, s += "\";\n";
	}
}

} // namespace mpa

using namespace mpa;

// mp_write_output (format.c:453-473) applied to a whole batch, with the output filters and the running hit id of
// worker_pipeline step 2 (map.c:298-311).
static int64_t format_batch(const mpa_idx_t *mi, const mpa_mapopt_t *opt, const mpa_qbatch_t *q, const char *const *names,
                            const mpa_result_t *r, int64_t *id_io, char **out)
{
	double t0 = now_ms();
	const mpa_hit_t *hits = mpa_result_hits(r);
	const int64_t *off = mpa_result_hit_off(r);
	const uint32_t *cig = mpa_result_cigars(r);
	const mpa_feat_t *feats = mpa_result_feats(r);
	const int32_t n_seq = q->n_seq;
	// which hits pass the filters: needed first, because the GFF/GTF ids count printed hits across queries
	auto passes = [&](int32_t i, int64_t j, int32_t best_sc) {
		const mpa_hit_t &h = hits[off[i] + j];
		const int32_t sc = h.has_aln ? h.dp_max : h.chn_sc;
		const int32_t qlen = (int32_t)(q->q_off[i + 1] - q->q_off[i]);
		if (sc <= 0 || sc < (double)best_sc * opt->out_sim) return false;
		if (h.qe - h.qs < (double)qlen * opt->out_cov) return false;
		return true;
	};
	std::vector<int64_t> id_base((size_t)n_seq + 1, 0);
	for (int32_t i = 0; i < n_seq; ++i) {
		const int64_t n_reg = off[i + 1] - off[i];
		int32_t best_sc = -1;
		int64_t n_out = 0;
		if (n_reg > 0) best_sc = hits[off[i]].has_aln ? hits[off[i]].dp_max : hits[off[i]].chn_sc;
		for (int64_t j = 0; j < n_reg && j < opt->out_n; ++j) n_out += passes(i, j, best_sc);
		id_base[i + 1] = id_base[i] + n_out;
	}
	const int64_t id0 = id_io ? *id_io : 0;
	std::vector<std::string> part((size_t)n_seq);
	auto one = [&](int32_t i) {
		std::string &s = part[i];
		const char *seq = q->seqs + q->q_off[i];
		const int32_t qlen = (int32_t)(q->q_off[i + 1] - q->q_off[i]);
		const int64_t n_reg = off[i + 1] - off[i];
		const mpa_hit_t *h = hits + off[i];
		int32_t best_sc = -1;
		int64_t id = id0 + id_base[i];
		if (n_reg > 0) best_sc = h[0].has_aln ? h[0].dp_max : h[0].chn_sc;
		for (int64_t j = 0; j < n_reg && j < opt->out_n; ++j) {
			if (!passes(i, j, best_sc)) continue;
			++id;
			const bool residues = (opt->flag & (MPA_MF_SHOW_RESIDUE | MPA_MF_SHOW_TRANS)) != 0;
			if (opt->flag & MPA_MF_GTF) {
				if (residues) put_paf(s, mi, *opt, names[i], seq, qlen, &h[j], cig), put_residues(s, mi, *opt, seq, h[j], cig + h[j].cigar_off);
				put_gtf(s, mi, *opt, qlen, h[j], feats + h[j].feat_off, id);
			} else {
				if (!(opt->flag & MPA_MF_NO_PAF)) put_paf(s, mi, *opt, names[i], seq, qlen, &h[j], cig);
				if (residues) put_residues(s, mi, *opt, seq, h[j], cig + h[j].cigar_off);
				if (opt->flag & MPA_MF_GFF) put_gff(s, mi, *opt, names[i], qlen, h[j], feats + h[j].feat_off, id, (int32_t)j + 1);
			}
		}
		if (id_base[i + 1] == id_base[i] && (opt->flag & MPA_MF_SHOW_UNMAP)) put_paf(s, mi, *opt, names[i], seq, qlen, nullptr, cig);
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
	if (id_io) *id_io = id0 + id_base[n_seq];
	size_t total = 0;
	for (const std::string &x : part) total += x.size();
	std::string s;
	s.reserve(total);
	for (const std::string &x : part) s += x;
	timing_note("format output", now_ms() - t0);
	char *buf = (char*)malloc(s.size() + 1);
	memcpy(buf, s.data(), s.size());
	buf[s.size()] = 0;
	*out = buf;
	return (int64_t)s.size();
}

// how many hit records mpa_format_output() prints for this batch = how far it advances the running hit id (map.c:298-306)
extern "C" int64_t mpa_result_n_output(const mpa_mapopt_t *opt, const mpa_qbatch_t *q, const mpa_result_t *r)
{
	const mpa_hit_t *hits = mpa_result_hits(r);
	const int64_t *off = mpa_result_hit_off(r);
	int64_t n = 0;
	for (int32_t i = 0; i < q->n_seq; ++i) {
		const int64_t n_reg = off[i + 1] - off[i];
		if (n_reg == 0) continue;
		const int32_t best_sc = hits[off[i]].has_aln ? hits[off[i]].dp_max : hits[off[i]].chn_sc;
		const int32_t qlen = (int32_t)(q->q_off[i + 1] - q->q_off[i]);
		for (int64_t j = 0; j < n_reg && j < opt->out_n; ++j) {
			const mpa_hit_t &h = hits[off[i] + j];
			const int32_t sc = h.has_aln ? h.dp_max : h.chn_sc;
			if (sc <= 0 || sc < (double)best_sc * opt->out_sim) continue;
			if (h.qe - h.qs < (double)qlen * opt->out_cov) continue;
			++n;
		}
	}
	return n;
}

extern "C" int64_t mpa_format_paf(const mpa_idx_t *mi, const mpa_mapopt_t *opt, const mpa_qbatch_t *q, const char *const *names,
                                  const mpa_result_t *r, char **out)
{
	mpa_mapopt_t o = *opt;
	o.flag &= ~(uint32_t)(MPA_MF_GFF | MPA_MF_GTF | MPA_MF_NO_PAF);
	return format_batch(mi, &o, q, names, r, nullptr, out);
}

static int64_t mpa_format_output_impl(const mpa_idx_t *mi, const mpa_mapopt_t *opt, const mpa_qbatch_t *q, const char *const *names,
                                     const mpa_result_t *r, int64_t *id_io, char **out)
{
	return format_batch(mi, opt, q, names, r, id_io, out);
}

extern "C" int64_t mpa_format_output(const mpa_idx_t *mi, const mpa_mapopt_t *opt, const mpa_qbatch_t *q, const char *const *names,
                                     const mpa_result_t *r, int64_t *id_io, char **out)
{
	return mpa::guarded<int64_t>((int64_t)MPA_ERR_HIP, [&] { return mpa_format_output_impl(mi, opt, q, names, r, id_io, out); });
}
