// index.cpp -- the .mpi index on the host: load, dump, window fetch.
//
// Replaces index.c:11-44 (block offsets, block->contig lookup), index.c:189-229 (dump/restore) and
// ntseq.c:89-114,163-205 (4-bit genome store) of the reference.  File layout:
//   "MPI\3" | idxopt (20 B) | int64 n_kb | int32 n_ctg | int32 l_name | int64 l_seq | int64 len[n_ctg]
//   | uint8 seq[(l_seq+1)/2] | char name[l_name] | int64 ki[1<<(4k-m)] | uint32 kb[n_kb]
#include <sys/mman.h>
#include <sys/stat.h>
#include <fcntl.h>
#include <unistd.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <atomic>
#include <thread>
#include <zlib.h>
#include <unordered_map>
#include "mpa_internal.h"
#include "host_core.h"
#include "fasta.h"

namespace mpa {

static const char kMagic[4] = { 'M', 'P', 'I', 3 };

static void compute_block_offsets(mpa_idx_s *mi)
{
	const int32_t bbit = mi->opt.bbit;
	const size_t n = mi->ctg.size();
	int64_t acc = 0;
	mi->bo.assign(2 * n + 1, 0);
	for (size_t i = 0; i < n; ++i) {
		int64_t nb = (mi->ctg[i].len + (1LL << bbit) - 1) >> bbit;
		mi->bo[2 * i] = (uint32_t)acc, acc += nb;      // forward strand
		mi->bo[2 * i + 1] = (uint32_t)acc, acc += nb;  // reverse strand
	}
	mi->bo[2 * n] = mi->n_block = (uint32_t)acc;
}

int32_t block2vid(const mpa_idx_s *mi, uint32_t blk)
{
	// largest v with bo[v] <= blk; -1 when blk is past the last block
	const int32_t n = (int32_t)mi->bo.size() - 1;
	if (n <= 0 || blk >= mi->bo[n]) return -1;
	int32_t lo = 0, hi = n - 1;
	while (lo < hi) {
		int32_t mid = (lo + hi + 1) >> 1;
		if (mi->bo[mid] <= blk) lo = mid; else hi = mid - 1;
	}
	// contigs shorter than a block cannot exist (len>0 gives >=1 block), but strands of equal offset could
	// only arise for len==0; walk forward to the interval that really contains blk
	while (lo + 1 < n && mi->bo[lo + 1] <= blk) ++lo;
	return lo;
}

int64_t fetch_nt(const mpa_idx_s *mi, int32_t vid, int64_t st, int64_t en, uint8_t *out)
{
	const int32_t cid = vid >> 1;
	if (cid < 0 || cid >= (int32_t)mi->ctg.size()) return -1;
	const Contig &c = mi->ctg[cid];
	if (st < 0 || en < 0 || st >= c.len) return -1;
	if (en > c.len) en = c.len;
	// two bases per packed byte through a 256-entry table (forward: low nibble first; reverse strand: complemented,
	// high nibble first); a nibble above 3 is an ambiguous base and stays what it is
	static const struct Tab {
		uint16_t fwd[256], rev[256];
		Tab() {
			for (int b = 0; b < 256; ++b) {
				const uint8_t lo = b & 0xf, hi = b >> 4;
				const uint8_t clo = lo >= 4 ? lo : 3 - lo, chi = hi >= 4 ? hi : 3 - hi;
				fwd[b] = (uint16_t)(lo | hi << 8), rev[b] = (uint16_t)(chi | clo << 8);
			}
		}
	} tab;
	int64_t k = 0;
	if (!(vid & 1)) {
		int64_t p = c.off + st;
		const int64_t pe = c.off + en;
		if (p < pe && (p & 1)) out[k++] = nt_at(mi, p++);
		for (; p + 2 <= pe; p += 2, k += 2) memcpy(out + k, &tab.fwd[mi->seq[p >> 1]], 2);
		if (p < pe) out[k++] = nt_at(mi, p);
	} else {                                          // reverse strand: walk the forward strand backwards, complement
		int64_t p = c.off + c.len - 1 - st;           // first base to emit, going down to pe (inclusive)
		const int64_t pe = c.off + c.len - en;
		auto one = [&](int64_t q) { const uint8_t b = nt_at(mi, q); out[k++] = b >= 4 ? b : (uint8_t)(3 - b); };
		if (p >= pe && !(p & 1)) one(p--);            // p even: its byte's high nibble belongs to p+1, emit alone
		for (; p - 1 >= pe; p -= 2, k += 2) memcpy(out + k, &tab.rev[mi->seq[p >> 1]], 2);   // p odd: byte holds (p-1, p)
		if (p >= pe) one(p);
	}
	return k;
}

template<typename T> static bool rd(FILE *fp, T *dst, size_t n) { return fread(dst, sizeof(T), n, fp) == n; }

} // namespace mpa

using namespace mpa;

extern "C" {

// The file is MAPPED, not read: the packed genome and the occurrence lists (1.5 + 6 GB at 3 Gbp) are views into a private
// read-only mapping -- nothing is copied, the pages come from the page cache, and the ranks of a multi-GPU job that restore
// the same file share them (index.c:204-237 reads the file into malloc'd arrays).  Contig table, names and the bucket offsets
// (64 MB, which the host stages index as aligned 64-bit integers) are copied out.
mpa_idx_t *mpa_idx_restore(const char *fn)
{
	const int fd = open(fn, O_RDONLY);
	if (fd < 0) { set_error(std::string("cannot open ") + fn); return nullptr; }
	struct stat st;
	if (fstat(fd, &st) != 0 || st.st_size < 44) { close(fd); set_error(std::string(fn) + " is not a valid .mpi index"); return nullptr; }
	const size_t flen = (size_t)st.st_size;
	// (MAP_PRIVATE + PROT_WRITE: callers of the reference's mp_idx_t get non-const pointers; a write would copy the page, never reach
	// the file.  MAP_NORESERVE: nobody is expected to write, so the mapping must not be charged against the commit limit once per
	// rank.  Should the mapping fail -- a file system without mmap, an address-space limit -- the file is read into memory instead.)
	void *base = mmap(nullptr, flen, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_NORESERVE, fd, 0);
	bool mapped = base != MAP_FAILED;
	std::vector<char> *owned = nullptr;
	if (!mapped) {
		owned = new (std::nothrow) std::vector<char>();
		bool got = owned != nullptr;
		if (got) try { owned->resize(flen); } catch (const std::exception&) { got = false; }
		size_t done = 0;
		while (got && done < flen) { const ssize_t k = pread(fd, owned->data() + done, flen - done, (off_t)done); if (k <= 0) got = false; else done += (size_t)k; }
		if (!got) { delete owned; close(fd); set_error(std::string("cannot map or read ") + fn); return nullptr; }
		base = owned->data();
	}
	close(fd);
	const char *f = (const char*)base;
	size_t at = 0;
	auto take = [&](void *dst, size_t bytes) -> bool { if (bytes > flen - at) return false; if (dst) memcpy(dst, f + at, bytes); at += bytes; return true; };
	mpa_idx_s *mi = new mpa_idx_s();
	char magic[4];
	int32_t hdr[2];
	bool ok = take(magic, 4) && memcmp(magic, kMagic, 4) == 0;
	ok = ok && take(&mi->opt, sizeof(mi->opt)) && take(&mi->n_kb, 8) && take(hdr, 8) && take(&mi->l_seq, 8);
	if (ok && set_trans_code((int)mi->opt.trans_code) < 0) ok = false;
	// the header sizes come from a file: refuse nonsense before it turns into an allocation (and an exception across the C ABI)
	if (ok) ok = hdr[0] >= 0 && hdr[1] >= 0 && mi->l_seq >= 0 && mi->n_kb >= 0 && mi->opt.kmer >= 1 && mi->opt.kmer <= 7 &&
	             mi->opt.mod_bit >= 0 && mi->opt.mod_bit < 4 * mi->opt.kmer && mi->opt.bbit >= 0 && mi->opt.bbit < 31 &&
	             (int64_t)hdr[0] * 8 <= (int64_t)1 << 40 && mi->l_seq <= (int64_t)1 << 40 && mi->n_kb <= (int64_t)1 << 40;
	if (ok) try {
		const int32_t n_ctg = hdr[0], l_name = hdr[1];
		mi->ctg.resize(n_ctg);
		int64_t off = 0;
		for (int32_t i = 0; i < n_ctg && ok; ++i) {
			ok = take(&mi->ctg[i].len, 8);
			mi->ctg[i].off = off, off += mi->ctg[i].len;
		}
		const size_t seq_bytes = (size_t)((mi->l_seq + 1) >> 1), seq_at = at;
		ok = ok && take(nullptr, seq_bytes);
		mi->names.resize(l_name);
		ok = ok && take(mi->names.data(), mi->names.size());
		const char *p = mi->names.data(), *end = p + mi->names.size();
		for (int32_t i = 0; i < n_ctg && ok; ++i) {
			if (p >= end) { ok = false; break; }
			mi->ctg[i].name = p;
			p += mi->ctg[i].name.size() + 1;
		}
		const size_t n_bucket = (size_t)1 << (mi->opt.kmer * 4 - mi->opt.mod_bit);
		mi->ki.resize(n_bucket);
		ok = ok && take(mi->ki.data(), n_bucket * 8);
		const size_t kb_at = at;
		ok = ok && take(nullptr, (size_t)mi->n_kb * 4);
		if (ok && off != mi->l_seq) ok = false;
		if (ok) {
			if (mapped) mi->seq.view((uint8_t*)base + seq_at, seq_bytes);
			else { mi->seq.resize(seq_bytes); memcpy(mi->seq.data(), (const char*)base + seq_at, seq_bytes); }
			// (the occurrence lists follow the names, whose length is arbitrary: the view may be misaligned for uint32_t -- 6 GB at
			// 3 Gbp are not copied for that; the library reads them through IdxArray's memcpy accessors and uploads them as bytes)
			if (mapped) mi->kb.view((uint32_t*)((char*)base + kb_at), (size_t)mi->n_kb);
			else { mi->kb.resize((size_t)mi->n_kb); memcpy(mi->kb.data(), (const char*)base + kb_at, (size_t)mi->n_kb * 4); }
		}
	} catch (const std::exception&) { ok = false; }
	if (!ok) { if (mapped) munmap(base, flen); delete owned; set_error(std::string(fn) + " is not a valid .mpi index"); delete mi; return nullptr; }
	delete owned;                                            // (read path: everything has been copied out)
	if (mapped) mi->map_base = base, mi->map_len = flen;
	compute_block_offsets(mi);
	return mi;
}

int mpa_idx_dump(const char *fn, const mpa_idx_t *mi)
{
	FILE *fp = fopen(fn, "wb");
	if (!fp) { set_error(std::string("cannot write ") + fn); return MPA_ERR_IO; }
	int32_t hdr[2] = { (int32_t)mi->ctg.size(), (int32_t)mi->names.size() };
	fwrite(kMagic, 1, 4, fp);
	fwrite(&mi->opt, sizeof(mi->opt), 1, fp);
	fwrite(&mi->n_kb, 8, 1, fp);
	fwrite(hdr, 4, 2, fp);
	fwrite(&mi->l_seq, 8, 1, fp);
	for (const Contig &c : mi->ctg) fwrite(&c.len, 8, 1, fp);
	fwrite(mi->seq.data(), 1, mi->seq.size(), fp);
	fwrite(mi->names.data(), 1, mi->names.size(), fp);
	fwrite(mi->ki.data(), 8, mi->ki.size(), fp);
	fwrite(mi->kb.data(), 4, mi->kb.size(), fp);
	fclose(fp);
	return MPA_OK;
}

// Genome-only index built from nt4 codes in memory (no k-mer table): enough for the DP operator
// (mpa_dp_run) and used by the operator-level parity tests.  Packing follows ntseq.c:56-70.
static mpa_idx_t *mpa_idx_from_nt4_impl(int32_t n_ctg, const char *const *names, const int64_t *lens, const uint8_t *nt4)
{
	mpa_idx_s *mi = new mpa_idx_s();
	mpa_idxopt_init(&mi->opt);
	int64_t off = 0;
	mi->ctg.resize(n_ctg);
	for (int32_t i = 0; i < n_ctg; ++i) {
		mi->ctg[i].off = off, mi->ctg[i].len = lens[i], mi->ctg[i].name = names[i];
		off += lens[i];
		mi->names.insert(mi->names.end(), names[i], names[i] + strlen(names[i]) + 1);
	}
	mi->l_seq = off;
	mi->seq.assign((size_t)((off + 1) >> 1), 0);
	for (int64_t p = 0; p < off; ++p) mi->seq[p >> 1] |= (uint8_t)((nt4[p] > 4 ? 4 : nt4[p]) << ((p & 1) * 4));
	mi->ki.assign((size_t)1 << (mi->opt.kmer * 4 - mi->opt.mod_bit), 0);
	compute_block_offsets(mi);
	return mi;
}

mpa_idx_t *mpa_idx_from_nt4(int32_t n_ctg, const char *const *names, const int64_t *lens, const uint8_t *nt4)
{
	return mpa::guarded<mpa_idx_t *>(nullptr, [&] { return mpa_idx_from_nt4_impl(n_ctg, names, lens, nt4); });
}

// mp_idx_build (index.c:97-136) = mp_ntseq_read (ntseq.c:29-75) + the k-mer table: FASTA (plain or gzip) -> index.
mpa_idx_t *mpa_idx_build(const char *fn, const mpa_idxopt_t *io, int n_threads)
{
	mpa_idx_t *mi = mpa_idx_read_fasta(fn, io);
	if (mi) mpa_idx_build_kmers(mi, n_threads > 1 ? n_threads : 1);
	return mi;
}

mpa_idx_t *mpa_idx_read_fasta(const char *fn, const mpa_idxopt_t *io)
{
	FastaReader rd(fn);
	if (!rd.ok()) { set_error(std::string("cannot open ") + fn); return nullptr; }
	if (set_trans_code((int)io->trans_code) < 0) return nullptr;
	mpa_idx_s *mi = new mpa_idx_s();
	mi->opt = *io;
	const uint8_t *nt4 = tab_nt4();
	std::string name, comment, seq;
	int64_t off = 0;
	while (rd.next(name, comment, seq)) {
		Contig c;
		c.off = off, c.len = (int64_t)seq.size(), c.name = name;
		mi->ctg.push_back(c);
		mi->names.insert(mi->names.end(), name.c_str(), name.c_str() + name.size() + 1);
		mi->seq.resize((size_t)((off + c.len + 1) >> 1), 0);
		for (size_t i = 0; i < seq.size(); ++i, ++off) mi->seq[(size_t)off >> 1] |= (uint8_t)(nt4[(uint8_t)seq[i]] << ((off & 1) * 4));
	}
	mi->l_seq = off;
	compute_block_offsets(mi);
	mi->ki.assign((size_t)1 << (mi->opt.kmer * 4 - mi->opt.mod_bit), 0);
	return mi;
}

// Build the k-mer -> block table of an index whose genome is already loaded.  Same result as
// build_worker + build_bidx (index.c:52-95): every (contig,strand) is sketched in all three frames with
// the index parameters, then a counting sort lays the block ids out bucket by bucket, strands in order.
static int mpa_idx_build_kmers_impl(mpa_idx_t *mi, int n_threads)
{
	const int32_t n_strand = (int32_t)mi->ctg.size() * 2;
	std::vector<std::vector<uint64_t>> sk(n_strand);
	std::atomic<int32_t> next(0);
	auto work = [&]() {
		std::vector<uint8_t> seq;
		for (;;) {
			const int32_t j = next.fetch_add(1);
			if (j >= n_strand) break;
			const int64_t len = mi->ctg[j >> 1].len;
			seq.resize((size_t)len);
			fetch_nt(mi, j, 0, len, seq.data());
			sketch_nt4(seq.data(), len, mi->opt.min_aa_len, mi->opt.kmer, mi->opt.mod_bit, mi->opt.bbit, mi->bo[j], sk[j]);
		}
	};
	std::vector<std::thread> pool;
	for (int t = 1; t < n_threads && t < n_strand; ++t) pool.emplace_back(work);
	work();
	for (auto &th : pool) th.join();
	const size_t n_bucket = (size_t)1 << (mi->opt.kmer * 4 - mi->opt.mod_bit);
	mi->ki.assign(n_bucket, 0);
	for (const auto &v : sk) for (uint64_t x : v) ++mi->ki[x >> 32];
	int64_t acc = 0;
	for (size_t b = 0; b < n_bucket; ++b) { const int64_t c = mi->ki[b]; mi->ki[b] = acc, acc += c; }
	mi->n_kb = acc;
	mi->kb.assign((size_t)acc, 0);
	std::vector<int64_t> fill(mi->ki.data(), mi->ki.data() + mi->ki.size());
	for (const auto &v : sk) for (uint64_t x : v) mi->kb[fill[x >> 32]++] = (uint32_t)x;
	return MPA_OK;
}

int mpa_idx_build_kmers(mpa_idx_t *mi, int n_threads)
{
	return mpa::guarded<int>(MPA_ERR_HIP, [&] { return mpa_idx_build_kmers_impl(mi, n_threads); });
}

void mpa_idx_destroy(mpa_idx_t *mi)
{
	if (!mi) return;
	dev_free_index(mi);
	if (mi->map_base) munmap(mi->map_base, mi->map_len);
	delete mi;
}

int32_t mpa_idx_n_ctg(const mpa_idx_t *mi) { return (int32_t)mi->ctg.size(); }
int64_t mpa_idx_ctg_len(const mpa_idx_t *mi, int32_t cid) { return mi->ctg[cid].len; }
const char *mpa_idx_ctg_name(const mpa_idx_t *mi, int32_t cid) { return mi->ctg[cid].name.c_str(); }
int64_t mpa_idx_genome_len(const mpa_idx_t *mi) { return mi->l_seq; }
const mpa_idxopt_t *mpa_idx_opt(const mpa_idx_t *mi) { return &mi->opt; }
int64_t mpa_idx_get_nt(const mpa_idx_t *mi, int32_t vid, int64_t st, int64_t en, uint8_t *out) { return fetch_nt(mi, vid, st, en, out); }

// mp_set_spsc (index.c:239-248) + mp_ntseq_read_spsc (ntseq.c:234-296).  The reference keeps sorted (position, score) lists
// per strand and expands them per alignment window; here the track is dense, one byte per strand-local position (0xff = no
// score, else (score + 64) << 1 | is_acceptor, the larger byte winning where several lines name a position), so that the
// device can index it directly.
int64_t mpa_idx_set_spsc(mpa_idx_t *mi, const char *fn, mpa_mapopt_t *mo, int keep_io)
{
	if (!fn) return 0;
	if (!keep_io) mo->io += 10, mo->io_end += 10;
	int32_t max_sc = (mo->io + 1) / 2 - 1;
	if (max_sc > mo->io - mo->go) max_sc = mo->io - mo->go;
	if (max_sc > mo->sp_max_bonus) max_sc = mo->sp_max_bonus;
	return mpa::idx_read_spsc(mi, fn, max_sc);
}
} // extern "C"

// mp_ntseq_read_spsc (ntseq.c:234-296): the score file into the dense per-strand track, scores clamped to +-max_sc
int64_t mpa::idx_read_spsc(mpa_idx_s *mi, const char *fn, int32_t max_sc)
{
	if (max_sc > 63) max_sc = 63;
	gzFile fp = gzopen(fn, "rb");
	if (!fp) { set_error(std::string("cannot open ") + fn); return MPA_ERR_IO; }
	std::unordered_map<std::string, int32_t> cid_of;
	for (size_t i = 0; i < mi->ctg.size(); ++i) cid_of.emplace(mi->ctg[i].name, (int32_t)i);
	mi->spsc.assign((size_t)mi->l_seq * 2, 0xff);
	dev_free_index(mi);                                      // a resident copy without the track is stale
	int64_t n_kept = 0;
	std::string line;
	char buf[65536];
	auto take_line = [&]() -> bool {                        // one line of any length, without its terminator
		line.clear();
		for (;;) {
			if (!gzgets(fp, buf, sizeof(buf))) return !line.empty();
			line += buf;
			if (!line.empty() && line.back() == '\n') { line.pop_back(); if (!line.empty() && line.back() == '\r') line.pop_back(); return true; }
		}
	};
	while (take_line()) {
		const char *f[5];
		int n_f = 0;
		f[n_f++] = line.c_str();
		for (size_t k = 0; k < line.size() && n_f < 5; ++k) if (line[k] == '\t') line[k] = 0, f[n_f++] = line.c_str() + k + 1;
		if (n_f < 5) continue;                               // not enough fields
		{	// the fifth field ends at the next tab, if any
			char *e = const_cast<char*>(strchr(f[4], '\t'));
			if (e) *e = 0;
		}
		const int64_t pos_in = atol(f[1]);
		const int strand = f[2][0] == '+' ? 1 : -1;          // (ntseq.c:262: anything but '+' counts as the reverse strand)
		const int type = f[3][0] == 'D' ? 0 : f[3][0] == 'A' ? 1 : -1;
		int32_t score = atoi(f[4]);
		if (score > max_sc) score = max_sc;
		if (score < -max_sc) score = -max_sc;
		auto it = cid_of.find(f[0]);
		if (it == cid_of.end() || type < 0 || pos_in < 0) continue;
		const Contig &c = mi->ctg[it->second];
		const int64_t pos = strand < 0 ? c.len - pos_in : pos_in;
		if (pos > 0 && pos < c.len) {                        // scores at the contig ends are ignored
			uint8_t &slot = mi->spsc[(size_t)(strand > 0 ? 0 : mi->l_seq) + (size_t)(c.off + pos)];
			const uint8_t v = (uint8_t)((score + 64) << 1 | type);
			if (slot == 0xff || slot < v) slot = v;
			++n_kept;
		}
	}
	gzclose(fp);
	return n_kept;
}

extern "C" {

int64_t mpa_idx_get_spsc(const mpa_idx_t *mi, int32_t vid, int64_t st, int64_t en, uint8_t *out)
{
	const int32_t cid = vid >> 1;
	if (mi->spsc.empty() || cid < 0 || cid >= (int32_t)mi->ctg.size()) return -1;
	const Contig &c = mi->ctg[cid];
	if (st < 0 || en < 0 || st >= c.len) return -1;
	if (en > c.len) en = c.len;
	memset(out, 0xff, (size_t)(en - st));
	const uint8_t *t = mi->spsc.data() + (size_t)((vid & 1) ? mi->l_seq : 0) + (size_t)c.off;
	for (int64_t p = st + 1; p < en; ++p) out[p - st] = t[p];       // the window's first position never gets a score (ntseq.c:144-151)
	return en - st;
}

} // extern "C"
