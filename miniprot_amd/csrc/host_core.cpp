// host_core.cpp -- sorting, sketching and chaining on the host.  See host_core.h.
//
// Reference behaviour reproduced here:
//   sort_pairs_by_x     radix_sort_mp128x  ksort.h:109-162 (tie order is part of the behaviour)
//   sketch_protein      mp_sketch_prot     sketch.c:18-38
//   sketch_nt4          mp_sketch_nt4 + mp_sketch_clean_orf  sketch.c:40-100
//   chain_anchors       mp_chain + mp_chain_backtrack + compact_a  chain.c:8-219
#include <algorithm>
#include <cstdlib>
#include "host_core.h"

namespace mpa {

// ------------------------------------------------------------------------------------------------
// sorting
// ------------------------------------------------------------------------------------------------
// Full-key sort of 64-bit words.  Equal keys are indistinguishable, so any correct algorithm gives the
// reference's result; this is an LSD radix sort that skips constant digits.
void sort_u64(uint64_t *beg, uint64_t *end)
{
	const size_t n = (size_t)(end - beg);
	if (n < 2) return;
	if (n <= 256) { std::sort(beg, end); return; }
	uint64_t all_or = 0, all_and = ~0ULL;
	for (size_t i = 0; i < n; ++i) all_or |= beg[i], all_and &= beg[i];
	const uint64_t varying = all_or ^ all_and;
	static thread_local std::vector<uint64_t> tmp;
	if (tmp.size() < n) tmp.resize(n);
	uint64_t *src = beg, *dst = tmp.data();
	for (int shift = 0; shift < 64; shift += 11) {
		if (((varying >> shift) & 0x7ff) == 0) continue;
		size_t cnt[2048];
		memset(cnt, 0, sizeof(cnt));
		for (size_t i = 0; i < n; ++i) ++cnt[(src[i] >> shift) & 0x7ff];
		size_t acc = 0;
		for (int d = 0; d < 2048; ++d) { size_t c = cnt[d]; cnt[d] = acc, acc += c; }
		for (size_t i = 0; i < n; ++i) dst[cnt[(src[i] >> shift) & 0x7ff]++] = src[i];
		std::swap(src, dst);
	}
	if (src != beg) memcpy(beg, src, n * sizeof(uint64_t));
}

// Anchors (block<<32 | query position) gathered seed by seed in ascending query position: a STABLE sort on the block
// bits alone yields the full-key order (map.c:178 sorts the whole 64-bit word).  Two or three counting passes over the
// significant block bits instead of one per 11 bits of the whole word.
void sort_anchors_by_block(std::vector<uint64_t> &a)
{
	const size_t n = a.size();
	if (n < 2) return;
	if (n <= 256) { std::sort(a.begin(), a.end()); return; }
	uint64_t all_or = 0;
	for (size_t i = 0; i < n; ++i) all_or |= a[i] >> 32;
	int bits = 0;
	while (bits < 32 && (all_or >> bits) != 0) ++bits;
	if (bits == 0) return;
	const int n_pass = bits <= 13 ? 1 : bits <= 26 ? 2 : 3, width = (bits + n_pass - 1) / n_pass;
	static thread_local std::vector<uint64_t> tmp;
	static thread_local std::vector<uint32_t> cnt;
	if (tmp.size() < n) tmp.resize(n);
	cnt.resize((size_t)1 << width);
	uint64_t *src = a.data(), *dst = tmp.data();
	const uint32_t mask = (1u << width) - 1;
	for (int pass = 0; pass < n_pass; ++pass) {
		const int shift = 32 + pass * width;
		std::fill(cnt.begin(), cnt.end(), 0u);
		for (size_t i = 0; i < n; ++i) ++cnt[(src[i] >> shift) & mask];
		uint32_t acc = 0;
		for (size_t d = 0; d < cnt.size(); ++d) { const uint32_t c = cnt[d]; cnt[d] = acc, acc += c; }
		for (size_t i = 0; i < n; ++i) dst[cnt[(src[i] >> shift) & mask]++] = src[i];
		std::swap(src, dst);
	}
	if (src != a.data()) memcpy(a.data(), src, n * sizeof(uint64_t));
}

// Sort of (x,y) pairs by x whose treatment of EQUAL x must match the reference's in-place MSD radix sort
// (it is unstable, and chain extraction / region order / extension limits consume the resulting order).
// Runs of <= 64 elements use a stable insertion sort; longer runs are permuted digit by digit with the
// same cycle-leader walk.
static void pairs_insertion(Pair64 *beg, Pair64 *end)
{
	for (Pair64 *i = beg + 1; i < end; ++i) {
		if (i->x >= (i - 1)->x) continue;
		Pair64 v = *i, *j = i;
		for (; j > beg && v.x < (j - 1)->x; --j) *j = *(j - 1);
		*j = v;
	}
}

static void pairs_digit_pass(Pair64 *beg, Pair64 *end, int shift)
{
	Pair64 *cursor[256], *limit[256];
	size_t cnt[256];
	memset(cnt, 0, sizeof(cnt));
	for (Pair64 *p = beg; p != end; ++p) ++cnt[p->x >> shift & 0xff];
	{
		Pair64 *p = beg;
		for (int d = 0; d < 256; ++d) cursor[d] = p, p += cnt[d], limit[d] = p;
	}
	bool single = false;
	for (int d = 0; d < 256; ++d) if (cnt[d] == (size_t)(end - beg)) { single = true; break; }
	for (int k = 0; !single && k < 256;) {
		if (cursor[k] == limit[k]) { ++k; continue; }
		int d = (int)(cursor[k]->x >> shift & 0xff);
		if (d == k) { ++cursor[k]; continue; }
		Pair64 carried = *cursor[k];             // follow the displacement cycle until an element of bucket k turns up
		do {
			Pair64 placed = carried;
			carried = *cursor[d];
			*cursor[d]++ = placed;
			d = (int)(carried.x >> shift & 0xff);
		} while (d != k);
		*cursor[k]++ = carried;
	}
	if (shift == 0) return;
	const int next = shift > 8 ? shift - 8 : 0;
	Pair64 *p = beg;
	for (int d = 0; d < 256; ++d) {
		Pair64 *e = p + cnt[d];
		if (cnt[d] > 64) pairs_digit_pass(p, e, next);
		else if (cnt[d] > 1) pairs_insertion(p, e);
		p = e;
	}
}

void sort_pairs_by_x(Pair64 *beg, Pair64 *end)
{
	if (end - beg <= 64) { pairs_insertion(beg, end); return; }
	// A digit level at which every key has the same byte leaves the array untouched (each element is already in
	// "its" bucket) and just recurses into that single bucket, so the leading constant bytes can be skipped without
	// changing the result -- keys here are chain scores or block ids, i.e. mostly zero in the top bytes.
	uint64_t all_or = 0, all_and = ~0ULL;
	for (const Pair64 *p = beg; p != end; ++p) all_or |= p->x, all_and &= p->x;
	const uint64_t varying = all_or ^ all_and;
	int shift = 56;
	while (shift > 0 && ((varying >> shift) & 0xff) == 0) shift -= 8;
	pairs_digit_pass(beg, end, shift);
}

// ------------------------------------------------------------------------------------------------
// sketching
// ------------------------------------------------------------------------------------------------
uint32_t hash32_mask(uint32_t key, uint32_t mask)   // sketch.c:7-16; a bijection of the masked key space
{
	key = (key + ~(key << 15)) & mask;
	key ^= key >> 10;
	key = (key + (key << 3)) & mask;
	key ^= key >> 6;
	key = (key + ~(key << 11)) & mask;
	key ^= key >> 16;
	return key;
}

// A rolling k-mer over the 4-bit reduced alphabet with modimizer selection (keep iff hash & (2^m - 1) == 0).
struct KmerRoller {
	uint32_t mask_k, mask_mod, word = 0;
	int32_t kmer, mod_bit, run = 0;
	KmerRoller(int32_t k, int32_t m) : mask_k((1U << k * 4) - 1), mask_mod((1U << m) - 1), kmer(k), mod_bit(m) {}
	void reset() { word = 0, run = 0; }
	// feeds one reduced residue; returns true and sets `bucket` when a selected k-mer ends here
	bool feed(uint32_t c, uint32_t *bucket) {
		word = (word << 4 | c) & mask_k;
		if (++run < kmer) return false;
		uint32_t h = hash32_mask(word, mask_k);
		if (h & mask_mod) return false;
		*bucket = h >> mod_bit;
		return true;
	}
};

void sketch_protein(const char *seq, int32_t len, int32_t kmer, int32_t mod_bit, std::vector<uint64_t> &out)
{
	const uint8_t *aa13 = tab_aa13();
	KmerRoller roll(kmer, mod_bit);
	out.clear();
	for (int32_t i = 0; i < len; ++i) {
		uint32_t c = aa13[(uint8_t)seq[i]], bucket;
		if (c >= 14) { roll.reset(); continue; }        // stop codon / unknown residue
		if (roll.feed(c, &bucket)) out.push_back((uint64_t)bucket << 32 | (uint32_t)i);
	}
}

// Walks the three forward reading frames of a nucleotide string (sketch.c:64-118): a frame's open reading frame
// ends at a stop codon or an ambiguous base, and only ORFs of at least min_aa_len codons contribute k-mers.
// sink(bucket, last_base_index) is called for every selected k-mer.
template<class Sink>
static inline void scan_nt4_kmers(const uint8_t *seq, int64_t len, int32_t min_aa_len, int32_t kmer, int32_t mod_bit, Sink &&sink)
{
	const uint8_t *codon_tab = tab_codon(), *c13 = tab_codon13();
	// one open reading frame per codon phase: its number of codons so far and the position of its last base
	int64_t orf_len[3] = { 0, 0, 0 }, orf_end[3] = { -1, -1, -1 };
	auto close = [&](int f) {
		if (orf_len[f] >= min_aa_len) {
			KmerRoller roll(kmer, mod_bit);
			for (int64_t i = orf_end[f] + 1 - orf_len[f] * 3, en = orf_end[f] + 1; i < en; i += 3) {
				uint32_t bucket;
				if (roll.feed(c13[seq[i] << 4 | seq[i + 1] << 2 | seq[i + 2]], &bucket)) sink(bucket, i + 2);
			}
		}
		orf_len[f] = 0, orf_end[f] = -1;
	};
	int32_t valid = 0;
	uint32_t codon = 0;
	for (int64_t i = 0; i < len; ++i) {
		const int f = (int)((i + 1) % 3);              // phase label of the codon that ends at i
		if (seq[i] < 4) {
			codon = (codon << 2 | seq[i]) & 0x3f;
			if (++valid >= 3) {
				if (codon_tab[codon] >= 20) close(f);     // stop codon
				else orf_end[f] = i, ++orf_len[f];
			}
		} else {                                       // ambiguous base ends everything
			close(0), close(1), close(2);
			valid = 0, codon = 0;
		}
	}
	close(0), close(1), close(2);
}

void sketch_nt4(const uint8_t *seq, int64_t len, int32_t min_aa_len, int32_t kmer, int32_t mod_bit, int32_t bbit, int64_t boff, std::vector<uint64_t> &out, bool sorted)
{
	out.clear();
	scan_nt4_kmers(seq, len, min_aa_len, kmer, mod_bit, [&](uint32_t bucket, int64_t last) {
		out.push_back((uint64_t)bucket << 32 | (uint64_t)((last >> bbit) + boff));
	});
	if (!sorted || out.size() <= 1) return;
	sort_u64(out.data(), out.data() + out.size());
	out.erase(std::unique(out.begin(), out.end()), out.end());
}

// mp_refine_reg's seeding (map.c:97-131) without materialising the window's k-mer list: the reference sorts the
// window's and the query's k-mers together and pairs up equal hashes; the same pairs come out of streaming the window
// past a bitmap of the query's hashes, keeping the (rare) hits, and grouping those.  `qk` is the query's k-mer list
// (hash<<32 | position) sorted; `filter` has bit h set for every hash h in it.
void refine_seed_pairs(const uint8_t *nt, int64_t len, int32_t min_aa_len, int32_t kmer, const std::vector<uint64_t> &qk,
                       const uint64_t *filter, int32_t max_ava, std::vector<uint64_t> &hits, std::vector<uint64_t> &a)
{
	hits.clear(), a.clear();
	// One pass over the window, three interleaved reading frames.  `filter` is indexed by the packed k-mer word (the hash
	// is a bijection of it, so it is only evaluated for the rare hits).  A frame's hits are provisional until its open
	// reading frame turns out to be at least min_aa_len codons long.  The loop is unrolled by three so that the state of
	// each frame (k-mer word, ORF length) lives in registers.
	uint8_t tab[64];                                    // codon -> reduced residue, 0xff for a stop
	{
		const uint8_t *codon_tab = tab_codon(), *c13 = tab_codon13();
		for (int c = 0; c < 64; ++c) tab[c] = codon_tab[c] >= 20 ? 0xff : c13[c];
	}
	const uint32_t mask = (1U << kmer * 4) - 1;
	uint32_t word[3] = { 0, 0, 0 }, codon = 0;
	int32_t orf_len[3] = { 0, 0, 0 }, valid = 0;
	static thread_local std::vector<uint64_t> pend[3];
	for (int f = 0; f < 3; ++f) pend[f].clear();
	auto close = [&](int f) {
		if (!pend[f].empty()) {
			if (orf_len[f] >= min_aa_len) hits.insert(hits.end(), pend[f].begin(), pend[f].end());
			pend[f].clear();
		}
		orf_len[f] = 0, word[f] = 0;
	};
	// f = phase label of the codon ending at base i: (i + 1) % 3
	auto step = [&](int64_t i, const int f) __attribute__((always_inline)) {
		const uint32_t b = nt[i];
		if (__builtin_expect(b < 4, 1)) {
			codon = (codon << 2 | b) & 0x3f;
			if (++valid >= 3) {
				const uint32_t t = tab[codon];
				if (t == 0xff) close(f);
				else {
					const uint32_t w = word[f] = (word[f] << 4 | t) & mask;
					if (++orf_len[f] >= kmer && (filter[w >> 6] >> (w & 63) & 1))
						pend[f].push_back((uint64_t)hash32_mask(w, mask) << 32 | (uint64_t)i);
				}
			}
		} else {
			close(0), close(1), close(2);
			valid = 0, codon = 0;
		}
	};
	int64_t i = 0;
	for (; i + 3 <= len; i += 3) step(i, 1), step(i + 1, 2), step(i + 2, 0);
	if (i < len) step(i, 1), ++i;
	if (i < len) step(i, 2), ++i;
	close(0), close(1), close(2);
	refine_pairs_from_hits(hits, qk, max_ava, a);
}

// second half of the refinement seeding: hits = (hash << 32 | window position) of the window k-mers whose hash the query
// has, in any order (sorted here); all pairs per hash unless there are too many (map.c:112-125)
void refine_pairs_from_hits(std::vector<uint64_t> &hits, const std::vector<uint64_t> &qk, int32_t max_ava, std::vector<uint64_t> &a)
{
	a.clear();
	if (hits.empty()) return;
	sort_u64(hits.data(), hits.data() + hits.size());
	size_t q0 = 0;
	for (size_t k = 0, i = 1; i <= hits.size(); ++i) {
		if (i < hits.size() && hits[k] >> 32 == hits[i] >> 32) continue;
		const uint64_t h = hits[k] >> 32;
		while (q0 < qk.size() && qk[q0] >> 32 < h) ++q0;
		size_t q1 = q0;
		while (q1 < qk.size() && qk[q1] >> 32 == h) ++q1;
		const int64_t n1 = (int64_t)(i - k), n2 = (int64_t)(q1 - q0);
		if (n2 > 0 && (int32_t)n1 * (int32_t)n2 <= max_ava)
			for (size_t i1 = k; i1 < i; ++i1)
				for (size_t i2 = q0; i2 < q1; ++i2)
					a.push_back((uint64_t)(uint32_t)hits[i1] << 32 | (uint32_t)qk[i2]);
		k = i;
	}
	sort_u64(a.data(), a.data() + a.size());
}

// ------------------------------------------------------------------------------------------------
// chaining
// ------------------------------------------------------------------------------------------------
static inline float log2_poly(float x)                 // mp_log2, mppriv.h:91-99 (valid for x >= 2)
{
	union { float f; uint32_t i; } z = { x };
	float r = (float)((int32_t)((z.i >> 23) & 255) - 128);
	z.i &= ~(255u << 23);
	z.i += 127u << 23;
	r += (-0.34484843f * z.f + 2.02466578f) * z.f - 0.67487759f;
	return r;
}

static const int32_t kIncompatible = INT32_MIN;

// score of appending anchor `cur` after anchor `prev` (chain.c:112-151)
static inline int32_t link_score(uint64_t cur, uint64_t prev, const ChainParams &p, int32_t max_dist_x, int32_t max_dist_y)
{
	const int32_t dq = (int32_t)cur - (int32_t)prev, dq3 = dq * 3;
	if (dq <= 0 || dq3 > max_dist_x || dq > max_dist_y) return kIncompatible;
	int32_t dr3, gap, gap_signed, sc;
	if (p.bbit > 0) {
		const int32_t bs = 1 << p.bbit;
		dr3 = (int32_t)(((cur >> 32) - (prev >> 32)) << p.bbit);
		if (dq3 < dr3 - bs) gap = dr3 - bs - dq3, gap_signed = -gap;
		else if (dq3 > dr3 + bs) gap = dq3 - dr3 - bs, gap_signed = gap;
		else gap = gap_signed = 0;
	} else {
		dr3 = (int32_t)((cur >> 32) - (prev >> 32));
		if (dr3 == 0) return kIncompatible;
		gap_signed = dq3 - dr3;
		gap = gap_signed < 0 ? -gap_signed : gap_signed;
	}
	if (gap > p.bw) return kIncompatible;
	if (p.bbit > 0) sc = std::min(p.kmer, dq);
	else if (p.kmer <= dq && p.kmer * 3 <= dr3) sc = p.kmer;
	else {
		const int32_t dr = dr3 / 3;
		sc = std::min(std::min(dr, dq), p.kmer);
		if (dr3 != dr * 3) --sc;                        // the two anchors are in different frames
	}
	if (gap > 0) {
		const float lin = (float)gap * .33334f;
		const float lg = gap >= 2 ? p.coef_log * (log2_poly((float)(gap + 1)) - 1.0f) + 1.0f : (float)gap;
		if (p.is_spliced && gap_signed < 0) sc -= (int)(lin < lg ? lin : lg);   // genomic side longer: possibly an intron
		else sc -= (int)(lin + lg);
	}
	if (p.bbit > 0 && cur >> 32 == prev >> 32 && gap == 0) sc += 2;             // MP_BLOCK_BONUS
	return sc;
}

// The chain ends in the order the reference's sort leaves them -- for the common case that almost every anchor is a
// chain of one.  With min_cnt > 1 such "root" anchors (no predecessor, score == kmer, the smallest key) never yield a
// chain, so only the order of the OTHER ends matters; but that order is a product of the in-place MSD radix sort
// (pairs_digit_pass) run over the whole array, roots included.  With all scores below 256 that sort is a single digit
// pass, and its effect on the non-roots can be replayed without the roots: bucket `kmer` is the first region
// [0, n_root); walking it only ever moves the non-roots found there (in ascending position) into the heads of their
// buckets, displacing what sits there; everything after that happens inside the tail [n_root, n), which is small.
// The view may therefore leave out anchors that are alone in their neighbourhood (v.pos != nullptr).
// Returns false if the preconditions do not hold (the caller then builds and sorts the full list).
struct SparseItem { int64_t pos; Pair64 e; };        // an element that is not a root, and the slot it occupies

// The last digit pass (shift 0) of the reference's sort over a region of n slots that holds roots (key k0, the smallest
// key present) everywhere except at `items` (ascending pos).  Appends the non-roots in their final order.
static void replay_low_digit(int64_t n, int k0, const std::vector<SparseItem> &items, std::vector<Pair64> &out)
{
	const size_t m = items.size(), base = out.size();
	if (m == 0) return;
	size_t cnt[256];
	memset(cnt, 0, sizeof(cnt));
	for (const SparseItem &it : items) ++cnt[it.e.x & 0xff];
	const int64_t n_root = n - (int64_t)m;
	out.resize(base + m, Pair64{ (uint64_t)k0, ~0ULL });
	Pair64 *tail = out.data() + base;
	// the slots [n_root, n) as they are before the pass
	size_t first_tail = 0;
	while (first_tail < m && items[first_tail].pos < n_root) ++first_tail;
	for (size_t i = first_tail; i < m; ++i) tail[items[i].pos - n_root] = items[i].e;
	size_t cursor[256], limit[256];
	{
		size_t at = 0;
		for (int d = 0; d < 256; ++d) cursor[d] = at, at += cnt[d], limit[d] = at;
	}
	// walk of bucket k0 over [0, n_root): roots stay, every other element starts a displacement cycle
	for (size_t i = 0; i < first_tail; ++i) {
		Pair64 carried = items[i].e;
		int d = (int)(carried.x & 0xff);
		do {
			const Pair64 placed = carried;
			carried = tail[cursor[d]];
			tail[cursor[d]++] = placed;
			d = (int)(carried.x & 0xff);
		} while (d != k0);
	}
	// the remaining buckets, entirely inside the tail (same walk as pairs_digit_pass)
	for (int k = k0 + 1; k < 256;) {
		if (cursor[k] == limit[k]) { ++k; continue; }
		int d = (int)(tail[cursor[k]].x & 0xff);
		if (d == k) { ++cursor[k]; continue; }
		Pair64 carried = tail[cursor[k]];
		do {
			const Pair64 placed = carried;
			carried = tail[cursor[d]];
			tail[cursor[d]++] = placed;
			d = (int)(carried.x & 0xff);
		} while (d != k);
		tail[cursor[k]++] = carried;
	}
}

static bool sorted_chain_ends_sparse(const ChainParams &p, const ChainView &v, std::vector<Pair64> &out)
{
	const int64_t n = v.n_total;
	if (p.min_cnt <= 1 || p.min_sc > p.kmer || n <= 64 || p.kmer < 0 || p.kmer > 255) return false;
	static thread_local std::vector<SparseItem> items, moved, merged;
	items.clear();
	int32_t max_f = 0;
	for (int64_t i = 0; i < v.m; ++i)
		if (v.pred[i] >= 0) {
			if (v.f[i] <= p.kmer) return false;           // (f > kmer always holds for a chained anchor; belt and braces)
			max_f = std::max(max_f, v.f[i]);
			items.push_back(SparseItem{ v.pos ? (int64_t)v.pos[i] : i, Pair64{ (uint64_t)(int64_t)v.f[i], (uint64_t)i } });
		}
	out.clear();
	if (max_f < 256) { replay_low_digit(n, p.kmer, items, out); return true; }
	if (max_f >= 65536) return false;
	// ---- two digit levels.  Level 1 (shift 8): bucket 0 = all scores below 256 (the roots among them) is the region
	// [0, c0); walking it moves the high scores found there into the heads of their buckets behind c0, and what they
	// displace from there -- low scores -- into the holes they leave.
	size_t cnt8[256];
	memset(cnt8, 0, sizeof(cnt8));
	int64_t m_hi = 0;
	for (const SparseItem &it : items) if (it.e.x >= 256) ++cnt8[it.e.x >> 8 & 0xff], ++m_hi;
	const int64_t c0 = n - m_hi;
	static thread_local std::vector<Pair64> tail8;
	const uint64_t kAbsent = ~0ULL;
	tail8.assign((size_t)m_hi, Pair64{ (uint64_t)p.kmer, kAbsent });
	{	// slots [c0, n) as they are: every view element there, roots included (their identity never matters)
		int64_t i = v.m;
		if (!v.pos) i = std::min<int64_t>(c0, v.m);
		else i = std::lower_bound(v.pos, v.pos + v.m, (int32_t)std::min<int64_t>(c0, INT32_MAX)) - v.pos;
		for (; i < v.m; ++i) tail8[(size_t)((v.pos ? (int64_t)v.pos[i] : i) - c0)] = Pair64{ (uint64_t)(int64_t)v.f[i], v.pred[i] >= 0 ? (uint64_t)i : kAbsent };
	}
	size_t cursor[256], limit[256], start[256];
	{
		size_t at = 0;
		for (int d = 0; d < 256; ++d) start[d] = cursor[d] = at, at += d ? cnt8[d] : 0, limit[d] = at;
	}
	moved.clear(), merged.clear();
	for (const SparseItem &it : items) {
		if (it.pos >= c0) break;
		if (it.e.x < 256) continue;                        // a low score inside region 0 stays where it is
		Pair64 carried = it.e;
		int d = (int)(carried.x >> 8 & 0xff);
		do {
			const Pair64 placed = carried;
			carried = tail8[cursor[d]];
			tail8[cursor[d]++] = placed;
			d = (int)(carried.x >> 8 & 0xff);
		} while (d != 0);
		if (carried.y != kAbsent) moved.push_back(SparseItem{ it.pos, carried });   // a chained anchor with a low score: it now sits in the hole
	}
	for (int k = 1; k < 256;) {                             // buckets 1..255 (same walk as pairs_digit_pass)
		if (cursor[k] == limit[k]) { ++k; continue; }
		int d = (int)(tail8[cursor[k]].x >> 8 & 0xff);
		if (d == k) { ++cursor[k]; continue; }
		Pair64 carried = tail8[cursor[k]];
		do {
			const Pair64 placed = carried;
			carried = tail8[cursor[d]];
			tail8[cursor[d]++] = placed;
			d = (int)(carried.x >> 8 & 0xff);
		} while (d != k);
		tail8[cursor[k]++] = carried;
	}
	// ---- level 2 inside bucket 0: the low scores at their (possibly new) positions, ascending
	{
		size_t a = 0, b = 0;
		while (a < items.size() && items[a].pos < c0) {
			if (items[a].e.x >= 256) { ++a; continue; }
			while (b < moved.size() && moved[b].pos < items[a].pos) merged.push_back(moved[b++]);
			merged.push_back(items[a++]);
		}
		while (b < moved.size()) merged.push_back(moved[b++]);
	}
	if (c0 > 64) replay_low_digit(c0, p.kmer, merged, out);
	else {                                                  // a short bucket is insertion-sorted: stable
		std::stable_sort(merged.begin(), merged.end(), [](const SparseItem &x, const SparseItem &y) { return x.e.x < y.e.x; });
		for (const SparseItem &it : merged) out.push_back(it.e);
	}
	// ---- level 2 inside the other buckets (all of it is present): digit pass or insertion sort, as the reference recurses
	for (int d = 1; d < 256; ++d) {
		Pair64 *b = tail8.data() + start[d], *e = b + cnt8[d];
		if (cnt8[d] > 64) pairs_digit_pass(b, e, 0);
		else if (cnt8[d] > 1) pairs_insertion(b, e);
		out.insert(out.end(), b, e);
	}
	return true;
}

// chain.c:8-110 on the result of the forward pass: extraction (best chain end first), then the kept chains laid out in
// ascending anchor order and sorted by their first target position.
// set_only: the caller wants the anchors of the kept chains as a SET, ascending (the pre-chain, map.c:186-192, whose chains
// are dissolved again by the sort that follows it): no chain list, no layout, and no sort either -- the view is in ascending
// anchor order, so marking the kept anchors and reading the view through gives them sorted.
static void chain_extract_impl(const ChainParams &p, const ChainView &v, std::vector<uint64_t> &a_out, std::vector<uint64_t> &u, bool set_only)
{
	u.clear();
	const int32_t max_drop = p.is_spliced ? INT32_MAX : p.bw;
	const int32_t *f = v.f, *pred = v.pred;
	static thread_local std::vector<int32_t> mark, order;
	static thread_local std::vector<Pair64> ends, first;
	static thread_local std::vector<uint64_t> packed, u_sorted;
	const uint64_t kAbsent = ~0ULL;                     // an anchor the view leaves out (alone in its neighbourhood)
	ends.clear();
	if (!sorted_chain_ends_sparse(p, v, ends)) {
		// the full list: absent anchors are roots (score kmer); they can only be left out when min_cnt rejects them anyway
		int64_t vi = 0;
		for (int64_t q = 0; q < v.n_total; ++q) {
			if (vi < v.m && (v.pos ? v.pos[vi] : vi) == q) {
				if (f[vi] >= p.min_sc) ends.push_back(Pair64{ (uint64_t)(int64_t)f[vi], (uint64_t)vi });
				++vi;
			} else if (p.kmer >= p.min_sc) ends.push_back(Pair64{ (uint64_t)(int64_t)p.kmer, kAbsent });
		}
		sort_pairs_by_x(ends.data(), ends.data() + ends.size());
	}
	if (ends.empty()) { a_out.clear(); return; }
	mark.assign((size_t)v.m, 0);
	order.clear();                                     // anchor indices of the kept chains, each chain from its END backwards
	for (int64_t k = (int64_t)ends.size() - 1; k >= 0; --k) {
		if (ends[k].y == kAbsent) continue;
		const int64_t tail = (int64_t)ends[k].y;
		const int32_t tail_f = (int32_t)ends[k].x;
		if (mark[tail] != 0) continue;
		// An unclaimed anchor without predecessor is a one-anchor chain.  When min_cnt rejects those, the only trace it
		// would leave is its own claim mark, which nothing reads any more: every anchor chained onto it scores above
		// kmer == f[tail] and has therefore been visited already.
		if (pred[tail] < 0 && p.min_cnt > 1) continue;
		// where does the best-scoring suffix of this chain start?  walk back until a claimed anchor or the chain's root
		int64_t cut = tail, walk = tail, stop = -1;
		int32_t best_suffix = 0;
		do {
			mark[walk] = 2;
			stop = walk = pred[walk];
			const int32_t s = walk < 0 ? tail_f : tail_f - f[walk];
			if (s > best_suffix) best_suffix = s, cut = walk;
			else if (best_suffix - s > max_drop) break;
		} while (walk >= 0 && mark[walk] == 0);
		for (int64_t x = tail; x >= 0 && x != stop; x = pred[x]) mark[x] = 0;
		// claim the anchors of the suffix
		const size_t before = order.size();
		int64_t x = tail;
		for (; x != cut; x = pred[x]) order.push_back((int32_t)x), mark[x] = 1;
		const int32_t sc = x < 0 ? tail_f : tail_f - f[x];
		const int64_t cnt = (int64_t)(order.size() - before);
		if (sc >= p.min_sc && cnt > 0 && cnt >= p.min_cnt) u.push_back((uint64_t)sc << 32 | (uint64_t)cnt);
		else order.resize(before);                       // rejected chains keep their anchors claimed
	}
	if (u.empty()) { a_out.clear(); return; }
	if (set_only) {
		// (mark: 1 = claimed; the anchors of the kept chains are exactly those in `order`)
		static thread_local std::vector<uint8_t> kept;
		kept.assign((size_t)v.m, 0);
		for (int32_t x : order) kept[(size_t)x] = 1;
		a_out.resize(order.size());
		size_t k = 0;
		for (int64_t i = 0; i < v.m; ++i) if (kept[(size_t)i]) a_out[k++] = v.a[i];
		u.clear();
		return;
	}

	// ---- lay the chains out in ascending anchor order, chains sorted by their first target position (chain.c:77-110)
	packed.resize(order.size());
	first.resize(u.size());
	{
		size_t k = 0;
		for (size_t c = 0; c < u.size(); ++c) {
			const size_t cnt = (uint32_t)u[c];
			for (size_t j = 0; j < cnt; ++j) packed[k + j] = v.a[order[k + cnt - 1 - j]];
			first[c] = Pair64{ packed[k] >> 32, (uint64_t)k << 32 | (uint64_t)c };
			k += cnt;
		}
	}
	sort_pairs_by_x(first.data(), first.data() + first.size());
	u_sorted.resize(u.size());
	a_out.resize(packed.size());
	size_t k = 0;
	for (size_t c = 0; c < u.size(); ++c) {
		const uint32_t which = (uint32_t)first[c].y;
		const size_t cnt = (uint32_t)u[which], from = (size_t)(first[c].y >> 32);
		u_sorted[c] = u[which];
		memcpy(&a_out[k], &packed[from], cnt * sizeof(uint64_t));
		k += cnt;
	}
	u.assign(u_sorted.begin(), u_sorted.end());
}

void chain_extract(const ChainParams &p, const ChainView &v, std::vector<uint64_t> &a_out, std::vector<uint64_t> &u)
{
	chain_extract_impl(p, v, a_out, u, false);
}

void chain_extract_set(const ChainParams &p, const ChainView &v, std::vector<uint64_t> &a_out)
{
	static thread_local std::vector<uint64_t> u;
	chain_extract_impl(p, v, a_out, u, true);
}

void chain_anchors_after_forward(const ChainParams &p, const int32_t *f, const int32_t *pred, std::vector<uint64_t> &a, std::vector<uint64_t> &u)
{
	u.clear();
	const int64_t n = (int64_t)a.size();
	if (n == 0) return;
	static thread_local std::vector<uint64_t> a_in;
	a_in.assign(a.begin(), a.end());
	const ChainView v{ n, n, nullptr, f, pred, a_in.data() };
	chain_extract(p, v, a, u);
}

void chain_forward(const ChainParams &p, const uint64_t *a_ptr, int64_t n, int32_t *f_out, int32_t *pred_out)
{
	std::vector<uint64_t> a(a_ptr, a_ptr + n), u;
	chain_anchors(p, a, u, f_out, pred_out);
}

void chain_anchors_set(const ChainParams &p, std::vector<uint64_t> &a)
{
	static thread_local std::vector<uint64_t> u;
	chain_anchors(p, a, u, nullptr, nullptr, true);
}

void chain_anchors(const ChainParams &p, std::vector<uint64_t> &a, std::vector<uint64_t> &u, int32_t *f_copy, int32_t *pred_copy, bool set_only)
{
	u.clear();
	const int64_t n = (int64_t)a.size();
	if (n == 0) return;
	int32_t max_dist_x = std::max(p.max_dist_x, p.bw), max_dist_y = p.max_dist_y;
	if (max_dist_y < p.bw && !p.is_spliced) max_dist_y = p.bw;

	// per-thread scratch, reused across calls: with hundreds of workers, fresh multi-hundred-KB vectors per call turn
	// into mmap/munmap traffic on the shared address space
	static thread_local std::vector<int32_t> f, mark, pred;
	static thread_local std::vector<uint64_t> a_in;
	f.resize(n), pred.resize(n), mark.assign(n, 0);

	// ---- forward pass: best predecessor of every anchor (chain.c:181-209)
	int64_t win = 0, hot = -1;         // start of the target-distance window; anchor with the best score so far
	int32_t hot_f = 0;
	for (int64_t i = 0; i < n; ++i) {
		int64_t best_j = -1;
		int32_t best = p.kmer, skipped = 0;
		while (win < i && (int64_t)(((a[i] >> 32) - (a[win] >> 32)) << p.bbit) > max_dist_x) ++win;
		if (hot >= 0 && hot >= win) {                   // always try the best anchor seen so far, whatever max_skip does below
			int32_t sc = hot_f + link_score(a[i], a[hot], p, max_dist_x, max_dist_y);
			if (sc > best) best = sc, best_j = hot;
		} else hot_f = 0, hot = -1;
		if (i - win > p.max_iter) win = i - p.max_iter;
		for (int64_t j = i - 1; j >= win; --j) {
			int32_t sc = link_score(a[i], a[j], p, max_dist_x, max_dist_y);
			if (sc == kIncompatible) continue;
			sc += f[j];
			if (sc > best) {
				best = sc, best_j = j;
				if (skipped > 0) --skipped;
			} else if (mark[j] == (int32_t)i) {         // j is already a predecessor of a better-placed anchor
				if (++skipped > p.max_skip) break;
			}
			if (pred[j] >= 0) mark[pred[j]] = (int32_t)i;
		}
		f[i] = best, pred[i] = (int32_t)best_j;
		if (hot_f < best) hot_f = best, hot = i;
	}
	if (f_copy) memcpy(f_copy, f.data(), (size_t)n * 4), memcpy(pred_copy, pred.data(), (size_t)n * 4);
	a_in.assign(a.begin(), a.end());
	const ChainView v{ n, n, nullptr, f.data(), pred.data(), a_in.data() };
	chain_extract_impl(p, v, a, u, set_only);
}

} // namespace mpa

// Test hooks (not part of include/mpamd.h): the host stages with reference-defined tie behaviour, callable on raw arrays so
// that tests/ can fuzz them against the oracle.  Same argument order as mp_chain (chain.c:160).
extern "C" int32_t mpa_dbg_chain(int32_t max_dist_x, int32_t max_dist_y, int32_t bw, int32_t max_skip, int32_t max_iter, int32_t min_cnt,
                                 int32_t min_sc, float coef_log, int32_t is_spliced, int32_t kmer, int32_t bbit,
                                 int64_t n, uint64_t *a, uint64_t *u, int64_t *n_out)
{
	mpa::ChainParams p{ max_dist_x, max_dist_y, bw, max_skip, max_iter, min_cnt, min_sc, coef_log, is_spliced, kmer, bbit };
	std::vector<uint64_t> av(a, a + n), uv;
	mpa::chain_anchors(p, av, uv);
	memcpy(a, av.data(), av.size() * 8);
	memcpy(u, uv.data(), uv.size() * 8);
	*n_out = (int64_t)av.size();
	return (int32_t)uv.size();
}

// The pre-chain through a SPARSE view built the way the GPU stage builds it (seed_exec.hip): forward pass, then only the
// anchors of runs of >= 2 anchors whose consecutive blocks differ by at most one.  Lets the CPU tests cover
// chain_extract()'s sparse branch.
extern "C" int32_t mpa_dbg_prechain_sparse(int32_t bw, int32_t max_skip, int32_t max_iter, float coef_log, int32_t is_spliced, int32_t kmer, int32_t bbit,
                                           int64_t n, uint64_t *a, uint64_t *u, int64_t *n_out)
{
	using namespace mpa;
	ChainParams p{ bw, bw, bw, max_skip, max_iter, 2, 0, coef_log, is_spliced, kmer, bbit };
	// dense forward pass (same loop as chain_anchors)
	std::vector<int32_t> f(n), pred(n), mark(n, 0);
	const int32_t max_dist_x = std::max(p.max_dist_x, p.bw);
	int32_t max_dist_y = p.max_dist_y;
	if (max_dist_y < p.bw && !p.is_spliced) max_dist_y = p.bw;
	int64_t win = 0, hot = -1;
	int32_t hot_f = 0;
	for (int64_t i = 0; i < n; ++i) {
		int64_t best_j = -1;
		int32_t best = p.kmer, skipped = 0;
		while (win < i && (int64_t)(((a[i] >> 32) - (a[win] >> 32)) << p.bbit) > max_dist_x) ++win;
		if (hot >= 0 && hot >= win) {
			int32_t sc = hot_f + link_score(a[i], a[hot], p, max_dist_x, max_dist_y);
			if (sc > best) best = sc, best_j = hot;
		} else hot_f = 0, hot = -1;
		if (i - win > p.max_iter) win = i - p.max_iter;
		for (int64_t j = i - 1; j >= win; --j) {
			int32_t sc = link_score(a[i], a[j], p, max_dist_x, max_dist_y);
			if (sc == kIncompatible) continue;
			sc += f[j];
			if (sc > best) { best = sc, best_j = j; if (skipped > 0) --skipped; }
			else if (mark[j] == (int32_t)i) { if (++skipped > p.max_skip) break; }
			if (pred[j] >= 0) mark[pred[j]] = (int32_t)i;
		}
		f[i] = best, pred[i] = (int32_t)best_j;
		if (hot_f < best) hot_f = best, hot = i;
	}
	// sparse view: runs of >= 2
	const uint64_t max_dblock = (uint64_t)(max_dist_x >> p.bbit);
	std::vector<int32_t> vpos, vf, vpred;
	std::vector<uint64_t> va;
	for (int64_t s = 0; s < n;) {
		int64_t e = s + 1;
		while (e < n && (a[e] >> 32) - (a[e - 1] >> 32) <= max_dblock) ++e;
		if (e - s >= 2)
			for (int64_t i = s; i < e; ++i) {
				const int32_t c = (int32_t)vpos.size();
				vpos.push_back((int32_t)i), vf.push_back(f[i]), va.push_back(a[i]);
				vpred.push_back(pred[i] < 0 ? -1 : c - (int32_t)(i - pred[i]));
			}
		s = e;
	}
	const ChainView v{ n, (int64_t)vpos.size(), vpos.data(), vf.data(), vpred.data(), va.data() };
	std::vector<uint64_t> av, uv;
	chain_extract(p, v, av, uv);
	memcpy(a, av.data(), av.size() * 8);
	memcpy(u, uv.data(), uv.size() * 8);
	*n_out = (int64_t)av.size();
	return (int32_t)uv.size();
}

extern "C" void mpa_dbg_sort_pairs(uint64_t *beg, uint64_t *end)
{
	mpa::sort_pairs_by_x((mpa::Pair64*)beg, (mpa::Pair64*)end);
}

extern "C" void mpa_dbg_sort_anchors(uint64_t *a, int64_t n)
{
	std::vector<uint64_t> v(a, a + n);
	mpa::sort_anchors_by_block(v);
	memcpy(a, v.data(), (size_t)n * 8);
}

