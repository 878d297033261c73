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

// Sort of (x,y) pairs by x whose treatment of EQUAL x must match the reference's in-place MSD radix sort (it is unstable,
// and chain extraction / region order / extension limits consume the resulting order): chain_core.h, shared with the device.
void sort_pairs_by_x(Pair64 *beg, Pair64 *end)
{
	static thread_local std::vector<SortRange> stack;
	const size_t need = (size_t)(end - beg) / 64 + 4;
	if (stack.size() < need) stack.resize(need);
	uint32_t hist[768];
	sort_pairs_by_x_core(beg, end, stack.data(), hist);
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

// chain.c:8-110 on the result of the forward pass (chain_core.h: chain_extract_core, the same source the device compiles);
// here only the work space.
static void chain_extract_impl(const ChainParams &p, const ChainView &v, std::vector<uint64_t> &a_out, std::vector<uint64_t> &u, bool set_only)
{
	static thread_local std::vector<int32_t> mark, order;
	static thread_local std::vector<Pair64> ends, tail8, first;
	static thread_local std::vector<SparseItem> items, moved, merged;
	static thread_local std::vector<uint8_t> kept;
	static thread_local std::vector<uint64_t> packed, u_sorted;
	static thread_local std::vector<SortRange> stack;
	const size_t m = (size_t)v.m;
	const size_t ends_cap = chain_ends_sparse_possible(p, v) ? m : (size_t)std::max(v.n_total, v.m);
	auto grow = [](auto &vec, size_t n) { if (vec.size() < n) vec.resize(n + n / 4 + 16); };
	grow(mark, m), grow(order, m), grow(ends, std::max(ends_cap, m)), grow(tail8, m), grow(first, m), grow(items, m), grow(moved, m), grow(merged, m);
	grow(kept, m), grow(packed, m), grow(u_sorted, m), grow(stack, std::max(ends_cap, m) / 64 + 4);
	uint32_t hist[1280];
	const ExtractScratch S{ mark.data(), order.data(), ends.data(), (int64_t)std::max(ends_cap, m), items.data(), moved.data(), merged.data(), tail8.data(), first.data(),
	                        kept.data(), packed.data(), u_sorted.data(), stack.data(), hist };
	static thread_local std::vector<uint64_t> a_buf, u_buf;  // (grow-only: a resize of the outputs would zero m words per call)
	grow(a_buf, m), grow(u_buf, m);
	int64_t n_a = 0, n_u = 0;
	int rc = chain_extract_core<CoopSerial>(p, v, S, a_buf.data(), &n_a, u_buf.data(), &n_u, set_only);
	if (rc != 0) {                                          // (a sparse view whose preconditions failed after all: the full list)
		grow(ends, (size_t)v.n_total), grow(stack, (size_t)v.n_total / 64 + 4);
		ExtractScratch S2 = S;
		S2.ends = ends.data(), S2.ends_cap = v.n_total, S2.stack = stack.data();
		rc = chain_extract_core<CoopSerial>(p, v, S2, a_buf.data(), &n_a, u_buf.data(), &n_u, set_only);
	}
	a_out.assign(a_buf.begin(), a_buf.begin() + n_a), u.assign(u_buf.begin(), u_buf.begin() + n_u);
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

