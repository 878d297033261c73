// seed_exec.hip -- GPU side of the seeding stage, included by dp_exec.hip.
//
// For a whole mini-batch at once:
//   k_seed_gather   anchors of every kept seed: (query, block) key + query position      map.c:163-177
//   rocPRIM         one stable LSD radix sort over (query, block)                          map.c:178 (radix_sort_mp64)
//   k_prechain_fwd  forward pass of the block-resolution pre-chain                         chain.c:112-151,181-209
//   scan + k_seed_compact   only anchors that are chained -- have a predecessor or are one -- travel back to the host
// The pre-chain only links anchors at most one 256-bp block apart, so the sorted anchors of a query fall into
// independent "runs" (maximal stretches in which consecutive blocks differ by at most max_dblock); ~97 % of the anchors
// at genome scale are runs of one, and of the rest about half link to nothing (wrong order or distance on the query): the
// extraction (host, chain_extract on a sparse ChainView) never needs to see an anchor that is neither chained nor chained onto.
// One thread walks one run with exactly the reference's loop; the arithmetic of comput_sc() includes two float
// operations, compiled without FMA contraction like the host code.
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

namespace mpa {

struct U32ToU64 { __host__ __device__ uint64_t operator()(uint32_t x) const { return (uint64_t)x; } };   // scan inputs of 32-bit counts as 64-bit sums

struct SeedJobDev { int64_t kb_off, dst; int32_t cnt, qpos, qid, pad; };

__global__ __launch_bounds__(256) void k_seed_gather(const SeedJobDev *jobs, int32_t n_jobs, const uint32_t *kb, int nb, uint64_t *key, uint32_t *val)
{
	MPA_SHORT_KERNEL();
	const int32_t j = blockIdx.x;
	if (j >= n_jobs) return;
	const SeedJobDev s = jobs[j];
	const uint64_t hi = (uint64_t)s.qid << nb;
	for (int32_t k = threadIdx.x; k < s.cnt; k += 256) {
		key[s.dst + k] = hi | kb[s.kb_off + k];
		val[s.dst + k] = (uint32_t)s.qpos;
	}
}

struct PreParams { int32_t max_dist_x, max_dist_y, bw, max_skip, max_iter, kmer, bbit, is_spliced, max_dblock; float coef_log; };

__device__ __forceinline__ float d_log2_poly(float x)          // mp_log2 (mppriv.h:91-99)
{
	union { float f; uint32_t i; } z = { x };
	float r = (float)((int32_t)((z.i >> 23) & 255) - 128);
	z.i &= ~(255u << 23);
	z.i += 127u << 23;
	r += (-0.34484843f * z.f + 2.02466578f) * z.f - 0.67487759f;
	return r;
}

#define D_INCOMPAT INT32_MIN
// comput_sc (chain.c:112-151) for block anchors (bbit > 0); blocks are query-local differences, so the query bits cancel
__device__ __forceinline__ int32_t d_link_score(uint64_t kcur, uint32_t qcur, uint64_t kprev, uint32_t qprev, const PreParams &p)
{
	const int32_t dq = (int32_t)qcur - (int32_t)qprev, dq3 = dq * 3;
	if (dq <= 0 || dq3 > p.max_dist_x || dq > p.max_dist_y) return D_INCOMPAT;
	const int32_t bs = 1 << p.bbit;
	const int32_t dr3 = (int32_t)((kcur - kprev) << p.bbit);
	int32_t gap, gap_signed;
	if (dq3 < dr3 - bs) gap = dr3 - bs - dq3, gap_signed = -gap;
	else if (dq3 > dr3 + bs) gap = dq3 - dr3 - bs, gap_signed = gap;
	else gap = gap_signed = 0;
	if (gap > p.bw) return D_INCOMPAT;
	int32_t sc = p.kmer < dq ? p.kmer : dq;
	if (gap > 0) {
		const float lin = (float)gap * .33334f;
		const float lg = gap >= 2 ? p.coef_log * (d_log2_poly((float)(gap + 1)) - 1.0f) + 1.0f : (float)gap;
		if (p.is_spliced && gap_signed < 0) sc -= (int)(lin < lg ? lin : lg);
		else sc -= (int)(lin + lg);
	}
	if (kcur == kprev && gap == 0) sc += 2;                    // MP_BLOCK_BONUS
	return sc;
}

// one thread per anchor; the thread of a run's first anchor walks the run (mp_chain's forward loop, chain.c:181-209)
__global__ __launch_bounds__(256) void k_prechain_fwd(const uint64_t *key, const uint32_t *val, int64_t n, int nb, const int64_t *qfirst, PreParams p,
                                                      int32_t *f, int32_t *pred, int32_t *mark, uint32_t *flag)
{
	MPA_SHORT_KERNEL();
	const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
	if (s >= n) return;
	const uint64_t ks = key[s];
	if (s > 0 && ks - key[s - 1] <= (uint64_t)p.max_dblock) return;      // not the head of its run
	int64_t e = s + 1;
	{
		uint64_t prev = ks;
		while (e < n) { const uint64_t k = key[e]; if (k - prev > (uint64_t)p.max_dblock) break; prev = k, ++e; }
	}
	if (e - s < 2) return;                                               // alone: f = kmer, no predecessor (the defaults)
	const int64_t q0 = qfirst[ks >> nb];                                 // positions are stored relative to the query's first anchor
	int64_t win = s, hot = -1;
	int32_t hot_f = 0;
	for (int64_t i = s; i < e; ++i) {
		const uint64_t ki = key[i];
		const uint32_t qi = val[i];
		const int32_t me = (int32_t)(i - q0);
		int64_t best_j = -1;
		int32_t best = p.kmer, skipped = 0;
		while (win < i && (int64_t)((ki - key[win]) << p.bbit) > p.max_dist_x) ++win;
		if (hot >= 0 && hot >= win) {
			const int32_t l = d_link_score(ki, qi, key[hot], val[hot], p);
			const int32_t sc = hot_f + l;                                  // (INT32_MIN + hot_f stays far below kmer, as on the host)
			if (sc > best) best = sc, best_j = hot;
		} else hot_f = 0, hot = -1;
		if (i - win > p.max_iter) win = i - p.max_iter;
		for (int64_t j = i - 1; j >= win; --j) {
			int32_t sc = d_link_score(ki, qi, key[j], val[j], p);
			if (sc == D_INCOMPAT) continue;
			sc += f[j];
			if (sc > best) {
				best = sc, best_j = j;
				if (skipped > 0) --skipped;
			} else if (mark[j] == me) {
				if (++skipped > p.max_skip) break;
			}
			const int32_t pj = pred[j];
			if (pj >= 0) mark[q0 + pj] = me;
		}
		f[i] = best, pred[i] = best_j < 0 ? -1 : (int32_t)(best_j - q0);
		if (best_j >= 0) flag[i] = 1, flag[best_j] = 1;                 // (this thread owns the whole run: no race)
		if (hot_f < best) hot_f = best, hot = i;
	}
}

// ------------------------------------------------------------------------------------------------
// k_chain_fwd: the forward pass of mp_chain (chain.c:181-209) for ANY chaining problem of the path -- the main chain over the
// pre-chain's survivors (block anchors, bbit > 0) and the refinement chain of a region (base resolution, bbit = 0, the exact-gap
// branch of comput_sc, chain.c:123-139).  Input: the sorted anchors x << 32 | y of many problems back to back (first[] = problem
// boundaries).  Two anchors further apart than max_dist_x on the target can never be linked, so a problem falls into
// independent runs; one thread per anchor, the thread of a run's first anchor walks the run with exactly the reference's loop
// (distance window, best-so-far rescue, max_iter, the max_skip counter over the marks).  Output: f[] and pred[] (index
// inside the problem, -1 for none) of every anchor; the extraction stays on the host (chain_extract), because it consumes
// the tie order of the reference's in-place radix sort.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int32_t d_link_score_a(uint64_t cur, uint64_t prev, const PreParams &p)
{
	const int32_t dq = (int32_t)(uint32_t)cur - (int32_t)(uint32_t)prev, dq3 = dq * 3;
	if (dq <= 0 || dq3 > p.max_dist_x || dq > p.max_dist_y) return D_INCOMPAT;
	int32_t dr3, gap, gap_signed, sc;
	if (p.bbit > 0) {
		const int32_t bs = 1 << p.bbit;
		dr3 = (int32_t)(((cur >> 32) - (prev >> 32)) << p.bbit);
		if (dq3 < dr3 - bs) gap = dr3 - bs - dq3, gap_signed = -gap;
		else if (dq3 > dr3 + bs) gap = dq3 - dr3 - bs, gap_signed = gap;
		else gap = gap_signed = 0;
	} else {
		dr3 = (int32_t)((cur >> 32) - (prev >> 32));
		if (dr3 == 0) return D_INCOMPAT;
		gap_signed = dq3 - dr3;
		gap = gap_signed < 0 ? -gap_signed : gap_signed;
	}
	if (gap > p.bw) return D_INCOMPAT;
	if (p.bbit > 0) sc = p.kmer < dq ? p.kmer : dq;
	else if (p.kmer <= dq && p.kmer * 3 <= dr3) sc = p.kmer;
	else {
		const int32_t dr = dr3 / 3;
		sc = dr < dq ? dr : dq;
		if (p.kmer < sc) sc = p.kmer;
		if (dr3 != dr * 3) --sc;                                    // the two anchors are in different frames
	}
	if (gap > 0) {
		const float lin = (float)gap * .33334f;
		const float lg = gap >= 2 ? p.coef_log * (d_log2_poly((float)(gap + 1)) - 1.0f) + 1.0f : (float)gap;
		if (p.is_spliced && gap_signed < 0) sc -= (int)(lin < lg ? lin : lg);
		else sc -= (int)(lin + lg);
	}
	if (p.bbit > 0 && cur >> 32 == prev >> 32 && gap == 0) sc += 2;   // MP_BLOCK_BONUS
	return sc;
}

__global__ __launch_bounds__(256) void k_chain_fwd(const uint64_t *a, int64_t n, const int64_t *first, int32_t n_prob, PreParams p,
                                                   int32_t *f, int32_t *pred, int32_t *mark)
{
	MPA_SHORT_KERNEL();
	const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
	if (s >= n) return;
	// the problem this anchor belongs to: largest q with first[q] <= s
	int32_t lo = 0, hi = n_prob - 1;
	while (lo < hi) { const int32_t mid = (lo + hi + 1) >> 1; if (first[mid] <= s) lo = mid; else hi = mid - 1; }
	const int64_t q0 = first[lo], q1 = first[lo + 1];
	const uint64_t xs = a[s] >> 32;
	const uint64_t max_dx = (uint64_t)(p.max_dist_x >> p.bbit);          // (x_i - x_j) << bbit > max_dist_x  <=>  x_i - x_j > max_dist_x >> bbit
	if (s > q0 && xs - (a[s - 1] >> 32) <= max_dx) return;               // not the head of its run
	int64_t e = s + 1;
	{
		uint64_t prev = xs;
		while (e < q1) { const uint64_t x = a[e] >> 32; if (x - prev > max_dx) break; prev = x, ++e; }
	}
	int64_t win = s, hot = -1;
	int32_t hot_f = 0;
	for (int64_t i = s; i < e; ++i) {
		const uint64_t ai = a[i];
		const int32_t me = (int32_t)(i - q0);
		int64_t best_j = -1;
		int32_t best = p.kmer, skipped = 0;
		while (win < i && (ai >> 32) - (a[win] >> 32) > max_dx) ++win;
		if (hot >= 0 && hot >= win) {
			const int32_t l = d_link_score_a(ai, a[hot], p);
			const int32_t sc = l == D_INCOMPAT ? D_INCOMPAT : hot_f + l;
			if (sc > best) best = sc, best_j = hot;
		} else hot_f = 0, hot = -1;
		if (i - win > p.max_iter) win = i - p.max_iter;
		for (int64_t j = i - 1; j >= win; --j) {
			int32_t sc = d_link_score_a(ai, a[j], p);
			if (sc == D_INCOMPAT) continue;
			sc += f[j];
			if (sc > best) {
				best = sc, best_j = j;
				if (skipped > 0) --skipped;
			} else if (mark[j] == me) {
				if (++skipped > p.max_skip) break;
			}
			const int32_t pj = pred[j];
			if (pj >= 0) mark[q0 + pj] = me;
		}
		f[i] = best, pred[i] = best_j < 0 ? -1 : (int32_t)(best_j - q0);
		if (hot_f < best) hot_f = best, hot = i;
	}
}

__global__ __launch_bounds__(256) void k_seed_fill(int64_t n, int32_t kmer, int32_t *f, int32_t *pred, int32_t *mark, uint32_t *flag)
{
	MPA_SHORT_KERNEL();
	const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
	if (i < n) f[i] = kmer, pred[i] = -1, mark[i] = -1, flag[i] = 0;
}

__global__ __launch_bounds__(256) void k_seed_compact(const uint64_t *key, const uint32_t *val, int64_t n, int nb, const int64_t *qfirst, const uint32_t *flag,
                                                      const uint32_t *idx, const int32_t *f, const int32_t *pred, int32_t *o_pos, int32_t *o_f, int32_t *o_pred, uint64_t *o_a)
{
	MPA_SHORT_KERNEL();
	const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
	if (i >= n || !flag[i]) return;
	const uint64_t k = key[i];
	const uint32_t d = idx[i];
	const int64_t q0 = qfirst[k >> nb];
	o_pos[d] = (int32_t)(i - q0);
	// the predecessor as an index into the query's part of the compacted view (a chained-onto anchor is always in the view)
	const int32_t pr = pred[i];
	o_f[d] = f[i], o_pred[d] = pr < 0 ? -1 : (int32_t)(idx[q0 + pr] - idx[q0]);
	o_a[d] = (k & ((1ULL << nb) - 1)) << 32 | val[i];
}

// compact offset of every query's first anchor (+ the total)
__global__ __launch_bounds__(256) void k_seed_bounds(const int64_t *qfirst, int32_t n_query, int64_t n, const uint32_t *idx, const uint32_t *flag, int64_t *cfirst)
{
	MPA_SHORT_KERNEL();
	const int32_t q = blockIdx.x * 256 + threadIdx.x;
	if (q > n_query) return;
	const int64_t i = qfirst[q];
	cfirst[q] = i < n ? (int64_t)idx[i] : (n > 0 ? (int64_t)idx[n - 1] + flag[n - 1] : 0);
}

// ------------------------------------------------------------------------------------------------
// Refinement scan (map.c:97-111 = mp_sketch_nt4 + mp_sketch_clean_orf at k = kmer2, every k-mer, base resolution): which
// positions of a region's window end a k-mer that the query also has?  One workgroup per 2048-position chunk of a window;
// the query's k-mer words sit in an LDS hash set, the chunk's bases (with a halo wide enough to decide the minimum ORF
// length) in LDS as well.  A thread forms the k-mer that ENDS at its position directly from five codons -- no rolling
// state, so chunks are independent.  Hits go to one global list (window, position, hash); the host groups and pairs them.
// ------------------------------------------------------------------------------------------------
struct RefineWindowDev { int64_t as; int32_t qid, vid, len, pad; };
struct RefineChunk { int32_t win, start; };
struct RefineTab { uint8_t t[64]; };                  // codon -> reduced residue (ns_tab_codon13), 0xff for a stop codon
#define REFINE_CHUNK 2048
#define REFINE_HALO 112

__device__ __forceinline__ uint32_t d_hash32_mask(uint32_t key, uint32_t mask)     // mp_hash32_mask (sketch.c:7-16)
{
	key = (key + ~(key << 15)) & mask;
	key ^= key >> 10;
	key = (key + (key << 3)) & mask;
	key ^= key >> 6;
	key = (key + ~(key << 11)) & mask;
	key ^= key >> 16;
	return key;
}

__global__ __launch_bounds__(256) void k_refine_scan(DevGenome g, const RefineWindowDev *wins, const RefineChunk *chunks, const int64_t *qw_first, const uint32_t *qwords,
                                                     RefineTab rt, int32_t kmer, int32_t min_aa_len, int32_t hs_log2, uint4 *hits, unsigned long long *n_hits, unsigned long long cap)
{
	MPA_SHORT_KERNEL();
	extern __shared__ uint32_t lds_refine[];
	const int32_t HS = 1 << hs_log2;
	uint32_t *table = lds_refine;                                   // [HS] open addressing, 0xffffffff = empty
	uint8_t *base = (uint8_t*)(table + HS);                         // [REFINE_CHUNK + 2 * REFINE_HALO] nt4 codes, 15 = outside the window
	__shared__ uint8_t tab[64];                                    // codon -> reduced residue, 0xff = stop
	const RefineChunk ch = chunks[blockIdx.x];
	const RefineWindowDev w = wins[ch.win];
	const int cid = w.vid >> 1, rev = w.vid & 1;
	const int64_t off = g.ctg_off[cid], clen = g.ctg_len[cid];
	if (threadIdx.x < 64) tab[threadIdx.x] = rt.t[threadIdx.x];
	for (int k = threadIdx.x; k < HS; k += 256) table[k] = 0xffffffffu;
	for (int k = threadIdx.x; k < REFINE_CHUNK + 2 * REFINE_HALO; k += 256) {
		const int64_t p = (int64_t)ch.start - REFINE_HALO + k;        // window-local
		base[k] = (p < 0 || p >= w.len) ? 15 : (uint8_t)strand_base(g.seq, off, clen, rev, w.as + p);
	}
	__syncthreads();
	for (int64_t k = qw_first[w.qid] + threadIdx.x; k < qw_first[w.qid + 1]; k += 256) {
		const uint32_t word = qwords[k];
		uint32_t slot = (word * 2654435761u) >> (32 - hs_log2);
		for (;;) {
			const uint32_t old = atomicCAS(&table[slot], 0xffffffffu, word);
			if (old == 0xffffffffu || old == word) break;
			slot = (slot + 1) & (HS - 1);
		}
	}
	__syncthreads();
	const uint32_t mask = (1u << (4 * kmer)) - 1;
	auto codon_at = [&](int e) -> uint32_t {                          // reduced residue of the codon whose last base is LDS index e; 0xff if none
		const uint32_t b0 = base[e - 2], b1 = base[e - 1], b2 = base[e];
		if ((b0 | b1 | b2) > 3) return 0xffu;
		return tab[b0 << 4 | b1 << 2 | b2];
	};
	for (int t = 0; t < REFINE_CHUNK / 256; ++t) {
		const int32_t pos = ch.start + t * 256 + (int32_t)threadIdx.x;   // window-local position of the k-mer's last base
		if (pos >= w.len) continue;
		const int e = pos - ch.start + REFINE_HALO;
		uint32_t word = 0;
		bool ok = true;
		for (int c = kmer - 1; c >= 0; --c) {
			const uint32_t r = codon_at(e - 3 * c);
			if (r == 0xffu) { ok = false; break; }
			word = word << 4 | r;
		}
		if (!ok) continue;
		word &= mask;
		bool found = false;
		for (uint32_t slot = (word * 2654435761u) >> (32 - hs_log2);; slot = (slot + 1) & (HS - 1)) {
			const uint32_t v = table[slot];
			if (v == word) { found = true; break; }
			if (v == 0xffffffffu) break;
		}
		if (!found) continue;
		// the open reading frame around the k-mer must be at least min_aa_len codons long (sketch.c:64-100)
		int32_t n = kmer;
		for (int q = e - 3 * kmer; n < min_aa_len && q >= 2 && codon_at(q) != 0xffu; q -= 3) ++n;
		for (int q = e + 3; n < min_aa_len && q < REFINE_CHUNK + 2 * REFINE_HALO && codon_at(q) != 0xffu; q += 3) ++n;
		if (n < min_aa_len) continue;
		const unsigned long long at = atomicAdd(n_hits, 1ULL);
		if (at < cap) hits[at] = make_uint4((uint32_t)ch.win, (uint32_t)pos, d_hash32_mask(word, mask), 0u);
	}
}

// ------------------------------------------------------------------------------------------------
// Index build on the device (mp_idx_build: build_worker + build_bidx, index.c:52-136; mp_sketch_nt4 + mp_sketch_clean_orf,
// sketch.c:40-100): every selected k-mer of every reading frame of both strands of every contig as (bucket << 32 | global
// block id); sorted and de-duplicated these keys ARE the index -- kb[] is their low words (inside a bucket ascending global
// block id = contig/strand order, then position, which is the reference's layout) and ki[] the bucket boundaries.
// k_index_scan: one workgroup per 2 048 positions of a strand (bases + halo in LDS); a thread forms the k-mer that ends at
// its position from `kmer` codons, hashes it, applies the modimizer test, and checks that the open reading frame around it
// has at least min_aa_len codons by walking its frame both ways (an ORF ends at a stop codon, an ambiguous base or the
// contig end).  Two passes over the genome: count per chunk, exclusive scan, then emit at exact offsets.
// ------------------------------------------------------------------------------------------------
struct IndexScanArgs {
	DevGenome g;
	const int64_t *chunk_first;     // [2 n_ctg + 1] first chunk of every strand
	const uint32_t *bo;             // [2 n_ctg] block offset of every strand
	int32_t n_strand, kmer, mod_bit, bbit, min_aa_len, halo;
	RefineTab rt;
};

template<bool EMIT>
__global__ __launch_bounds__(256) void k_index_scan(IndexScanArgs a, uint32_t *count, const uint64_t *offset, uint64_t *keys)
{
	extern __shared__ uint32_t lds_index[];
	uint8_t *base = (uint8_t*)lds_index;                            // [REFINE_CHUNK + 2 halo] nt4 codes, 15 = outside the contig
	__shared__ uint8_t tab[64];
	__shared__ uint32_t n_here;
	const int64_t chunk = blockIdx.x;
	int32_t lo = 0, hi = a.n_strand - 1;
	while (lo < hi) { const int32_t mid = (lo + hi + 1) >> 1; if (a.chunk_first[mid] <= chunk) lo = mid; else hi = mid - 1; }
	const int32_t strand = lo, cid = strand >> 1, rev = strand & 1;
	const int64_t off = a.g.ctg_off[cid], clen = a.g.ctg_len[cid];
	const int64_t start = (chunk - a.chunk_first[strand]) * REFINE_CHUNK;
	const int32_t halo = a.halo, span = REFINE_CHUNK + 2 * halo;
	if (threadIdx.x < 64) tab[threadIdx.x] = a.rt.t[threadIdx.x];
	if (threadIdx.x == 0) n_here = 0;
	for (int k = threadIdx.x; k < span; k += 256) {
		const int64_t p = start - halo + k;
		base[k] = (p < 0 || p >= clen) ? 15 : (uint8_t)strand_base(a.g.seq, off, clen, rev, p);
	}
	__syncthreads();
	const uint32_t mask = (1u << (4 * a.kmer)) - 1, mask_mod = (1u << a.mod_bit) - 1;
	auto codon_at = [&](int e) -> uint32_t {                          // reduced residue of the codon whose last base is LDS index e; 0xff if none
		const uint32_t b0 = base[e - 2], b1 = base[e - 1], b2 = base[e];
		if ((b0 | b1 | b2) > 3) return 0xffu;
		return tab[b0 << 4 | b1 << 2 | b2];
	};
	const uint64_t out0 = EMIT ? offset[chunk] : 0;
	for (int t = 0; t < REFINE_CHUNK / 256; ++t) {
		const int64_t pos = start + t * 256 + (int64_t)threadIdx.x;     // strand-local position of the k-mer's last base
		if (pos >= clen) continue;
		const int e = (int)(pos - start) + halo;
		uint32_t word = 0;
		bool ok = true;
		for (int c = a.kmer - 1; c >= 0; --c) {
			const uint32_t r = codon_at(e - 3 * c);
			if (r == 0xffu) { ok = false; break; }
			word = word << 4 | r;
		}
		if (!ok) continue;
		const uint32_t h = d_hash32_mask(word & mask, mask);
		if (h & mask_mod) continue;
		int32_t n = a.kmer;
		for (int q = e - 3 * a.kmer; n < a.min_aa_len && q >= 2 && codon_at(q) != 0xffu; q -= 3) ++n;
		for (int q = e + 3; n < a.min_aa_len && q < span && codon_at(q) != 0xffu; q += 3) ++n;
		if (n < a.min_aa_len) continue;
		const uint32_t slot = atomicAdd(&n_here, 1u);
		if (EMIT) keys[out0 + slot] = (uint64_t)(h >> a.mod_bit) << 32 | (uint64_t)((uint32_t)(pos >> a.bbit) + a.bo[strand]);
	}
	if (!EMIT) {
		__syncthreads();
		if (threadIdx.x == 0) count[chunk] = n_here;
	}
}

// flag[i] = 1 where sorted key i differs from key i - 1
__global__ __launch_bounds__(256) void k_index_flag(const uint64_t *keys, int64_t n, uint32_t *flag)
{
	MPA_SHORT_KERNEL();
	const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
	if (i < n) flag[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
}
// the distinct keys: kb[] = their block ids, cnt[bucket] = how many per bucket
__global__ __launch_bounds__(256) void k_index_compact(const uint64_t *keys, int64_t n, const uint32_t *flag, const uint64_t *idx, uint32_t *kb, unsigned long long *cnt)
{
	MPA_SHORT_KERNEL();
	const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
	if (i >= n || !flag[i]) return;
	kb[idx[i]] = (uint32_t)keys[i];
	atomicAdd(&cnt[keys[i] >> 32], 1ULL);
}

} // namespace mpa
