// seed_exec.hip -- GPU side of the seeding stage, included by dp_exec.hip.
//
// For a whole mini-batch at once (drivers in dp_exec.hip):
//   k_seed_sift     the anchors of a query merged from its occurrence lists in (block, seed) order -- the order of
//                   radix_sort_mp64 over block << 32 | query position, map.c:163-178 -- and filtered in LDS: only anchors with
//                   another one in the same or an adjacent block survive (the only ones the pre-chain can link)
//   k_prechain_fwd  forward pass of the block-resolution pre-chain over the kept anchors    chain.c:112-151,181-209
//   scan + k_seed_compact   the linked anchors as a sparse view (position in the query's FULL sorted list, f, pred, anchor)
//   k_chain_extract / k_chain_fwd / k_chain_fwd_wave   pre-chain extraction, main chain, main-chain extraction (chain.c:8-110,160-219)
// The pre-chain only links anchors at most one 256-bp block apart, so the sorted anchors of a query fall into
// independent "runs" (maximal stretches in which consecutive blocks differ by at most max_dblock); ~97 % of the anchors
// at genome scale are runs of one.  One thread walks one run with exactly the reference's loop; the arithmetic of
// comput_sc() includes two float operations, compiled without FMA contraction like the host code.
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

namespace mpa {

struct U32ToU64 { __host__ __device__ uint64_t operator()(uint32_t x) const { return (uint64_t)x; } };   // scan inputs of 32-bit counts as 64-bit sums

struct SeedJobDev { int64_t kb_off, dst; int32_t cnt, qpos, qid, pad; };



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

// The value that travels with a kept anchor: query position | its rank in the query's full sorted anchor list << 32 (k_seed_sift).
__device__ __forceinline__ uint32_t qpos_of(uint64_t v) { return (uint32_t)v; }
__device__ __forceinline__ int32_t dropped_below(uint64_t v) { return (int32_t)(v >> 32); }

// one thread per anchor; the thread of a run's first anchor walks the run (mp_chain's forward loop, chain.c:181-209)
template<typename V>
__global__ __launch_bounds__(256) void k_prechain_fwd(const uint64_t *key, const V *val, int64_t n, int nb, const int64_t *qfirst, PreParams p,
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
		const uint32_t qi = qpos_of(val[i]);
		const int32_t me = (int32_t)(i - q0);
		int64_t best_j = -1;
		int32_t best = p.kmer, skipped = 0;
		while (win < i && (int64_t)((ki - key[win]) << p.bbit) > p.max_dist_x) ++win;
		if (hot >= 0 && hot >= win) {
			const int32_t l = d_link_score(ki, qi, key[hot], qpos_of(val[hot]), p);
			const int32_t sc = hot_f + l;                                  // (INT32_MIN + hot_f stays far below kmer, as on the host)
			if (sc > best) best = sc, best_j = hot;
		} else hot_f = 0, hot = -1;
		if (i - win > p.max_iter) win = i - p.max_iter;
		for (int64_t j = i - 1; j >= win; --j) {
			int32_t sc = d_link_score(ki, qi, key[j], qpos_of(val[j]), p);
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

// cnt != nullptr: problem q holds cnt[q] anchors from first[q] on (the slots up to first[q + 1] are unused).
// long_runs != nullptr: a run of more than max_serial anchors is not walked here but handed to k_chain_fwd_wave (a wavefront
// per run): one thread walking a gene locus of a few hundred anchors, window by window, is what the kernel used to wait for.
struct LongRun { int64_t s, e, q0; };
__global__ __launch_bounds__(256) void k_chain_fwd(const uint64_t *a, int64_t n, const int64_t *first, const int64_t *cnt, int32_t n_prob, PreParams p,
                                                   int32_t *f, int32_t *pred, int32_t *mark, int32_t max_serial = 0x7fffffff, LongRun *long_runs = nullptr,
                                                   unsigned int *n_long = nullptr, unsigned int long_cap = 0)
{
	MPA_SHORT_KERNEL();
	const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
	if (s >= n) return;
	// the problem this anchor belongs to: largest q with first[q] <= s
	int32_t lo = 0, hi = n_prob - 1;
	while (lo < hi) { const int32_t mid = (lo + hi + 1) >> 1; if (first[mid] <= s) lo = mid; else hi = mid - 1; }
	const int64_t q0 = first[lo], q1 = cnt ? q0 + cnt[lo] : first[lo + 1];
	if (s >= q1) return;                                                 // an unused slot
	const uint64_t xs = a[s] >> 32;
	const uint64_t max_dx = (uint64_t)(p.max_dist_x >> p.bbit);          // (x_i - x_j) << bbit > max_dist_x  <=>  x_i - x_j > max_dist_x >> bbit
	if (s > q0 && xs - (a[s - 1] >> 32) <= max_dx) return;               // not the head of its run
	int64_t e = s + 1;
	{
		uint64_t prev = xs;
		while (e < q1) { const uint64_t x = a[e] >> 32; if (x - prev > max_dx) break; prev = x, ++e; }
	}
	if (long_runs && e - s > (int64_t)max_serial) {
		const unsigned int at = atomicAdd(n_long, 1u);
		if (at < long_cap) { long_runs[at] = LongRun{ s, e, q0 }; return; }
		// (the list is full -- it is sized for the worst case, so this cannot happen -- walk the run here after all)
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

// ------------------------------------------------------------------------------------------------
// k_chain_extract: mp_chain_backtrack + compact_a (chain.c:8-110) on the device, one WAVEFRONT per chaining problem.
//
// The extraction is a sequential algorithm whose result depends on the order the reference's unstable radix sort leaves equal
// chain scores in; it is therefore not re-derived here but COMPILED from the host's own source (chain_core.h:
// chain_extract_core), which tests/test_host_core.py pins to the oracle.  A problem is the pre-chain of a query (sparse view
// of the anchors that k_prechain_fwd linked, set_only: the survivors as an ascending set), its main chain (dense view over
// those survivors: chains laid out and sorted by first target position) or a refinement chain.  The code runs on lane 0 of the
// wave (it is serial and latency-bound; what the device buys is that 4 000 problems run side by side and that no anchor has
// to travel to the host and back between the two chaining rounds).  Layout: the view of problem q sits at first[q] in every
// view / scratch / output array; status[q] = 1 when the problem needs the full list of chain ends and has no room for it
// (tiny or degenerate problems: the host takes the query over).
// ------------------------------------------------------------------------------------------------
struct ExtractArgs {
	const int64_t *first;            // [n_prob + 1] offset of every problem's view
	const int64_t *cnt;              // [n_prob] anchors in the view, or nullptr: first[q + 1] - first[q]
	const int64_t *ntot_first;       // sparse views: n_total of problem q = ntot_first[q + 1] - ntot_first[q]; nullptr: dense views (n_total = m)
	const int32_t *v_pos, *v_f, *v_pred;   // view arrays (v_pos == nullptr for dense views)
	const uint64_t *v_a;
	// scratch, all indexed like the views; ends / stack have 64 / 5 extra entries per problem
	int32_t *mark, *order;
	Pair64 *ends, *tail8;
	SparseItem *items, *moved, *merged;   // (the chain layout's scratch -- packed anchors, sorted u, first positions -- reuses a problem's moved / merged: dead by then)
	uint8_t *kept;
	SortRange *stack;
	// results
	uint64_t *a_out, *u_out;         // at first[q]
	int64_t *n_a, *n_u;              // [n_prob]
	int32_t *status;                 // [n_prob]
	ChainParams p;
	int32_t set_only;
};

// MPA_TIMING=2: wall-clock stamps of k_chain_extract's phases, 8 per problem (start, chain ends sorted, trees, extraction, end)
__device__ long long *g_extract_prof = nullptr;
__device__ int g_extract_prof_n = 0;         // problems the buffer has room for (launches of other contexts may have more)

// the team's fast scratch memory: 2 KB of LDS per wavefront (k_chain_extract is one wavefront per workgroup).  Round 6: 8 KB -> 2 KB.
// With the 5 KB of digit tables a wave then takes 7.2 KB, so that LDS allows the five waves per SIMD the 85 VGPRs do (8 KB: three), all
// 4 000 problems of a launch are resident at once and a wave's footprint next to the DP round's workgroups is half of what it was:
// lone launch 11.9 -> 10.6 ms, stream +5 % (3 of 3 interleaved repeats; 512 words: the same; profiles/r06_experiments.txt).  Buckets
// that do not fit are walked in place, as before.
#ifndef EXTRACT_STAGE_WORDS
#define EXTRACT_STAGE_WORDS 256
#endif
// (DYNAMIC LDS: with a static array the compiler knows that LDS allows three waves per SIMD and lets the registers grow to 512 / 3 --
// 166 VGPRs, a wave that fits next to no DP workgroup's waves; with the size hidden, amdgpu_waves_per_eu below is what it allocates for)
#define EXTRACT_LDS_BYTES (EXTRACT_STAGE_WORDS * 8 + 1280 * 4)
__device__ __forceinline__ uint64_t *g_extract_stage()
{
	extern __shared__ __attribute__((aligned(16))) uint64_t mpa_extract_lds[];
	return mpa_extract_lds;
}

// the team of chain_core.h on the device: the 64 lanes of one wavefront
struct CoopWave {
	static __device__ __forceinline__ int lane() { return (int)(threadIdx.x & 63); }
	static __device__ __forceinline__ int width() { return 64; }
	// lanes of one wave share their L1: ordering their global / LDS accesses needs no cache action, only completion + a barrier
	static __device__ __forceinline__ void sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
	static __device__ __forceinline__ uint64_t ballot(bool p) { return __ballot(p); }
	static __device__ __forceinline__ int rank(uint64_t m) { return __popcll(m & ((1ull << lane()) - 1ull)); }
	static __device__ __forceinline__ int popc(uint64_t m) { return __popcll(m); }
	static __device__ __forceinline__ int32_t reduce_max(int32_t v)
	{
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) { const int32_t w = __shfl_xor(v, o); v = w > v ? w : v; }
		return v;
	}
	static __device__ __forceinline__ bool any(bool p) { return __ballot(p) != 0; }
	static __device__ __forceinline__ void count(uint32_t *slot) { atomicAdd(slot, 1u); }
	static __device__ __forceinline__ void atomic_min(int32_t *slot, int32_t v) { atomicMin(slot, v); }
	static __device__ __forceinline__ int64_t scan_excl(int64_t v, int64_t *total)
	{
		int64_t inc = v;
#pragma unroll
		for (int o = 1; o < 64; o <<= 1) { const int64_t w = __shfl_up(inc, o); if (lane() >= o) inc += w; }
		*total = __shfl(inc, 63);
		return inc - v;
	}
	static __device__ __forceinline__ int64_t scan_max_excl(int64_t v, int64_t *total)
	{
		int64_t inc = v;
#pragma unroll
		for (int o = 1; o < 64; o <<= 1) { const int64_t w = __shfl_up(inc, o); if (lane() >= o && w > inc) inc = w; }
		*total = __shfl(inc, 63);
		const int64_t below = __shfl_up(inc, 1);
		return lane() ? below : INT64_MIN;
	}
	static __device__ __forceinline__ void digit_rank(int d, bool have, int *rank, int *cnt)
	{
		unsigned long long eq = __ballot(have);                  // lanes that take part and hold the same 8-bit digit: eight ballots
#pragma unroll
		for (int b = 0; b < 8; ++b) { const unsigned long long m = __ballot((d >> b) & 1); eq &= ((d >> b) & 1) ? m : ~m; }
		*rank = __popcll(eq & ((1ull << lane()) - 1ull)), *cnt = __popcll(eq);
	}
	static __device__ __forceinline__ int first_unset(uint64_t m) { return m == ~0ull ? 64 : __ffsll((long long)~m) - 1; }
	static __device__ __forceinline__ int lowest(uint64_t m) { return __ffsll((long long)m) - 1; }
	static __device__ __forceinline__ uint64_t *scratch(int64_t *cap) { *cap = EXTRACT_STAGE_WORDS; return g_extract_stage(); }
	static __device__ __forceinline__ void mark_time(int k) { if (g_extract_prof && (int)blockIdx.x < g_extract_prof_n && lane() == 0) g_extract_prof[(int64_t)blockIdx.x * 16 + k] = (long long)wall_clock64(); }
	static __device__ __forceinline__ void note(int k, int64_t v) { if (g_extract_prof && (int)blockIdx.x < g_extract_prof_n && lane() == 0) g_extract_prof[(int64_t)blockIdx.x * 16 + 8 + k] = (long long)v; }
};

// (8 waves per SIMD = a budget of 64 VGPRs: the compiler gets there with 2 spilled registers, where it takes 85 when asked for 4..5 waves and
// 166 when the LDS is static.  The walk is serial and waits for memory 84 % of its time: it does not miss the registers, and everything
// that has to share a SIMD with it -- the DP round's waves first of all -- gets them: stream +4..5 % in 9 of 9 interleaved repeats,
// profiles/r06_experiments.txt section 11)
#ifndef MPA_EXTRACT_WAVES
#define MPA_EXTRACT_WAVES 8
#endif
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(MPA_EXTRACT_WAVES))) void k_chain_extract(ExtractArgs x, int32_t n_prob)
{
	uint32_t *hist = (uint32_t*)(g_extract_stage() + EXTRACT_STAGE_WORDS);   // [1280]
	const int32_t q = blockIdx.x;
	if (q >= n_prob) return;
	const int64_t off = x.first[q];
	const int64_t m = x.cnt ? x.cnt[q] : x.first[q + 1] - off;
	if (m <= 0) { if (threadIdx.x == 0) x.n_a[q] = 0, x.n_u[q] = 0; return; }
	const int64_t n_total = x.ntot_first ? x.ntot_first[q + 1] - x.ntot_first[q] : m;
	const ChainView v{ n_total, m, x.v_pos ? x.v_pos + off : nullptr, x.v_f + off, x.v_pred + off, x.v_a + off };
	const int64_t eoff = off + 64 * (int64_t)q;
	// (packed: m words, u_sorted: m words -- inside the 3 m words of THIS problem's moved list; first: m pairs inside its merged list)
	const ExtractScratch S{ x.mark + off, x.order + off, x.ends + eoff, m + 64, x.items + off, x.moved + off, x.merged + off, x.tail8 + off, (Pair64*)(x.merged + off),
	                        x.kept + off, (uint64_t*)(x.moved + off), (uint64_t*)(x.moved + off) + m, x.stack + (eoff / 64 + 5 * (int64_t)q), hist };
	int64_t n_a = 0, n_u = 0;
	const int rc = chain_extract_core<CoopWave>(x.p, v, S, x.a_out + off, &n_a, x.u_out + off, &n_u, x.set_only != 0);
	if (threadIdx.x != 0) return;
	if (rc != 0) { x.status[q] = 1; x.n_a[q] = 0, x.n_u[q] = 0; return; }
	x.n_a[q] = n_a, x.n_u[q] = n_u;
}

// exclusive prefixes of two per-problem counts (one workgroup; a few thousand problems), + the totals at [n]
__global__ __launch_bounds__(256) void k_offsets2(const int64_t *c0, const int64_t *c1, int32_t n, int64_t *o0, int64_t *o1)
{
	__shared__ int64_t wsum[8];
	const int t = threadIdx.x;
	int64_t run0 = 0, run1 = 0;
	for (int32_t base = 0; base < n; base += 256) {
		const int32_t i = base + t;
		const int64_t v0 = i < n ? c0[i] : 0, v1 = i < n ? c1[i] : 0;
		int64_t i0 = v0, i1 = v1;
#pragma unroll
		for (int o = 1; o < 64; o <<= 1) {
			const int64_t u0 = __shfl_up(i0, o), u1 = __shfl_up(i1, o);
			if ((t & 63) >= o) i0 += u0, i1 += u1;
		}
		__syncthreads();
		if ((t & 63) == 63) wsum[t >> 6] = i0, wsum[4 + (t >> 6)] = i1;
		__syncthreads();
		int64_t b0 = 0, b1 = 0;
		for (int w = 0; w < (t >> 6); ++w) b0 += wsum[w], b1 += wsum[4 + w];
		if (i < n) o0[i] = run0 + b0 + i0 - v0, o1[i] = run1 + b1 + i1 - v1;
		run0 += wsum[0] + wsum[1] + wsum[2] + wsum[3], run1 += wsum[4] + wsum[5] + wsum[6] + wsum[7];
	}
	if (t == 0) o0[n] = run0, o1[n] = run1;
}
// the chains of every problem, densely: u (score << 32 | anchors) and their anchors; dst may be pinned host memory
__global__ __launch_bounds__(256) void k_chain_pack(const int64_t *first, const int64_t *n_a, const int64_t *n_u, const int64_t *o_a, const int64_t *o_u,
                                                    const uint64_t *a_src, const uint64_t *u_src, uint64_t *a_dst, uint64_t *u_dst)
{
	MPA_SHORT_KERNEL();
	const int32_t q = blockIdx.x;
	const int64_t off = first[q], na = n_a[q], nu = n_u[q], da = o_a[q], du = o_u[q];
	for (int64_t i = threadIdx.x; i < na; i += 256) a_dst[da + i] = a_src[off + i];
	for (int64_t i = threadIdx.x; i < nu; i += 256) u_dst[du + i] = u_src[off + i];
}

// The forward pass of mp_chain (chain.c:181-209) over ONE run, by one wavefront: the candidates j = i-1, i-2, ... of anchor i are
// scored 64 at a time, one per lane.  What the reference does sequentially over j is recovered exactly:
//   * "sc > best" with a running best is an exclusive prefix maximum over the lanes (seeded with the best so far): strict
//     improvement, the nearest candidate wins ties -- as in the loop;
//   * t[p[j]] = i marks set by candidates visited EARLIER can only land on candidates visited LATER (p[j] < j), so every
//     compatible lane writes its mark first, and after a wave-level fence each lane reads its own;
//   * the max_skip counter (++ on a marked non-improving candidate, -- on an improving one, clamped at 0, stop when it exceeds
//     max_skip) is replayed by scalar code over the two ballots; candidates behind the stop are dropped.  Marks written by
//     dropped lanes carry the value i and are never read again (the next anchor compares with i + 1).
__global__ __launch_bounds__(64) void k_chain_fwd_wave(const uint64_t *a, const LongRun *runs, const unsigned int *n_runs, unsigned int cap, PreParams p,
                                                       int32_t *f, int32_t *pred, int32_t *mark)
{
	const unsigned int nr = *n_runs < cap ? *n_runs : cap;
	const int lane = (int)(threadIdx.x & 63);
	for (unsigned int r = blockIdx.x; r < nr; r += gridDim.x) {
	const LongRun run = runs[r];
	const int64_t q0 = run.q0;
	const uint64_t max_dx = (uint64_t)(p.max_dist_x >> p.bbit);
	int64_t win = run.s, hot = -1;
	int32_t hot_f = 0;
	for (int64_t i = run.s; i < run.e; ++i) {
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
		bool stopped = false;
		for (int64_t jhi = i - 1; jhi >= win && !stopped; jhi -= 64) {
			const int64_t j = jhi - lane;
			const bool valid = j >= win;
			int32_t sc = valid ? d_link_score_a(ai, a[j], p) : D_INCOMPAT;
			const bool compat = sc != D_INCOMPAT;
			int32_t pj = -1;
			if (compat) {
				sc += f[j];
				pj = pred[j];
				if (pj >= 0) mark[q0 + pj] = me;
			}
			__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
			__builtin_amdgcn_wave_barrier();
			const bool marked = compat && mark[j] == me;
			// exclusive prefix maximum of the compatible scores, in visiting order (lane 0 first)
			int32_t pm = compat ? sc : INT32_MIN;
#pragma unroll
			for (int o = 1; o < 64; o <<= 1) { const int32_t w = __shfl_up(pm, o); if (lane >= o) pm = w > pm ? w : pm; }
			int32_t before = __shfl_up(pm, 1);
			if (lane == 0) before = INT32_MIN;
			if (best > before) before = best;
			const bool improve = compat && sc > before;
			const unsigned long long I = __ballot(improve), M = __ballot(compat && !improve && marked);
			// the skip counter over the candidates in order
			unsigned long long ev = I | M, live = ~0ull;
			while (ev) {
				const int b = __ffsll((long long)ev) - 1;
				ev &= ev - 1;
				if ((I >> b) & 1ull) { if (skipped > 0) --skipped; }
				else if (++skipped > p.max_skip) { stopped = true; live = b >= 63 ? ~0ull : ((2ull << b) - 1ull); break; }
			}
			const unsigned long long Iv = I & live;
			if (Iv) {
				const int last = 63 - __clzll((long long)Iv);
				best = __shfl(sc, last), best_j = jhi - last;
			}
		}
		if (lane == 0) f[i] = best, pred[i] = best_j < 0 ? -1 : (int32_t)(best_j - q0);
		if (hot_f < best) hot_f = best, hot = i;
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
		__builtin_amdgcn_wave_barrier();
	}
	}
}

__global__ __launch_bounds__(256) void k_seed_fill(int64_t n, int32_t kmer, int32_t *f, int32_t *pred, int32_t *mark, uint32_t *flag)
{
	MPA_SHORT_KERNEL();
	const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
	if (i < n) f[i] = kmer, pred[i] = -1, mark[i] = -1, flag[i] = 0;
}

// ABS_POS: the high half of the value IS the anchor's rank in the query's full sorted list (k_seed_sift).  The outputs may be
// pinned host memory: every element is written once.
template<typename V, bool ABS_POS = false>
__global__ __launch_bounds__(256) void k_seed_compact(const uint64_t *key, const V *val, int64_t n, int nb, const int64_t *qfirst, const uint32_t *flag,
                                                      const uint32_t *idx, const int32_t *f, const int32_t *pred, int32_t *o_pos, int32_t *o_f, int32_t *o_pred, uint64_t *o_a)
{
	MPA_SHORT_KERNEL();
	const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
	if (i >= n || !flag[i]) return;
	const uint64_t k = key[i];
	const uint32_t d = idx[i];
	const int64_t q0 = qfirst[k >> nb];
	o_pos[d] = ABS_POS ? dropped_below(val[i]) : (int32_t)(i - q0) + dropped_below(val[i]);   // position in the query's FULL sorted anchor list
	// the predecessor as an index into the query's part of the compacted view (a chained-onto anchor is always in the view)
	const int32_t pr = pred[i];
	o_f[d] = f[i], o_pred[d] = pr < 0 ? -1 : (int32_t)(idx[q0 + pr] - idx[q0]);
	o_a[d] = (k & ((1ULL << nb) - 1)) << 32 | qpos_of(val[i]);
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

__device__ __forceinline__ int32_t d_lower_bound(const uint32_t *a, int32_t n, uint64_t x)   // first index with a[i] >= x
{
	int32_t lo = 0, hi = n;
	while (lo < hi) { const int32_t mid = (lo + hi) >> 1; if ((uint64_t)a[mid] < x) lo = mid + 1; else hi = mid; }
	return lo;
}

// ------------------------------------------------------------------------------------------------
// k_seed_sift: anchors of a query, merged from its occurrence lists IN ORDER, isolated ones dropped on the way (default seeding
// path; it replaced rounds 1-2's gather + global radix sort of every anchor).
//
// The anchors of a query (map.c:163-177) are the union of its seeds' occurrence lists, each of which is ascending in the block
// id, and the order the reference sorts them into (map.c:178, radix_sort_mp64 on block << 32 | query position) is (block, seed)
// with the seeds in ascending query position -- the order of the query's jobs.  One workgroup owns a SEGMENT of a query's block
// space (a whole query when it has fewer than ~64 k anchors) and walks it in ascending ranges with one cursor per list: the
// anchors of a range -- at most SIFT_CAP, the range is halved and read again when they do not fit -- go to LDS as
// (block - range start + 1) << LB | list in one pass over the lists, are sorted there (buckets by the key's leading bits, then
// each key counts the smaller keys of its bucket), and every anchor that has another one in its own or in an adjacent block (the only ones the
// pre-chain, map.c:186-192 with max_dist_x = one block, can link) is written out at its final sorted position together with
// its rank in the query's FULL sorted list (which the host's replay of the reference's unstable chain-end sort needs,
// host_core.cpp sorted_chain_ends_sparse).  Anchors in the last block of a range wait for the next range (the "carry"): whether
// they have a right neighbour is not known yet.  Segments are independent: the first cursor positions come from one binary
// search per list, which also gives the number of the query's anchors below the segment (the rank base) and whether block
// lo - 1 / block hi hold an anchor.  A query whose anchors of ONE block (plus carry) exceed the buffer -- thousands of seeds
// hitting the same block: degenerate low-complexity input -- is flagged and seeded on the host.
// ------------------------------------------------------------------------------------------------
#define SIFT_THREADS 256
#define SIFT_CARRY_MAX 1024                               /* anchors that may wait for the next range */
/* buckets of the in-LDS sort: SIFT_CAP / 4 */
#define SIFT_BUCKET_MAX 512u                              /* a larger bucket sends the range to the bitonic network */
// SIFT_CAP (template parameter CAP): anchors per LDS buffer -- 2048 (8 KB) or 4096 (16 KB); a range is sized for 5/8 of it
struct SiftSeg { int32_t qid; uint32_t lo, hi; int32_t cur_off; }; // blocks [lo, hi) of query qid; cur_off: the segment's own cursors (one per list of the query)

__device__ __forceinline__ uint32_t sift_wave_sum(uint32_t v)
{
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
	return v;
}
// sum over the workgroup (4 waves); scr = 4 words of LDS
__device__ __forceinline__ uint32_t sift_block_sum(uint32_t v, volatile uint32_t *scr)
{
	v = sift_wave_sum(v);
	__syncthreads();                                        // (scr may still be read from the previous call)
	if ((threadIdx.x & 63) == 0) scr[threadIdx.x >> 6] = v;
	__syncthreads();
	return scr[0] + scr[1] + scr[2] + scr[3];
}
// exclusive prefix of v over the workgroup's threads; scr = 4 words of LDS
__device__ __forceinline__ uint32_t sift_block_scan(uint32_t v, volatile uint32_t *scr)
{
	const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	uint32_t inc = v;
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) {
		const uint32_t n = __shfl_up(inc, o);
		if (lane >= (uint32_t)o) inc += n;
	}
	__syncthreads();                                        // (scr may still be read from the previous call)
	if (lane == 63) scr[wv] = inc;
	__syncthreads();
	uint32_t ex = inc - v;
	for (uint32_t w = 0; w < wv; ++w) ex += scr[w];
	return ex;
}
// skey/sval: staging.  A query with few anchors owns as many slots as it has anchors (a segment writes its kept anchors from
// sfirst[q] + rank base on: the kept ones can never outnumber the anchors below them); a query with more than SIFT_FULL_STAGE
// anchors owns HALF as many -- at genome scale one anchor in eight is kept -- laid out the same way with every offset halved, and
// a segment that keeps more than half of its anchors (a degenerate, repeat-only query) flags the query for the host instead of
// writing past its share.  seg_kept[seg] / seg_base[seg]: how many and from where.
#define SIFT_FULL_STAGE 16384
__host__ __device__ inline int64_t sift_stage_slots(int64_t n_anchor) { return n_anchor > SIFT_FULL_STAGE ? (n_anchor + 1) / 2 : n_anchor; }
// One step of a list's cursor: the next W entries (fewer at the list's end: m of them, the last one repeated) are read at once
// (sift_fetch); those below block hi take slots of the range's buffer as (block - lo + 1) << LB | l (sift_place, which returns
// how many there were -- W: the list may hold more of them).
template<int W>
__device__ __forceinline__ void sift_fetch(uint32_t (&v)[W], const uint32_t *list, int32_t c, int32_t cnt)
{
	const int32_t m = cnt - c < W ? cnt - c : W;
#pragma unroll
	for (int k = 0; k < W; ++k) v[k] = list[c + (k < m ? k : m - 1)];
}
template<int W>
__device__ __forceinline__ int32_t sift_place(const uint32_t (&v)[W], int32_t c, int32_t cnt, uint32_t lo, uint32_t hi, int LB, uint32_t l, uint32_t *keys, uint32_t *fill, uint32_t cap)
{
	const int32_t m = cnt - c < W ? cnt - c : W;
	int32_t below = 0;
#pragma unroll
	for (int k = 0; k < W; ++k) below += (k < m && v[k] < hi) ? 1 : 0;
	if (below > 0) {
		const uint32_t slot = atomicAdd(fill, (uint32_t)below);
#pragma unroll
		for (int k = 0; k < W; ++k)
			if (k < below && slot + k < cap) keys[slot + k] = ((v[k] - lo + 1u) << LB) | l;
	}
	return below;
}
template<int W>
__device__ __forceinline__ int32_t sift_take(const uint32_t *list, int32_t c, int32_t cnt, uint32_t lo, uint32_t hi, int LB, uint32_t l, uint32_t *keys, uint32_t *fill, uint32_t cap)
{
	uint32_t v[W];
	sift_fetch<W>(v, list, c, cnt);
	return sift_place<W>(v, c, cnt, lo, hi, LB, l, keys, fill, cap);
}
template<int SIFT_CAP>
// (round 6 measured the kernel with dynamic LDS and a budget of 96 / 80 registers -- 16 / 31 spilled --: the stream is level with all three,
// profiles/r06_experiments.txt section 10; the 128 registers and the static arrays stay)
__global__ __launch_bounds__(SIFT_THREADS) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_seed_sift(const SiftSeg *segs, const SeedJobDev *jobs, const int64_t *jfirst, const int64_t *qfirst, const int64_t *sfirst, const uint32_t *kb,
                                                            uint32_t n_block, int nb, int32_t *cur, int32_t *cur2, uint64_t *skey, uint64_t *sval, uint32_t *seg_kept,
                                                            int64_t *seg_base, int32_t *qflag)
{
	MPA_SHORT_KERNEL();
	constexpr uint32_t SIFT_TARGET = (uint32_t)SIFT_CAP * 5u / 8u;      // anchors a range is sized for
	constexpr int SIFT_NBK_LOG2 = SIFT_CAP >= 4096 ? 10 : 9, SIFT_NBK = 1 << SIFT_NBK_LOG2;
	__shared__ uint32_t keys[SIFT_CAP], kbuf[SIFT_CAP];        // a range's anchors; the same in bucket order
	__shared__ uint32_t bcnt[SIFT_NBK];                        // bucket sizes, then starts / ends
	__shared__ uint32_t bmax;
	__shared__ uint32_t scr[4];
	__shared__ uint32_t wtot[SIFT_CAP / SIFT_THREADS * 4];      // kept keys per chunk of 256 and wave
	__shared__ uint32_t fill;                                  // slots of `keys` taken while a range is loaded
	const int t = threadIdx.x;
	const SiftSeg sg = segs[blockIdx.x];
	const int32_t q = sg.qid;
	const int64_t j0 = jfirst[q];
	const int32_t nl = (int32_t)(jfirst[q + 1] - j0);
	const int64_t n_anchor = qfirst[q + 1] - qfirst[q];
	int LB = 1;
	while ((1 << LB) < nl) ++LB;
	const uint32_t lmask = (1u << LB) - 1u;
	const uint32_t w_max = (LB >= 31 ? 1u : (1u << (32 - LB))) - 2u;           // widest range whose local block numbers fit next to LB list bits
	const uint32_t lo0 = sg.lo, hi0 = sg.hi;
	const int64_t c0s = sg.cur_off;                            // this segment's cursors: segments of one query run side by side

	// ---- cursors at the segment's first block; rank base; is there an anchor in block lo0 - 1?
	const bool half_stage = n_anchor > SIFT_FULL_STAGE;      // (see sift_stage_slots)
	// The first 256 lists (all of them for most queries) live in their threads' registers: the job, the cursor, and the NEXT eight
	// entries, fetched as soon as the cursor is known -- a list sits at a random place of a multi-GB array, and the round trip of
	// its first touch in a range (TLB miss included) then runs behind the sort and the output of the range before.
	SeedJobDev s0;
	s0.kb_off = 0, s0.cnt = 0;
	int32_t c_reg = 0;
	uint32_t pre[8];
	uint32_t below = 0, left_nb = 0, inside = 0;
	for (int32_t l = t; l < nl; l += SIFT_THREADS) {
		const SeedJobDev s = jobs[j0 + l];
		const uint32_t *list = kb + s.kb_off;
		int32_t c = 0;
		if (lo0 > 0) {
			c = d_lower_bound(list, s.cnt, (uint64_t)lo0);
			if (c > 0 && list[c - 1] == lo0 - 1) left_nb = 1;
		}
		cur[c0s + l] = c;
		if (l == t) s0 = s, c_reg = c;
		below += (uint32_t)c;
		if (half_stage) inside += (uint32_t)((hi0 < n_block ? d_lower_bound(list, s.cnt, (uint64_t)hi0) : s.cnt) - c);
	}
	const uint32_t pos_base = sift_block_sum(below, scr);
	left_nb = sift_block_sum(left_nb, scr) ? 1u : 0u;
	// where this segment's kept anchors start in the staging arrays, and how many of them fit there
	uint32_t stage_cap = 0xffffffffu;
	int64_t out0 = sfirst[q] + (int64_t)pos_base;
	if (half_stage) {
		const uint32_t seg_total = sift_block_sum(inside, scr);
		out0 = sfirst[q] + (int64_t)(pos_base >> 1);
		stage_cap = ((pos_base + seg_total) >> 1) - (pos_base >> 1);
	}

	if (c_reg < s0.cnt) sift_fetch<8>(pre, kb + s0.kb_off, c_reg, s0.cnt);
	uint32_t lo = lo0, n_carry = 0, kept_total = 0, emitted = pos_base;
	uint32_t prev_blk = left_nb ? lo0 - 1 : 0xffffffffu;      // block of the last anchor below the buffer (0xffffffff: none that matters)
	// width of the first range: SIFT_TARGET anchors if the query's anchors were spread evenly over the genome
	uint32_t width;
	{
		const uint64_t wv = n_anchor > 0 ? (uint64_t)n_block * SIFT_TARGET / (uint64_t)n_anchor : (uint64_t)n_block;
		width = (uint32_t)(wv < 1 ? 1 : wv > w_max ? w_max : wv);
	}
	while (lo < hi0) {
		// ---- the range [lo, hi): its anchors go to LDS as (block - lo + 1) << LB | list, behind the carry (keys[0, n_carry), local
		// block 0), in ONE pass over the lists -- eight entries per dependent step, one slot request per step; a range whose
		// anchors do not fit is halved and read again (rare: the width follows the density)
		uint32_t hi, n;
		int32_t c_new0 = c_reg;
		for (;;) {
			hi = hi0 - lo > width ? lo + width : hi0;
			if (t == 0) fill = n_carry;
			__syncthreads();
			for (int32_t l0 = 0; l0 < nl; l0 += SIFT_THREADS) {
				const int32_t l = l0 + t;
				SeedJobDev s = s0;
				int32_t c = c_reg;
				if (l0 > 0) {
					s.kb_off = 0, s.cnt = 0, c = 0;
					if (l < nl) s = jobs[j0 + l], c = cur[c0s + l];
				}
				const uint32_t *list = kb + s.kb_off;
				// (the first step of a register-resident list was fetched while the range before was sorted; the steps after it are 16
				// entries wide: a few frequent k-mers hold hundreds of a range's anchors and the slowest thread sets the pace)
				if (c < s.cnt) {
					uint32_t v[8];
					if (l0 == 0) {
#pragma unroll
						for (int k = 0; k < 8; ++k) v[k] = pre[k];
					} else sift_fetch<8>(v, list, c, s.cnt);
					int32_t b = sift_place<8>(v, c, s.cnt, lo, hi, LB, (uint32_t)l, keys, &fill, SIFT_CAP);
					c += b;
					if (b == 8)
						while (c < s.cnt) {
							b = sift_take<16>(list, c, s.cnt, lo, hi, LB, (uint32_t)l, keys, &fill, SIFT_CAP);
							c += b;
							if (b < 16) break;
						}
				}
				if (l < nl) cur2[c0s + l] = c;
				if (l0 == 0) c_new0 = c;
			}
			__syncthreads();
			n = fill;
			if (n <= (uint32_t)SIFT_CAP) break;
			if (hi - lo <= 1) {                                    // one block alone does not fit: the host seeds this query
				if (t == 0) { atomicExch(&qflag[q], 1); seg_kept[blockIdx.x] = 0; seg_base[blockIdx.x] = out0; }
				return;
			}
			width = (hi - lo) >> 1;
			__syncthreads();                                       // (fill has been read)
		}
		for (int32_t l = t; l < nl; l += SIFT_THREADS) cur[c0s + l] = cur2[c0s + l];
		c_reg = c_new0;
		if (c_reg < s0.cnt && hi < hi0) sift_fetch<8>(pre, kb + s0.kb_off, c_reg, s0.cnt);       // the next range's first step
		const uint32_t total = n - n_carry;
		// ---- sort keys[0, n): into buckets by the leading bits (in arrival order), then every key counts the smaller keys of its
		// bucket -- a dozen instructions per key where a bitonic network spends a few hundred (keys are unique: no ties)
		{
			int sh = 32 - __clz((int)(((hi - lo + 1u) << LB) - 1u)) - SIFT_NBK_LOG2;      // every key is below (hi - lo + 1) << LB
			if (sh < 0) sh = 0;
			constexpr int BPT = SIFT_NBK / SIFT_THREADS;           // counters a thread owns in the scan
#pragma unroll
			for (int k = 0; k < BPT; ++k) bcnt[t * BPT + k] = 0u;
			if (t == 0) bmax = 0u;
			__syncthreads();
			for (uint32_t i = t; i < n; i += SIFT_THREADS) atomicAdd(&bcnt[keys[i] >> sh], 1u);
			__syncthreads();
			uint32_t c[BPT], mine = 0, big = 0;
#pragma unroll
			for (int k = 0; k < BPT; ++k) c[k] = bcnt[t * BPT + k], mine += c[k], big = c[k] > big ? c[k] : big;
			if (big > SIFT_BUCKET_MAX) atomicMax(&bmax, big);
			uint32_t at = sift_block_scan(mine, scr);               // (barriers inside)
#pragma unroll
			for (int k = 0; k < BPT; ++k) bcnt[t * BPT + k] = at, at += c[k];
			__syncthreads();
			if (bmax <= SIFT_BUCKET_MAX) {
				for (uint32_t i = t; i < n; i += SIFT_THREADS) {
					const uint32_t key = keys[i];
					kbuf[atomicAdd(&bcnt[key >> sh], 1u)] = key;       // (the counter ends up at the bucket's end = the next one's start)
				}
				__syncthreads();
				for (uint32_t p = t; p < n; p += SIFT_THREADS) {
					const uint32_t key = kbuf[p], bk = key >> sh;
					const uint32_t b0 = bk ? bcnt[bk - 1] : 0u, b1 = bcnt[bk];
					uint32_t r = b0;
					for (uint32_t j = b0; j < b1; ++j) r += kbuf[j] < key ? 1u : 0u;
					keys[r] = key;
				}
				__syncthreads();
			} else {                                               // a bucket of hundreds (a repeat cluster inside a wide range): bitonic network
				uint32_t P = 64;
				while (P < n) P <<= 1;
				for (uint32_t i = n + t; i < P; i += SIFT_THREADS) keys[i] = 0xffffffffu;
				__syncthreads();
				for (uint32_t k = 2; k <= P; k <<= 1)
					for (uint32_t j = k >> 1; j > 0; j >>= 1) {
						for (uint32_t i = t; i < (P >> 1); i += SIFT_THREADS) {
							const uint32_t a = ((i & ~(j - 1u)) << 1) | (i & (j - 1u)), b = a | j;
							const uint32_t x = keys[a], y = keys[b];
							if ((x > y) == ((a & k) == 0u)) keys[a] = y, keys[b] = x;
						}
						__syncthreads();
					}
			}
		}
		// ---- which anchors are decided now: all of them in the segment's last range, else all below block hi - 1
		const bool last = hi == hi0;
		uint32_t right_nb = 0;
		if (last && hi0 < n_block) {                            // is there an anchor in block hi0 (it belongs to the next segment)?
			for (int32_t l = t; l < nl; l += SIFT_THREADS) {
				const SeedJobDev s = jobs[j0 + l];
				const int32_t c = cur[c0s + l];
				if (c < s.cnt && kb[s.kb_off + c] == hi0) right_nb = 1;
			}
			right_nb = sift_block_sum(right_nb, scr) ? 1u : 0u;
		}
		uint32_t n_emit = n;
		if (!last) {                                             // first key with local block hi - lo (= block hi - 1)
			const uint32_t bound = (hi - lo) << LB;
			uint32_t a = 0, b = n;
			while (a < b) { const uint32_t mid = (a + b) >> 1; if (keys[mid] < bound) a = mid + 1; else b = mid; }
			n_emit = a;
		}
		if (n - n_emit > SIFT_CARRY_MAX) {                        // (uniform) too many anchors in one block to carry
			if (t == 0) { atomicExch(&qflag[q], 1); seg_kept[blockIdx.x] = 0; seg_base[blockIdx.x] = out0; }
			return;
		}
		const int64_t left0 = prev_blk == 0xffffffffu ? -(int64_t)(1 << 30) : (int64_t)prev_blk - (int64_t)lo + 1;   // local block of the anchor below the buffer
		const int64_t right_end = last && right_nb ? (int64_t)(hi0 - lo) + 1 : ((int64_t)1 << 40);                      // ... of the one above it
		// keep flags of the thread's keys (one per chunk of 256), the chunks' per-wave counts, one scan of those (each wave does
		// it for itself), then the writes: two barriers per range
		constexpr int ECH = SIFT_CAP / SIFT_THREADS;
		static_assert(ECH * 4 <= 64, "the scan of the chunk counts is one wave wide");
		const uint32_t lane = t & 63, wv = t >> 6;
		uint32_t keepbits = 0;
#pragma unroll
		for (int k = 0; k < ECH; ++k) {
			const uint32_t i = (uint32_t)k * SIFT_THREADS + t;
			bool keep = false;
			if (i < n_emit) {
				const int64_t b = (int64_t)(keys[i] >> LB);
				const int64_t lf = i > 0 ? (int64_t)(keys[i - 1] >> LB) : left0;
				const int64_t rt = i + 1 < n ? (int64_t)(keys[i + 1] >> LB) : right_end;
				keep = b - lf <= 1 || rt - b <= 1;
			}
			keepbits |= (keep ? 1u : 0u) << k;
			const unsigned long long bal = __ballot(keep);
			if (lane == 0) wtot[k * 4 + wv] = (uint32_t)__popcll(bal);
		}
		__syncthreads();
		uint32_t inc = lane < (uint32_t)ECH * 4u ? wtot[lane] : 0u;
		const uint32_t own = inc;
#pragma unroll
		for (int o = 1; o < 64; o <<= 1) {
			const uint32_t v = __shfl_up(inc, o);
			if (lane >= (uint32_t)o) inc += v;
		}
		const uint32_t ex = inc - own, n_keep = __shfl(inc, 63);
		if (kept_total + n_keep > stage_cap) {                     // (uniform) more than half of the segment's anchors kept: the host seeds this query
			if (t == 0) { atomicExch(&qflag[q], 1); seg_kept[blockIdx.x] = 0; seg_base[blockIdx.x] = out0; }
			return;
		}
#pragma unroll
		for (int k = 0; k < ECH; ++k) {
			if ((uint32_t)k * SIFT_THREADS >= n_emit) break;
			const bool keep = (keepbits >> k) & 1u;
			const unsigned long long bal = __ballot(keep);
			const uint32_t first = __shfl(ex, k * 4 + (int)wv);
			if (keep) {
				const uint32_t i = (uint32_t)k * SIFT_THREADS + t, key = keys[i], l = key & lmask;
				const uint32_t off = kept_total + first + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
				skey[out0 + off] = ((uint64_t)(uint32_t)q << nb) | (uint64_t)(lo + (key >> LB) - 1u);
				sval[out0 + off] = ((uint64_t)(emitted + i) << 32) | (uint32_t)jobs[j0 + l].qpos;
			}
		}
		kept_total += n_keep;
		// ---- the carry moves to the front (local block 0 of the next range); the range is done
		if (n_emit > 0) prev_blk = lo + (keys[n_emit - 1] >> LB) - 1u;
		uint32_t cv[SIFT_CARRY_MAX / SIFT_THREADS];
#pragma unroll
		for (int k = 0; k < SIFT_CARRY_MAX / SIFT_THREADS; ++k) {
			const uint32_t i = n_emit + t + k * SIFT_THREADS;
			cv[k] = i < n ? keys[i] & lmask : 0u;
		}
		__syncthreads();
#pragma unroll
		for (int k = 0; k < SIFT_CARRY_MAX / SIFT_THREADS; ++k) {
			const uint32_t i = t + k * SIFT_THREADS;
			if (n_emit + i < n) keys[i] = cv[k];
		}
		n_carry = n - n_emit;
		emitted += n_emit;
		lo = hi;
		if (total < SIFT_TARGET / 2) width = width > w_max / 2 ? w_max : width * 2;     // sparse here: wider ranges
		__syncthreads();
	}
	if (t == 0) seg_kept[blockIdx.x] = kept_total, seg_base[blockIdx.x] = out0;
}

// exclusive prefix of the segments' kept counts (a flagged query's segments count as empty), then the first kept anchor of
// every query: qseg[q] = the query's first segment, or, for a query without anchors, that of the next query that has some
// (qseg[n_query] = n_seg).  One workgroup: there are a few thousand segments.
__global__ __launch_bounds__(256) void k_sift_offsets(const SiftSeg *segs, int32_t n_seg, int32_t n_query, const int32_t *qseg, const int32_t *qflag,
                                                      const uint32_t *seg_kept, int64_t *seg_out, int64_t *qfirst2)
{
	__shared__ uint32_t wpre[4];
	const int t = threadIdx.x;
	int64_t running = 0;
	for (int32_t base = 0; base < n_seg; base += 256) {
		const int32_t i = base + t;
		uint32_t v = 0;
		if (i < n_seg && !qflag[segs[i].qid]) v = seg_kept[i];
		uint32_t inc = v;
#pragma unroll
		for (int o = 1; o < 64; o <<= 1) { const uint32_t u = __shfl_up(inc, o); if ((t & 63) >= o) inc += u; }
		__syncthreads();                                        // (wpre of the previous chunk has been read)
		if ((t & 63) == 63) wpre[t >> 6] = inc;
		__syncthreads();
		uint32_t before = 0;
		for (int w = 0; w < (t >> 6); ++w) before += wpre[w];
		if (i < n_seg) seg_out[i] = running + before + inc - v;
		running += (int64_t)wpre[0] + wpre[1] + wpre[2] + wpre[3];
	}
	if (t == 0) seg_out[n_seg] = running;
	__syncthreads();                                            // (one workgroup: its global writes are visible to it behind the barrier)
	for (int32_t q = t; q <= n_query; q += 256) qfirst2[q] = seg_out[qseg[q]];
}
// staging -> dense, segment by segment
__global__ __launch_bounds__(256) void k_sift_copy(const SiftSeg *segs, const int32_t *qflag, const uint32_t *seg_kept, const int64_t *seg_base, const int64_t *seg_out,
                                                   const uint64_t *skey, const uint64_t *sval, uint64_t *key, uint64_t *val)
{
	MPA_SHORT_KERNEL();
	const int32_t sgi = blockIdx.x;
	if (qflag[segs[sgi].qid]) return;
	const uint32_t n = seg_kept[sgi];
	const int64_t from = seg_base[sgi], to = seg_out[sgi];
	for (uint32_t i = threadIdx.x; i < n; i += 256) key[to + i] = skey[from + i], val[to + i] = sval[from + i];
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
#define REFINE_LCAP 384                                     /* hits a workgroup of k_refine_scan_map collects in LDS per chunk before it asks for room in the global list */
#define REFINE_SUPER 4                                      /* chunks of one window a workgroup of k_refine_scan_map sweeps (round 5, one atomic per hit: 8 chunks measured 10.9 ms against 9.0 alone; round 6, hits collected per workgroup: see profiles/r06_experiments.txt) */

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
// Refinement pairing on the device (mp_refine_reg, map.c:53-79): the window positions and the query positions that carry the
// same k-mer, all pairs per k-mer unless there are too many.
//   k_refine_scan_map   the refinement scan again, but the query's DISTINCT k-mers ("groups") sit in an LDS map word -> group, and
//                       every hit also counts itself in wcnt[window][group]: n1 of map.c:66 needs no sort
//   k_refine_pair_count n2 = the group's query positions; the pairs of a hit are n2 if n1 * n2 <= max_ava (32-bit product, as the
//                       reference computes it), else none
//   (exclusive scan)    where each hit's pairs go
//   k_refine_pair_emit  window << 44 | window position << 22 | query position -- one radix sort of these keys is the reference's
//                       sort of every window's pair list (map.c:80), and the window boundaries fall out of per-window counts
//   k_refine_pair_decode  position << 32 | query position: the anchors mp_chain() takes (bbit = 0)
// ------------------------------------------------------------------------------------------------
struct RefineGroups {
	const int64_t *qg_first;     // [n_query + 1] first group of every query
	const uint32_t *gword;       // [n_group] the packed k-mer word of the group
	const uint32_t *gfirst;      // [n_group] its first entry in qpos
	const uint32_t *gcount;      // [n_group] how many query positions carry it
	const uint32_t *qpos;        // query positions (index of the k-mer's last residue), group by group, ascending inside a group
};

__global__ __launch_bounds__(256) void k_refine_scan_map(DevGenome g, const RefineWindowDev *wins, const RefineChunk *chunks, RefineGroups gr, const int64_t *wg_first, RefineTab rt,
                                                         int32_t kmer, int32_t min_aa_len, int32_t hs_log2, uint4 *hits, unsigned long long *n_hits, unsigned long long cap, uint32_t *wcnt,
                                                         const int32_t n_super)
{
	MPA_SHORT_KERNEL();
	extern __shared__ uint32_t lds_refine[];
	const int32_t HS = 1 << hs_log2;
	uint32_t *tkey = lds_refine, *tval = tkey + HS;               // open addressing: word -> group (0xffffffff = empty)
	uint8_t *base = (uint8_t*)(tval + HS);                          // [REFINE_CHUNK + 2 * REFINE_HALO] nt4 codes, 15 = outside the window
	uint8_t *cod = base + REFINE_CHUNK + 2 * REFINE_HALO;           // [same] reduced-alphabet code of the codon ENDING at each position, 0xff = none
	__shared__ uint8_t tab[64];
	// The hits of a chunk are collected in LDS and get their places in the global list with ONE returning atomic per workgroup and
	// chunk that has any (round 5: one per hit -- 3.5 M returning atomics on one address per launch, which IS the 8 ms the kernel
	// took: profiles/r05_pmc_summary.json, 72 % of the wave cycles waiting).  A hit beyond REFINE_LCAP takes its place directly, as
	// before.  The order of the list means nothing: its consumers index it (k_refine_pair_count / _emit) and sort what they emit.
	__shared__ uint4 l_hit[REFINE_LCAP];
	__shared__ uint32_t l_n;
	__shared__ unsigned long long l_base;
	if (threadIdx.x == 0) l_n = 0;
	const RefineChunk ch = chunks[blockIdx.x];
	const RefineWindowDev w = wins[ch.win];
	const int cid = w.vid >> 1, rev = w.vid & 1;
	const int64_t off = g.ctg_off[cid], clen = g.ctg_len[cid];
	if (threadIdx.x < 64) tab[threadIdx.x] = rt.t[threadIdx.x];
	for (int k = threadIdx.x; k < HS; k += 256) tkey[k] = 0xffffffffu;
	__syncthreads();
	// the query's k-mer map, once per workgroup; a workgroup sweeps REFINE_SUPER consecutive chunks of its window
	const int64_t G0 = gr.qg_first[w.qid], G1 = gr.qg_first[w.qid + 1];
	for (int64_t k = G0 + threadIdx.x; k < G1; k += 256) {
		const uint32_t word = gr.gword[k];
		uint32_t slot = (word * 2654435761u) >> (32 - hs_log2);
		for (;;) {
			const uint32_t old = atomicCAS(&tkey[slot], 0xffffffffu, word);
			if (old == 0xffffffffu) { tval[slot] = (uint32_t)(k - G0); break; }     // (the groups' words are distinct)
			slot = (slot + 1) & (HS - 1);
		}
	}
	__syncthreads();
	const uint32_t mask = (1u << (4 * kmer)) - 1;
	auto codon_at = [&](int e) -> uint32_t { return cod[e]; };
	const int64_t wc0 = wg_first[ch.win];
	for (int32_t cstart = ch.start; cstart < w.len && cstart < ch.start + n_super * REFINE_CHUNK; cstart += REFINE_CHUNK) {
	if (cstart != ch.start) __syncthreads();                    // (the scan of the chunk before has finished with base[] and cod[])
	// the chunk's bases and its halo, sixteen per thread from three aligned words of the packed genome (a byte load per base before)
	static_assert((REFINE_CHUNK + 2 * REFINE_HALO) % 16 == 0, "chunk + halo must be a multiple of 16");
	for (int k = threadIdx.x * 16; k < REFINE_CHUNK + 2 * REFINE_HALO; k += 256 * 16) {
		const int64_t p = (int64_t)cstart - REFINE_HALO + k;         // window position of base[k]
		const int64_t x = w.as + p;                                    // strand position
		const uint64_t nib = packed_window16(g.seq, g.l_seq, rev ? off + clen - 1 - x : off + x, rev ? -1 : 1, rev);
		uint32_t o[4];
#pragma unroll
		for (int q = 0; q < 4; ++q) {
			uint32_t v = 0;
#pragma unroll
			for (int j = 0; j < 4; ++j) {
				const int64_t pj = p + 4 * q + j;
				const uint32_t bb = (pj < 0 || pj >= w.len) ? 15u : (uint32_t)(nib >> (4 * (4 * q + j))) & 15u;
				v |= bb << (8 * j);
			}
			o[q] = v;
		}
		uint32_t *dst = (uint32_t*)(base + k);                       // (k is a multiple of 16, base of 4)
		dst[0] = o[0], dst[1] = o[1], dst[2] = o[2], dst[3] = o[3];
	}
	__syncthreads();
	// the codon ending at every position, once (every k-mer and every reading-frame walk below reads these instead of three bases
	// and the table per codon)
	for (int e = threadIdx.x; e < REFINE_CHUNK + 2 * REFINE_HALO; e += 256) {
		uint32_t c = 0xffu;
		if (e >= 2) {
			const uint32_t b0 = base[e - 2], b1 = base[e - 1], b2 = base[e];
			if ((b0 | b1 | b2) <= 3) c = tab[b0 << 4 | b1 << 2 | b2];
		}
		cod[e] = (uint8_t)c;
	}
	__syncthreads();
	// Who scans what: thread 3q + r (q < 85) takes the PER positions 3 PER q + r, + 3, + 6, ... -- one reading frame of a stretch of 3 PER
	// bases -- so that the k-mer word ROLLS: one codon read per position instead of kmer (round 5: thread t took positions t, t + 256,
	// ...: five LDS byte reads per k-mer).  Thread 255 takes the chunk's last PER positions the old way.  `run` = valid codons in a row.
	constexpr int PER = REFINE_CHUNK / 256;
	auto try_hit = [&](const int32_t pos, const int e, const uint32_t word) {
		uint32_t grp = 0xffffffffu;
		for (uint32_t slot = (word * 2654435761u) >> (32 - hs_log2);; slot = (slot + 1) & (HS - 1)) {
			const uint32_t v = tkey[slot];
			if (v == word) { grp = tval[slot]; break; }
			if (v == 0xffffffffu) break;
		}
		if (grp == 0xffffffffu) return;
		int32_t n = kmer;
		for (int q = e - 3 * kmer; n < min_aa_len && q >= 2 && codon_at(q) != 0xffu; q -= 3) ++n;
		for (int q = e + 3; n < min_aa_len && q < REFINE_CHUNK + 2 * REFINE_HALO && codon_at(q) != 0xffu; q += 3) ++n;
		if (n < min_aa_len) return;
		const uint4 hit = make_uint4((uint32_t)ch.win, (uint32_t)pos, (uint32_t)(G0 + grp), grp);
		const uint32_t li = atomicAdd(&l_n, 1u);
		if (li < REFINE_LCAP) l_hit[li] = hit;
		else {
			const unsigned long long at = atomicAdd(n_hits, 1ULL);
			if (at < cap) hits[at] = hit;
		}
		atomicAdd(&wcnt[wc0 + grp], 1u);
	};
	if (threadIdx.x < 255) {
		const int q3 = (int)threadIdx.x / 3, fr = (int)threadIdx.x - 3 * q3;
		const int e0 = REFINE_HALO + 3 * PER * q3 + fr;              // LDS index of the last base of the thread's first k-mer
		uint32_t word = 0;
		int run = 0;
		for (int c = kmer - 1; c >= 1; --c) {                         // the kmer - 1 codons before it
			const uint32_t r = codon_at(e0 - 3 * c);
			if (r == 0xffu) run = 0, word = 0; else word = word << 4 | r, ++run;
		}
#pragma unroll
		for (int j = 0; j < PER; ++j) {
			const int e = e0 + 3 * j;
			const uint32_t r = codon_at(e);
			if (r == 0xffu) { run = 0, word = 0; continue; }
			word = (word << 4 | r) & mask, ++run;
			const int32_t pos = cstart + e - REFINE_HALO;
			if (run >= kmer && pos < w.len) try_hit(pos, e, word);
		}
	} else {
		for (int j = 0; j < PER; ++j) {
			const int e = REFINE_HALO + 3 * PER * 85 + j;
			const int32_t pos = cstart + e - REFINE_HALO;
			if (pos >= w.len) continue;
			uint32_t word = 0;
			bool ok = true;
			for (int c = kmer - 1; c >= 0; --c) {
				const uint32_t r = codon_at(e - 3 * c);
				if (r == 0xffu) { ok = false; break; }
				word = word << 4 | r;
			}
			if (ok) try_hit(pos, e, word & mask);
		}
	}
	__syncthreads();
	// (the buffer is emptied when a chunk leaves it more than half full, and behind the workgroup's last chunk)
	const bool last_chunk = cstart + REFINE_CHUNK >= w.len || cstart + REFINE_CHUNK >= ch.start + n_super * REFINE_CHUNK;
	const uint32_t ln = (l_n >= REFINE_LCAP / 2 || last_chunk) ? (l_n < REFINE_LCAP ? l_n : REFINE_LCAP) : 0u;
	if (ln) {                                                   // (uniform: every thread sees the same count behind the barrier)
		if (threadIdx.x == 0) l_base = atomicAdd(n_hits, (unsigned long long)ln);
		__syncthreads();
		for (uint32_t k = threadIdx.x; k < ln; k += 256) { const unsigned long long at = l_base + k; if (at < cap) hits[at] = l_hit[k]; }
		__syncthreads();
		if (threadIdx.x == 0) l_n = 0;
	}
	}
}

__global__ __launch_bounds__(256) void k_refine_pair_count(const uint4 *hits, int64_t n_hits, const int64_t *wg_first, const uint32_t *wcnt, const uint32_t *gcount, int32_t max_ava,
                                                           uint32_t *pc, uint32_t *wpairs)
{
	MPA_SHORT_KERNEL();
	const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
	if (i >= n_hits) return;
	const uint4 h = hits[i];
	const int32_t n1 = (int32_t)wcnt[wg_first[h.x] + h.w], n2 = (int32_t)gcount[h.z];
	const uint32_t c = (n2 > 0 && (int32_t)((uint32_t)n1 * (uint32_t)n2) <= max_ava) ? (uint32_t)n2 : 0u;   // (the reference's 32-bit product, wrap-around and all)
	pc[i] = c;
	if (c) atomicAdd(&wpairs[h.x], c);
}

__global__ __launch_bounds__(256) void k_refine_pair_emit(const uint4 *hits, int64_t n_hits, const uint32_t *pc, const uint64_t *po, RefineGroups gr, uint64_t *keys)
{
	MPA_SHORT_KERNEL();
	const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
	if (i >= n_hits) return;
	const uint32_t c = pc[i];
	if (!c) return;
	const uint4 h = hits[i];
	const uint64_t hi = (uint64_t)h.x << 44 | (uint64_t)h.y << 22;
	const uint32_t *qp = gr.qpos + gr.gfirst[h.z];
	uint64_t *dst = keys + po[i];
	for (uint32_t j = 0; j < c; ++j) dst[j] = hi | qp[j];
}

__global__ __launch_bounds__(256) void k_refine_pair_decode(const uint64_t *keys, int64_t n, uint64_t *a)
{
	MPA_SHORT_KERNEL();
	const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
	if (i < n) { const uint64_t k = keys[i]; a[i] = ((k >> 22) & 0x3fffffULL) << 32 | (k & 0x3fffffULL); }
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
