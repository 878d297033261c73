/* chain_oracle.c -- restatement of miniprot's anchor chaining (chain.c).
 *
 * TEST INFRASTRUCTURE ONLY (see mporacle.h).  Parity PINNED against oracle/_ref.
 *
 * Follows chain.c:112-151 (pair score), :160-219 (DP fill with max_skip / max_iter / rescue
 * heuristics), :8-75 (greedy backtracking with best-suffix cut) and :77-110 (compaction and
 * re-ordering of chains by target position).  Floating point: compiled with -ffp-contract=off,
 * the reference object contains no FMA (SURVEY.md section 7 "hard parts").
 */
#include <stdlib.h>
#include <string.h>
#include "mporacle.h"

static float log2_poly(float x)                                /* mp_log2, mppriv.h:91-99 */
{
	union { float f; uint32_t i; } z = { x };
	float r = (float)((int32_t)((z.i >> 23) & 255) - 128);
	z.i &= ~(255u << 23);
	z.i += 127u << 23;
	r += (-0.34484843f * z.f + 2.02466578f) * z.f - 0.67487759f;
	return r;
}

int32_t mpo_comput_sc(uint64_t ai, uint64_t aj, int32_t max_dist_x, int32_t max_dist_y, int32_t bw,
                      float coef_log, int32_t is_spliced, int32_t bbit, int32_t kmer)
{
	const int32_t dq = (int32_t)ai - (int32_t)aj, dq3 = dq * 3;
	int32_t dr3, gap, gap_signed, sc;
	if (dq <= 0 || dq3 > max_dist_x || dq > max_dist_y) return INT32_MIN;
	if (bbit > 0) {                      /* binned anchors: smallest gap compatible with the two blocks */
		const int32_t bs = 1 << bbit;
		dr3 = (int32_t)(((ai >> 32) - (aj >> 32)) << bbit);
		if (dq3 < dr3 - bs) gap = dr3 - bs - dq3, gap_signed = -gap;
		else if (dq3 > dr3 + bs) gap = dq3 - (dr3 + bs), gap_signed = gap;
		else gap = gap_signed = 0;
	} else {
		dr3 = (int32_t)((ai >> 32) - (aj >> 32));
		if (dr3 == 0) return INT32_MIN;
		gap = dr3 > dq3 ? dr3 - dq3 : dq3 - dr3;
		gap_signed = dq3 - dr3;
	}
	if (gap > bw) return INT32_MIN;
	if (bbit > 0) sc = kmer < dq ? kmer : dq;
	else if (kmer <= dq && kmer * 3 <= dr3) sc = kmer;
	else {
		const int32_t dr = dr3 / 3, rem = dr3 - dr * 3, m = dr < dq ? dr : dq;
		sc = m < kmer ? m : kmer;
		if (rem != 0) --sc;
	}
	if (gap > 0) {
		const float lin = (float)gap * .33334f;
		const float lg = gap >= 2 ? coef_log * (log2_poly((float)(gap + 1)) - 1.0f) + 1.0f : (float)gap;
		if (is_spliced && gap_signed < 0) sc -= (int)(lin < lg ? lin : lg);
		else sc -= (int)(lin + lg);
	}
	if (bbit > 0 && ai >> 32 == aj >> 32 && gap == 0) sc += 2;   /* MP_BLOCK_BONUS */
	return sc;
}

/* chain.c:8-24: walk back from chain end z[k], find where the best-scoring suffix starts */
static int64_t best_suffix_start(int32_t max_drop, const mpo128_t *z, const int32_t *f, const int64_t *p, int32_t *t, int64_t k)
{
	int64_t i = (int64_t)z[k].y, stop = -1, best_i = i;
	int32_t best = 0;
	if (i < 0 || t[i] != 0) return i;
	do {
		int32_t s;
		t[i] = 2;
		stop = i = p[i];
		s = i < 0 ? (int32_t)z[k].x : (int32_t)z[k].x - f[i];
		if (s > best) best = s, best_i = i;
		else if (best - s > max_drop) break;
	} while (i >= 0 && t[i] == 0);
	for (i = (int64_t)z[k].y; i >= 0 && i != stop; i = p[i]) t[i] = 0;
	return best_i;
}

int32_t mpo_chain(int32_t max_dist_x, int32_t max_dist_y, int32_t bw, int32_t max_skip, int32_t max_iter,
                  int32_t min_cnt, int32_t min_sc, float coef_log, int32_t is_spliced, int32_t kmer, int32_t bbit,
                  int64_t n, uint64_t *a, uint64_t *u, int64_t *n_out)
{
	int32_t *f, *t, *v, max_drop = bw, resc_f = 0, n_u = 0;
	int64_t *p, i, j, st = 0, resc_i = -1, n_z = 0, n_v = 0, k;
	mpo128_t *z, *w;
	uint64_t *b, *u2;

	*n_out = 0;
	if (n == 0) return 0;
	if (max_dist_x < bw) max_dist_x = bw;
	if (max_dist_y < bw && !is_spliced) max_dist_y = bw;
	if (is_spliced) max_drop = INT32_MAX;
	p = (int64_t*)malloc(n * 8); f = (int32_t*)malloc(n * 4); v = (int32_t*)malloc(n * 4); t = (int32_t*)calloc(n, 4);

	/* fill, chain.c:181-209 */
	for (i = 0; i < n; ++i) {
		int64_t arg = -1;
		int32_t best = kmer, n_skip = 0;
		while (st < i && (int64_t)(((a[i] >> 32) - (a[st] >> 32)) << bbit) > max_dist_x) ++st;
		if (resc_i >= 0 && resc_i >= st) {
			int32_t sc = resc_f + mpo_comput_sc(a[i], a[resc_i], max_dist_x, max_dist_y, bw, coef_log, is_spliced, bbit, kmer);
			if (sc > best) best = sc, arg = resc_i;
		} else resc_f = 0, resc_i = -1;
		if (i - st > max_iter) st = i - max_iter;
		for (j = i - 1; j >= st; --j) {
			int32_t sc = mpo_comput_sc(a[i], a[j], max_dist_x, max_dist_y, bw, coef_log, is_spliced, bbit, kmer);
			if (sc == INT32_MIN) continue;
			sc += f[j];
			if (sc > best) {
				best = sc, arg = j;
				if (n_skip > 0) --n_skip;
			} else if (t[j] == (int32_t)i) {
				if (++n_skip > max_skip) break;
			}
			if (p[j] >= 0) t[p[j]] = (int32_t)i;
		}
		f[i] = best, p[i] = arg;
		v[i] = arg >= 0 && v[arg] > best ? v[arg] : best;
		if (resc_f < best) resc_f = best, resc_i = i;
	}

	/* backtrack, chain.c:26-75 (the reference makes a counting pass then a filling pass; one pass suffices) */
	for (i = 0; i < n; ++i) if (f[i] >= min_sc) ++n_z;
	if (n_z == 0) { free(p); free(f); free(v); free(t); return 0; }
	z = (mpo128_t*)malloc(n_z * sizeof(mpo128_t));
	for (i = 0, k = 0; i < n; ++i) if (f[i] >= min_sc) z[k].x = (uint64_t)(int64_t)f[i], z[k++].y = (uint64_t)i;
	mpo_radix_sort128x(z, z + n_z);
	memset(t, 0, n * 4);
	for (k = n_z - 1; k >= 0; --k) {
		int64_t n_v0 = n_v, end_i;
		int32_t sc;
		if (t[z[k].y] != 0) continue;
		end_i = best_suffix_start(max_drop, z, f, p, t, k);
		for (i = (int64_t)z[k].y; i != end_i; i = p[i]) v[n_v++] = (int32_t)i, t[i] = 1;
		sc = i < 0 ? (int32_t)z[k].x : (int32_t)z[k].x - f[i];
		if (sc >= min_sc && n_v > n_v0 && n_v - n_v0 >= min_cnt) u[n_u++] = (uint64_t)sc << 32 | (uint64_t)(n_v - n_v0);
		else n_v = n_v0;
	}
	free(z); free(p); free(f); free(t);
	if (n_u == 0) { free(v); return 0; }

	/* compaction, chain.c:77-110: anchors of each chain in ascending order, chains sorted by first target pos */
	b = (uint64_t*)malloc(n_v * 8);
	for (i = 0, k = 0; i < n_u; ++i) {
		int64_t k0 = k, ni = (int32_t)u[i];
		for (j = 0; j < ni; ++j) b[k++] = a[v[k0 + (ni - j - 1)]];
	}
	free(v);
	w = (mpo128_t*)malloc(n_u * sizeof(mpo128_t));
	for (i = k = 0; i < n_u; ++i) {
		w[i].x = b[k] >> 32, w[i].y = (uint64_t)k << 32 | (uint64_t)i;
		k += (int32_t)u[i];
	}
	mpo_radix_sort128x(w, w + n_u);
	u2 = (uint64_t*)malloc(n_u * 8);
	for (i = k = 0; i < n_u; ++i) {
		int32_t src = (int32_t)w[i].y, cnt = (int32_t)u[src];
		u2[i] = u[src];
		memcpy(&a[k], &b[w[i].y >> 32], cnt * 8);
		k += cnt;
	}
	memcpy(u, u2, n_u * 8);
	free(b); free(w); free(u2);
	*n_out = k;
	return n_u;
}
