// host_regions.cpp -- bookkeeping of candidate regions ("hits") between the GPU stages.
//
// Behaviour follows hit.c of the reference: regions from chains (:32-76), ordering (:97-126),
// primary/secondary assignment (:128-187), secondary selection (:189-236), multi-exon preference
// (:238-250) and the limits on how far a region may be extended (:252-287).  Cheap, branchy, <2 % of
// the reference's CPU time -- it stays on the host, between kernels.
#include <algorithm>
#include <cassert>
#include "host_core.h"

namespace mpa {

// approximate ungapped score of a chain of BINNED anchors (hit.c:6-16)
static int32_t binned_chain_score(const uint64_t *a, int32_t n, int32_t kmer)
{
	int32_t x = kmer;
	for (int32_t i = 1; i < n; ++i) {
		const int32_t dq = (int32_t)a[i] - (int32_t)a[i - 1];
		x += std::min(dq, kmer);
		if (a[i] >> 32 == a[i - 1] >> 32) x += 2;
	}
	return x;
}

int32_t chain_score_ungapped(const uint64_t *a, int32_t n, int32_t kmer)
{
	int32_t x = kmer;
	for (int32_t i = 1; i < n; ++i) {
		const int32_t dq = (int32_t)a[i] - (int32_t)a[i - 1], dr3 = (int32_t)((a[i] >> 32) - (a[i - 1] >> 32));
		const int32_t dr = dr3 / 3;
		if (dq >= dr && dr3 != dr * 3) --x;                 // frameshift between the two anchors
		else x += std::min(std::min(dq, dr), kmer);
	}
	return x;
}

void regions_from_chains(const mpa_idx_s *mi, const std::vector<uint64_t> &u, const std::vector<uint64_t> &a, std::vector<Region> &out)
{
	const int32_t bbit = mi->opt.bbit;
	out.clear();
	out.resize(u.size());
	int32_t k = 0;
	for (size_t c = 0; c < u.size(); ++c) {
		const int32_t n = (int32_t)(uint32_t)u[c];
		Region &r = out[c];
		r.off = k, r.cnt = n;
		int32_t first = k, last = k + n - 1;
		const int32_t v_first = block2vid(mi, (uint32_t)(a[first] >> 32)), v_last = block2vid(mi, (uint32_t)(a[last] >> 32));
		assert(v_first <= v_last);
		if (v_first == v_last) r.vid = (uint32_t)v_first;
		else {                                            // the chain spans two strands/contigs: keep the larger part
			int32_t j, head_end, tail_beg;
			for (j = k; j < k + n; ++j) if ((a[j] >> 32) >= mi->bo[v_first + 1]) break;
			head_end = j;
			for (j = k + n - 1; j >= head_end; --j) if ((a[j] >> 32) < mi->bo[v_last]) break;
			tail_beg = j + 1;
			if (head_end - k > k + n - tail_beg) r.vid = (uint32_t)v_first, last = head_end - 1;
			else r.vid = (uint32_t)v_last, first = tail_beg;
		}
		r.vs = (int64_t)((a[first] >> 32) - mi->bo[r.vid]) << bbit;
		r.ve = (int64_t)((a[last] >> 32) - mi->bo[r.vid] + 1) << bbit;
		r.qs = (int32_t)(uint32_t)a[first];
		r.qe = (int32_t)(uint32_t)a[last];
		const int32_t sc = (int32_t)(u[c] >> 32);
		r.chn_sc = v_first == v_last ? sc : (int32_t)(uint32_t)((double)sc * (last - first + 1) / n + .499);
		r.chn_sc_ungap = binned_chain_score(&a[r.off], r.cnt, mi->opt.kmer);
		k += n;
	}
}

void sort_regions(std::vector<Region> &r)
{
	const int32_t n = (int32_t)r.size();
	if (n <= 1) return;
	std::vector<Pair64> key;
	key.reserve(n);
	for (int32_t i = 0; i < n; ++i) {
		if (r[i].cnt <= 0) continue;                       // soft-deleted
		const int32_t score = r[i].aligned ? r[i].dp_max : r[i].chn_sc;
		key.push_back(Pair64{ (uint64_t)score << 32 | r[i].hash, (uint64_t)i });
	}
	sort_pairs_by_x(key.data(), key.data() + key.size());
	std::vector<Region> sorted;
	sorted.reserve(key.size());
	for (size_t i = key.size(); i-- > 0;) sorted.push_back(std::move(r[key[i].y]));
	r.swap(sorted);
}

void assign_parents(float mask_level, int mask_len, std::vector<Region> &r, int sub_diff)
{
	const int n = (int)r.size();
	if (n <= 0) return;
	for (int i = 0; i < n; ++i) r[i].id = i;
	std::vector<int> prim;                               // indices of primary hits found so far
	std::vector<uint64_t> cov;
	prim.push_back(0);
	r[0].parent = 0;
	for (int i = 1; i < n; ++i) {
		Region &ri = r[i];
		const int si = ri.qs, ei = ri.qe;
		int uncovered = 0;
		bool is_primary = true;
		cov.clear();
		for (int w : prim) {                             // parts of [si,ei) covered by primaries
			int sj = r[w].qs, ej = r[w].qe;
			if (ej <= si || sj >= ei) continue;
			cov.push_back((uint64_t)std::max(sj, si) << 32 | (uint32_t)std::min(ej, ei));
		}
		if (!cov.empty()) {
			int x = si;
			sort_u64(cov.data(), cov.data() + cov.size());
			for (uint64_t c : cov) {
				if ((int)(c >> 32) > x) uncovered += (int)(c >> 32) - x;
				x = std::max((int)(int32_t)c, x);
			}
			if (ei > x) uncovered += ei - x;
			for (int w : prim) {
				Region &rp = r[w];
				const int sj = rp.qs, ej = rp.qe;
				if (ej <= si || sj >= ei) continue;
				const int lo = std::min(ej - sj, ei - si), hi = std::max(ej - sj, ei - si);
				const int ol = si < sj ? (ei < sj ? 0 : ei < ej ? ei - sj : ej - sj) : (ej < si ? 0 : ej < ei ? ej - si : ei - si);
				if ((float)ol / lo - (float)uncovered / hi > mask_level && uncovered <= mask_len) {   // secondary to rp
					int counts = 0, sci = ri.chn_sc;
					ri.parent = rp.parent;
					rp.subsc = std::max(rp.subsc, sci);
					if (ri.cnt >= rp.cnt) counts = 1;
					if (rp.aligned && ri.aligned && (rp.vid != ri.vid || rp.vs != ri.vs || rp.ve != ri.ve || ol != lo)) {
						sci = ri.dp_max;
						rp.dp_max2 = std::max(rp.dp_max2, sci);
						if (rp.dp_max - ri.dp_max <= sub_diff) counts = 1;
					}
					if (counts) ++rp.n_sub;
					is_primary = false;
					break;
				}
			}
		}
		if (is_primary) prim.push_back(i), ri.parent = i, ri.n_sub = 0;
	}
}

// after dropping hits: renumber ids and remap parents (hit.c:189-210)
static void renumber(std::vector<Region> &r)
{
	int max_id = -1;
	for (const Region &x : r) max_id = std::max(max_id, x.id);
	std::vector<int> where(max_id + 1, -1);
	for (size_t i = 0; i < r.size(); ++i) if (r[i].id >= 0) where[r[i].id] = (int)i;
	for (size_t i = 0; i < r.size(); ++i) {
		Region &x = r[i];
		x.id = (int)i;
		if (x.parent == -2) x.parent = (int)i;
		else if (x.parent >= 0 && x.parent <= max_id && where[x.parent] >= 0) x.parent = where[x.parent];
		else x.parent = -1;
	}
}

void select_secondary(float pri_ratio, int min_diff, int best_n, std::vector<Region> &r)
{
	if (!(pri_ratio > 0.0f) || r.empty()) return;
	const int n = (int)r.size();
	int top_ungap = -1, n_2nd = 0;
	for (const Region &x : r) top_ungap = std::max(top_ungap, x.chn_sc_ungap);
	auto same_place = [](const Region &x, const Region &y) {
		return x.qs == y.qs && x.qe == y.qe && x.vid == y.vid && x.vs == y.vs && x.ve == y.ve;
	};
	// NB: like the reference this compacts in place, so r[p] may already hold a moved element; parents always
	// precede their children and kept elements only move towards the front, which keeps r[p] the parent.
	std::vector<Region> kept;
	kept.reserve(n);
	std::vector<int> new_pos(n, -1);
	for (int i = 0; i < n; ++i) {
		const int p = r[i].parent;
		// the reference reads r[p] from the array being compacted: position p holds kept element #p' where p' <= p
		const Region &rp = (p >= 0 && p < (int)kept.size()) ? kept[p] : r[p];
		const int sci = r[i].aligned ? r[i].dp_max : r[i].chn_sc;
		const int scp = rp.aligned ? rp.dp_max : rp.chn_sc;
		bool keep = false;
		if (p == i) keep = true;
		else if ((sci >= scp * pri_ratio || sci + min_diff >= scp) && n_2nd < best_n) {
			if (!same_place(r[i], rp)) keep = true, ++n_2nd;
		} else if (!r[i].aligned && !rp.aligned && top_ungap > 0 && r[i].chn_sc_ungap >= top_ungap * pri_ratio && n_2nd < best_n) {
			if (!same_place(r[i], rp)) keep = true, ++n_2nd;
		}
		if (keep) kept.push_back(r[i]);
	}
	const bool dropped = (int)kept.size() != n;
	r.swap(kept);
	if (dropped) renumber(r);
}

void prefer_multi_exon(std::vector<Region> &r, int32_t single_penalty)
{
	const int n = (int)r.size();
	if (n < 2 || r[0].n_exon != 1) return;
	int i = 1;
	while (i < n && r[i].n_exon < 2) ++i;
	if (i == n || !r[0].aligned || !r[i].aligned) return;
	if (r[0].dp_max < r[i].dp_max + single_penalty) std::swap(r[0], r[i]);
}

void extension_limits(const mpa_idx_s *mi, const std::vector<Region> &r, const std::vector<uint64_t> *anchors, int32_t min_ext, int32_t max_ext, std::vector<uint64_t> &ext)
{
	const int32_t n = (int32_t)r.size();
	ext.assign(n, 0);
	if (n <= 0) return;
	std::vector<Pair64> pos(n);
	for (int32_t i = 0; i < n; ++i) {
		const Region &x = r[i];
		if (anchors) pos[i].x = (*anchors)[x.off] >> 32;
		else pos[i].x = (uint64_t)(x.vs + mi->ctg[x.vid >> 1].off + ((x.vid & 1) ? mi->ctg[x.vid >> 1].len : 0));
		pos[i].y = (uint64_t)i;
	}
	sort_pairs_by_x(pos.data(), pos.data() + n);
	for (int32_t i = 0; i < n; ++i) {
		const int32_t j = (int32_t)pos[i].y;
		const Region &x = r[j];
		int32_t left = max_ext, right = max_ext;
		if (i > 0) {
			const Region &q = r[pos[i - 1].y];
			if (q.vid == x.vid && q.qe >= x.qs) {
				left = (int32_t)(x.vs - q.ve < max_ext ? x.vs - q.ve : max_ext);
				left = std::max(left, min_ext);
			}
		}
		if (i < n - 1) {
			const Region &q = r[pos[i + 1].y];
			if (q.vid == x.vid && x.qe >= q.qs) {
				right = (int32_t)(q.vs - x.ve < max_ext ? q.vs - x.ve : max_ext);
				right = std::max(right, min_ext);
			}
		}
		ext[j] = (uint64_t)(uint32_t)left << 32 | (uint32_t)right;
	}
}

} // namespace mpa
