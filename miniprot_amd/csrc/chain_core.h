// chain_core.h -- chain extraction (mp_chain_backtrack + compact_a, chain.c:8-110) and the reference's pair sort
// (radix_sort_mp128x, ksort.h:109-162) as ONE source for the host and the device.
//
// The extraction consumes the order in which the reference's unstable in-place MSD radix sort leaves chain ends of equal
// score, so it cannot be re-derived independently for the GPU: the functions below are plain C++ on raw arrays (no
// allocation, no library calls, no recursion) and are compiled twice -- by the host pipeline (host_core.cpp, whose results
// tests/test_host_core.py fuzzes against the oracle) and, unchanged, into k_chain_extract (seed_exec.hip), where one
// wavefront runs them for one chaining problem.  Whatever the host does, the device does.
//
// Execution model ("Coop" policy C).  The code is written for a team of C::width() lanes that all run it with identical
// scalar state: the sequential parts (the sort replay, the greedy extraction) are executed by every lane redundantly -- same
// addresses, same values, one memory transaction per wave -- and the loops that merely fill, count, scatter or compact are
// strided over the lanes (MPA_COOP_FOR, coop_compact) and followed by C::sync().  On the host the team is one lane
// (CoopSerial): the strided loops are plain loops and the code is the sequential algorithm.
#pragma once
#include <cstdint>
#include <cstddef>
#ifdef __HIPCC__
#define MPA_HD __host__ __device__
#else
#define MPA_HD
#endif

namespace mpa {

struct Pair64 { uint64_t x, y; };     // mp128_t (miniprot.h:32)

struct ChainParams {
	int32_t max_dist_x, max_dist_y, bw, max_skip, max_iter, min_cnt, min_sc;
	float coef_log;
	int32_t is_spliced, kmer, bbit;
};

// The state after the forward pass (chain.c:181-209) of a chaining problem with n_total anchors.  A dense view holds all of
// them (pos == nullptr, m == n_total).  A sparse view (only valid when min_cnt > 1) may leave out anchors that have no
// predecessor and are nobody's predecessor; pos[] are then the ascending positions of the m anchors it does hold.
struct ChainView {
	int64_t n_total, m;
	const int32_t *pos;      // [m] or nullptr
	const int32_t *f;        // [m] chain score ending at the anchor
	const int32_t *pred;     // [m] view index of the predecessor, -1 for none
	const uint64_t *a;       // [m] the anchors
};

struct SparseItem { int64_t pos; Pair64 e; };        // an element that is not a root, and the slot it occupies
struct SortRange { int64_t beg, end; int32_t shift, pad; };

// Work space of chain_extract_core() for a view of m anchors (every array [m] unless noted).
struct ExtractScratch {
	int32_t *mark, *order;
	Pair64 *ends;            // [ends_cap]: m, or n_total when the full list of chain ends has to be built (see chain_ends_full_needed)
	int64_t ends_cap;
	SparseItem *items, *moved, *merged;
	Pair64 *tail8, *first;
	uint8_t *kept;
	uint64_t *packed, *u_sorted;
	SortRange *stack;        // [ends_cap / 64 + 4]
	uint32_t *hist;          // [1280] bucket counts / cursors / limits of the digit passes, + two 256-entry tables of the two-level replay (shared by the team: LDS on the device)
};

#define MPA_ABSENT (~0ULL)   /* an anchor the view leaves out (alone in its neighbourhood) */

struct CoopSerial {          // a team of one: the host
	static MPA_HD int lane() { return 0; }
	static MPA_HD int width() { return 1; }
	static MPA_HD void sync() {}
	static MPA_HD uint64_t ballot(bool p) { return p ? 1u : 0u; }
	static MPA_HD int rank(uint64_t) { return 0; }                     // lanes below this one whose ballot bit is set
	static MPA_HD int popc(uint64_t m) { return (int)m; }
	static MPA_HD int32_t reduce_max(int32_t v) { return v; }
	static MPA_HD bool any(bool p) { return p; }
	static MPA_HD void count(uint32_t *slot) { ++*slot; }              // histogram increment (shared counters)
	static MPA_HD void atomic_min(int32_t *slot, int32_t v) { if (v < *slot) *slot = v; }
	static MPA_HD int64_t scan_excl(int64_t v, int64_t *total) { *total = v; return 0; }   // sum of v over the lanes below this one; the team's total
	static MPA_HD int64_t scan_max_excl(int64_t v, int64_t *total) { *total = v; return INT64_MIN; }   // maximum over the lanes below (INT64_MIN: none); the team's
	static MPA_HD void mark_time(int) {}                               // phase stamps of the device profile (MPA_TIMING=2)
	static MPA_HD void note(int, int64_t) {}                           // ... and problem sizes
	// among the lanes with `have`, those whose digit d equals this lane's: how many sit below this lane, and how many there are
	static MPA_HD void digit_rank(int, bool, int *rank, int *cnt) { *rank = 0, *cnt = 1; }
	static MPA_HD int first_unset(uint64_t m) { return m & 1u ? 1 : 0; }
	static MPA_HD int lowest(uint64_t) { return 0; }                    // index of the lowest lane whose ballot bit is set (m != 0)
	static MPA_HD uint64_t *scratch(int64_t *cap) { *cap = 0; return nullptr; }   // fast memory shared by the team (LDS on the device), in 8-byte words   // index of the lowest lane whose ballot bit is clear (width() if none)
};
#define MPA_COOP_FOR(C, i, n) for (int64_t i = C::lane(); i < (int64_t)(n); i += C::width())

// order-preserving compaction of the i in [0, n) with pred(i): emit(i, k) for the k-th of them; returns how many
template<class C, class Pred, class Emit> MPA_HD inline int64_t coop_compact(int64_t n, Pred pred, Emit emit)
{
	int64_t total = 0;
	for (int64_t base = 0; base < n; base += C::width()) {
		const int64_t i = base + C::lane();
		const bool p = i < n && pred(i);
		const uint64_t mask = C::ballot(p);
		if (p) emit(i, total + C::rank(mask));
		total += C::popc(mask);
	}
	C::sync();
	return total;
}

// ---- radix_sort_mp128x (ksort.h:109-162): the order of EQUAL keys is part of the behaviour ------------------------------
// Runs of <= 64 elements: insertion sort (stable).  Longer runs: in-place cycle-leader permutation by the byte at `shift`,
// then the same for every bucket at the next byte.
MPA_HD inline void pairs_insertion(Pair64 *beg, Pair64 *end)
{
	for (Pair64 *i = beg + 1; i < end; ++i) {
		if (i->x >= (i - 1)->x) continue;
		Pair64 v = *i, *j = i;
		for (; j > beg && v.x < (j - 1)->x; --j) *j = *(j - 1);
		*j = v;
	}
}

// one digit level over [beg, end); hist[0..255] receives the bucket sizes (hist[256..767]: cursors and limits)
MPA_HD inline void pairs_digit_level(Pair64 *beg, Pair64 *end, int shift, uint32_t *hist)
{
	uint32_t *cnt = hist, *cursor = hist + 256, *limit = hist + 512;
	for (int d = 0; d < 256; ++d) cnt[d] = 0;
	for (Pair64 *p = beg; p != end; ++p) ++cnt[p->x >> shift & 0xff];
	bool single = false;
	{
		uint32_t at = 0;
		for (int d = 0; d < 256; ++d) { cursor[d] = at, at += cnt[d], limit[d] = at; if (cnt[d] == (uint32_t)(end - beg)) single = true; }
	}
	for (int k = 0; !single && k < 256;) {
		if (cursor[k] == limit[k]) { ++k; continue; }
		int d = (int)(beg[cursor[k]].x >> shift & 0xff);
		if (d == k) { ++cursor[k]; continue; }
		Pair64 carried = beg[cursor[k]];            // follow the displacement cycle until an element of bucket k turns up
		do {
			const Pair64 placed = carried;
			carried = beg[cursor[d]];
			beg[cursor[d]++] = placed;
			d = (int)(carried.x >> shift & 0xff);
		} while (d != k);
		beg[cursor[k]++] = carried;
	}
}

// The same digit level (shift 0) on elements packed into 8 bytes -- key << 32 | payload (0xffffffff: MPA_ABSENT) -- as they sit in
// the team's fast scratch memory: same counts, same walk, same result as pairs_digit_level(.., 0, ..) on the unpacked elements.
MPA_HD inline void packed_digit_level0(uint64_t *beg, uint32_t n, uint32_t *hist)
{
	uint32_t *cnt = hist, *cursor = hist + 256, *limit = hist + 512;
	for (int d = 0; d < 256; ++d) cnt[d] = 0;
	for (uint32_t i = 0; i < n; ++i) ++cnt[beg[i] >> 32 & 0xff];
	bool single = false;
	{
		uint32_t at = 0;
		for (int d = 0; d < 256; ++d) { cursor[d] = at, at += cnt[d], limit[d] = at; if (cnt[d] == n) single = true; }
	}
	for (int k = 0; !single && k < 256;) {
		if (cursor[k] == limit[k]) { ++k; continue; }
		int d = (int)(beg[cursor[k]] >> 32 & 0xff);
		if (d == k) { ++cursor[k]; continue; }
		uint64_t carried = beg[cursor[k]];
		do {
			const uint64_t placed = carried;
			carried = beg[cursor[d]];
			beg[cursor[d]++] = placed;
			d = (int)(carried >> 32 & 0xff);
		} while (d != k);
		beg[cursor[k]++] = carried;
	}
}

// pairs_digit_level(beg, end, 0, hist) by the whole team: a bucket that fits the team's scratch memory and whose keys fit 32 bits
// (the caller's guarantee) is copied there by all lanes, walked there (every step of the walk is a dependent memory access: tens
// of cycles in LDS against hundreds to HBM) and copied back; anything else is walked in place.  A team of one has no scratch.
template<class C> MPA_HD inline void team_digit_level0(Pair64 *beg, Pair64 *end, uint32_t *hist)
{
	int64_t cap = 0;
	uint64_t *fast = C::scratch(&cap);
	const int64_t n = end - beg;
	if (!fast || n > cap) { pairs_digit_level(beg, end, 0, hist); C::sync(); return; }
	MPA_COOP_FOR(C, i, n) fast[i] = beg[i].x << 32 | (beg[i].y == MPA_ABSENT ? 0xffffffffull : (beg[i].y & 0xffffffffull));
	C::sync();
	packed_digit_level0(fast, (uint32_t)n, hist);
	C::sync();
	MPA_COOP_FOR(C, i, n) { const uint64_t p = fast[i]; beg[i] = Pair64{ p >> 32, (p & 0xffffffffull) == 0xffffffffull ? MPA_ABSENT : (p & 0xffffffffull) }; }
	C::sync();
}

// every level below `shift` too (the reference recurses; the buckets are independent, so a work list gives the same result)
MPA_HD inline void pairs_digit_pass(Pair64 *base, int64_t beg, int64_t end, int shift, SortRange *stack, uint32_t *hist)
{
	int64_t top = 0;
	stack[top++] = SortRange{ beg, end, shift, 0 };
	while (top > 0) {
		const SortRange r = stack[--top];
		const uint32_t *cnt = hist;
		pairs_digit_level(base + r.beg, base + r.end, r.shift, hist);
		if (r.shift == 0) continue;
		const int next = r.shift > 8 ? r.shift - 8 : 0;
		int64_t p = r.beg;
		for (int d = 0; d < 256; ++d) {
			const int64_t e = p + cnt[d];
			if (cnt[d] > 64) stack[top++] = SortRange{ p, e, next, 0 };     // (at most (end - beg) / 65 ranges are ever pending)
			else if (cnt[d] > 1) pairs_insertion(base + p, base + e);
			p = e;
		}
	}
}

// stack: [(end - beg) / 64 + 4]
MPA_HD inline void sort_pairs_by_x_core(Pair64 *beg, Pair64 *end, SortRange *stack, uint32_t *hist)
{
	if (end - beg <= 64) { pairs_insertion(beg, end); return; }
	// A digit level at which every key has the same byte leaves the array untouched (each element is already in "its"
	// bucket) and just recurses into that single bucket, so the leading constant bytes can be skipped without changing the
	// result -- keys here are chain scores or block ids, i.e. mostly zero in the top bytes.
	uint64_t all_or = 0, all_and = ~0ULL;
	for (const Pair64 *p = beg; p != end; ++p) all_or |= p->x, all_and &= p->x;
	const uint64_t varying = all_or ^ all_and;
	int shift = 56;
	while (shift > 0 && ((varying >> shift) & 0xff) == 0) shift -= 8;
	pairs_digit_pass(beg, 0, end - beg, shift, stack, hist);
}

// ---- the chain ends in the order the reference's sort leaves them, from a SPARSE view --------------------------------------
// For the common case that almost every anchor is a chain of one.  With min_cnt > 1 such "root" anchors (no predecessor,
// score == kmer, the smallest key) never yield a chain, so only the order of the OTHER ends matters; but that order is a
// product of the in-place MSD radix sort run over the whole array, roots included.  With all scores below 256 that sort is a
// single digit pass, and its effect on the non-roots can be replayed without the roots: bucket `kmer` is the first region
// [0, n_root); walking it only ever moves the non-roots found there (in ascending position) into the heads of their
// buckets, displacing what sits there; everything after that happens inside the tail [n_root, n), which is small.
//
// The walk of the reference's cycle-leader pass over the FIRST bucket's region (ksort.h:132-142), seen from the elements that
// do not belong there: src[0, n_src), in order.  Each is dropped on the head of its own bucket in `tail` (tail[cursor[d]++]),
// and whatever it displaces is carried on the same way until an element of the first bucket (digit == stop) is displaced --
// that one ends up in the hole the walk started from, which sink(i, element) is told about.
// The team takes C::width() elements at a time and places them in rounds: the elements in front of the first one whose
// displacement carries on (it hit something that is not of the first bucket) do not interact -- elements of one digit take
// consecutive slots in their order, digits have disjoint slots -- and are placed side by side; that one is replayed alone; the next
// round starts behind it.  With a team of one this is the sequential walk.
// sink_par(i, element, have): called by every lane, have = ITS element was placed side by side in this round; sink_uni(i, element):
// called by all lanes alike for an element that was replayed alone.
template<class C, int SHIFT, class SinkPar, class SinkUni>
MPA_HD inline void place_in_order(const SparseItem *src, int64_t n_src, Pair64 *tail, uint32_t *cursor, int stop, SinkPar sink_par, SinkUni sink_uni)
{
	for (int64_t base = 0; base < n_src; base += C::width()) {
		const int64_t i = base + C::lane();
		const bool have = i < n_src;
		const Pair64 e = have ? src[i].e : Pair64{ 0, 0 };
		const int d = (int)(e.x >> SHIFT & 0xff);
		const int n_grp = (int)(n_src - base < C::width() ? n_src - base : C::width());
		// The elements of the group are placed in their order.  Each round: the elements not yet placed look at the slot they
		// would get if everyone before them just dropped in (elements of one digit take consecutive slots in their order, digits
		// have disjoint slots); those BEFORE the first one whose displacement carries on -- it hit a non-root: about one slot in
		// eleven of a pre-chain's tail -- do not interact and are placed side by side; that one is replayed alone, as the reference
		// walks it; the round after starts behind it.  (Round 3 replayed the whole group one element at a time whenever any of
		// its 64 displacements carried on, i.e. nearly always.)
		int done = 0;                                            // elements [0, done) of the group are placed
		while (done < n_grp) {
			const bool active = have && C::lane() >= done;
			int rank = 0, same = 1;
			C::digit_rank(d, active, &rank, &same);
			const uint32_t slot = active ? cursor[d] + (uint32_t)rank : 0u;
			const Pair64 old = active ? tail[slot] : Pair64{ 0, 0 };
			const bool carries_on = active && (int)(old.x >> SHIFT & 0xff) != stop;
			const uint64_t bad = C::ballot(carries_on);
			const int first_bad = bad ? C::lowest(bad) : n_grp;  // (a team of one: 0 if its element carries on)
			const bool commit = active && C::lane() < first_bad;
			int crank = 0, csame = 1;
			C::digit_rank(d, commit, &crank, &csame);
			if (commit) tail[slot] = e;
			sink_par(i, old, commit);
			C::sync();
			if (commit && crank == csame - 1) cursor[d] += (uint32_t)csame;
			C::sync();
			done = first_bad;
			if (first_bad < n_grp) {                             // (every lane replays it: identical state, as in the sequential parts)
				const int64_t g = base + first_bad;
				Pair64 carried = src[g].e;
				int dd = (int)(carried.x >> SHIFT & 0xff);
				do {
					const Pair64 placed = carried;
					carried = tail[cursor[dd]];
					tail[cursor[dd]++] = placed;
					dd = (int)(carried.x >> SHIFT & 0xff);
				} while (dd != stop);
				sink_uni(g, carried);
				C::sync();
				done = first_bad + 1;
			}
		}
	}
}

// The cycle-leader walk of the reference's digit pass (ksort.h:132-142) over the buckets [k_first, 256) of `tail`, whose bucket
// boundaries are cursor[] / limit[].  Nearly every slot already holds an element of its own bucket (the heads were filled by
// place_in_order, the rest are roots), and the reference's loop merely steps over those one at a time: the team tests
// C::width() consecutive slots at once and only the displacement cycles -- which move data -- are replayed one element at a
// time, by every lane alike.  With a team of one this is the reference's loop.
template<class C, int SHIFT> MPA_HD inline void walk_buckets(Pair64 *tail, uint32_t *cursor, const uint32_t *limit, int k_first)
{
	for (int k = k_first; k < 256;) {
		uint32_t c = cursor[k];
		const uint32_t lim = limit[k];
		while (c < lim) {                                        // step over the slots that are already in place
			const uint32_t i = c + (uint32_t)C::lane();
			const bool in_place = i < lim && (int)(tail[i < lim ? i : c].x >> SHIFT & 0xff) == k;
			const int n = C::first_unset(C::ballot(in_place));
			c += (uint32_t)n;
			if (n < C::width()) break;
		}
		C::sync();
		cursor[k] = c;
		C::sync();
		if (c == lim) { ++k; continue; }
		Pair64 carried = tail[c];                                // a displacement cycle: until an element of bucket k turns up
		int d = (int)(carried.x >> SHIFT & 0xff);
		do {
			const Pair64 placed = carried;
			carried = tail[cursor[d]];
			tail[cursor[d]++] = placed;
			d = (int)(carried.x >> SHIFT & 0xff);
		} while (d != k);
		tail[cursor[k]++] = carried;
		C::sync();
	}
}

// The last digit pass (shift 0) of the reference's sort over a region of n slots that holds roots (key k0, the smallest key
// present) everywhere except at `items` (m of them, ascending pos).  Writes the m non-roots in their final order to out[0, m).
template<class C> MPA_HD inline void replay_low_digit(int64_t n, int k0, const SparseItem *items, int64_t m, Pair64 *out, uint32_t *hist)
{
	if (m == 0) return;
	uint32_t *cnt = hist, *cursor = hist + 256, *limit = hist + 512;
	MPA_COOP_FOR(C, d, 256) cnt[d] = 0;
	C::sync();
	MPA_COOP_FOR(C, i, m) C::count(&cnt[items[i].e.x & 0xff]);
	const int64_t n_root = n - m;
	Pair64 *tail = out;
	MPA_COOP_FOR(C, i, m) tail[i] = Pair64{ (uint64_t)k0, MPA_ABSENT };
	C::sync();
	// the slots [n_root, n) as they are before the pass (items are ascending in pos: the first one at or behind n_root)
	int64_t first_tail = 0;
	{
		int64_t lo = 0, hi = m;
		while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (items[mid].pos < n_root) lo = mid + 1; else hi = mid; }
		first_tail = lo;
	}
	MPA_COOP_FOR(C, i, m - first_tail) tail[items[first_tail + i].pos - n_root] = items[first_tail + i].e;
	C::sync();
	{
		uint32_t at = 0;
		for (int d = 0; d < 256; ++d) { const uint32_t c = cnt[d]; cursor[d] = at, at += c, limit[d] = at; }
	}
	// walk of bucket k0 over [0, n_root): roots stay, every other element starts a displacement cycle
	C::sync();
	place_in_order<C, 0>(items, first_tail, tail, cursor, k0, [](int64_t, const Pair64 &, bool) {}, [](int64_t, const Pair64 &) {});
	// the remaining buckets, entirely inside the tail (same walk as pairs_digit_level)
	walk_buckets<C, 0>(tail, cursor, limit, k0 + 1);
	C::sync();
}

MPA_HD inline int64_t view_pos(const ChainView &v, int64_t i) { return v.pos ? (int64_t)v.pos[i] : i; }

// Returns the number of chain ends written to out[] (the non-roots, in the reference's order), or -1 if the preconditions do
// not hold (the caller then builds and sorts the full list).
template<class C> MPA_HD inline int64_t sorted_chain_ends_sparse(const ChainParams &p, const ChainView &v, const ExtractScratch &S, Pair64 *out)
{
	const int64_t n = v.n_total;
	if (p.min_cnt <= 1 || p.min_sc > p.kmer || n <= 64 || p.kmer < 0 || p.kmer > 255) return -1;
	SparseItem *items = S.items, *moved = S.moved, *merged = S.merged;
	int64_t n_moved = 0, n_merged = 0;
	// the non-roots, ascending in position (f > kmer always holds for a chained anchor; belt and braces)
	int32_t max_f = 0;
	bool odd = false;
	const int64_t n_items = coop_compact<C>(v.m,
		[&](int64_t i) { const bool c = v.pred[i] >= 0; if (c) { if (v.f[i] <= p.kmer) odd = true; if (v.f[i] > max_f) max_f = v.f[i]; } return c; },
		[&](int64_t i, int64_t k) { items[k] = SparseItem{ view_pos(v, i), Pair64{ (uint64_t)(int64_t)v.f[i], (uint64_t)i } }; });
	max_f = C::reduce_max(max_f);
	C::note(0, v.m), C::note(1, n_items);
	if (C::any(odd)) return -1;
	if (max_f < 256) { replay_low_digit<C>(n, p.kmer, items, n_items, out, S.hist); return n_items; }
	if (max_f >= 65536) return -1;
	// ---- two digit levels.  Level 1 (shift 8): bucket 0 = all scores below 256 (the roots among them) is the region
	// [0, c0); walking it moves the high scores found there into the heads of their buckets behind c0, and what they
	// displace from there -- low scores -- into the holes they leave.
	uint32_t *cnt8 = S.hist, *cursor = S.hist + 256, *limit = S.hist + 512, *start = S.hist + 768, *cnt_hi = S.hist + 1024;
	MPA_COOP_FOR(C, d, 256) cnt8[d] = 0;
	C::sync();
	int64_t m_hi = 0;
	MPA_COOP_FOR(C, i, n_items) if (items[i].e.x >= 256) { C::count(&cnt8[items[i].e.x >> 8 & 0xff]); }
	C::sync();
	for (int d = 1; d < 256; ++d) m_hi += cnt8[d];         // (scores below 256 are not counted: bucket 0 stays 0)
	const int64_t c0 = n - m_hi;
	C::note(2, m_hi);
	Pair64 *tail8 = S.tail8;
	MPA_COOP_FOR(C, i, m_hi) tail8[i] = Pair64{ (uint64_t)p.kmer, MPA_ABSENT };
	C::sync();
	{	// slots [c0, n) as they are: every view element there, roots included (their identity never matters)
		int64_t i0;
		if (!v.pos) i0 = c0 < v.m ? c0 : v.m;
		else {                                               // first view element at position >= c0
			const int64_t want = c0 < INT32_MAX ? c0 : INT32_MAX;
			int64_t lo = 0, hi = v.m;
			while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if ((int64_t)v.pos[mid] < want) lo = mid + 1; else hi = mid; }
			i0 = lo;
		}
		MPA_COOP_FOR(C, k, v.m - i0) { const int64_t i = i0 + k; tail8[view_pos(v, i) - c0] = Pair64{ (uint64_t)(int64_t)v.f[i], v.pred[i] >= 0 ? (uint64_t)i : MPA_ABSENT }; }
		C::sync();
	}
	{
		uint32_t at = 0;
		for (int d = 0; d < 256; ++d) { const uint32_t c = d ? cnt8[d] : 0u; start[d] = at, cursor[d] = at, at += c, limit[d] = at; }
	}
	// the walk of region 0: its high scores, in order (a low score inside region 0 stays where it is); what a walk finally
	// displaces is a low score: if it is a chained anchor, it now sits in the hole the walk started from
	int64_t a_end;                                          // items [0, a_end) sit in region 0
	{
		int64_t lo = 0, hi = n_items;
		while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (items[mid].pos < c0) lo = mid + 1; else hi = mid; }
		a_end = lo;
	}
	SparseItem *high = merged;                              // (the merged list is built after the walk)
	const int64_t n_high = coop_compact<C>(a_end, [&](int64_t i) { return items[i].e.x >= 256; }, [&](int64_t i, int64_t k) { high[k] = items[i]; });
	place_in_order<C, 8>(high, n_high, tail8, cursor, 0,
		[&](int64_t i, const Pair64 &old, bool have) {
			const bool keep = have && old.y != MPA_ABSENT;
			const uint64_t mask = C::ballot(keep);
			if (keep) moved[n_moved + C::rank(mask)] = SparseItem{ high[i].pos, old };
			n_moved += C::popc(mask);
		},
		[&](int64_t i, const Pair64 &old) { if (old.y != MPA_ABSENT) moved[n_moved++] = SparseItem{ high[i].pos, old }; });
	C::sync();
	C::mark_time(5);
	walk_buckets<C, 8>(tail8, cursor, limit, 1);            // buckets 1..255 (same walk as pairs_digit_level)
	C::mark_time(6);
	// ---- level 2 inside bucket 0: the low scores at their (possibly new) positions, ascending
	{
		C::sync();
		const int64_t n_low = coop_compact<C>(a_end, [&](int64_t i) { return items[i].e.x < 256; }, [&](int64_t i, int64_t k) { merged[k] = items[i]; });
		// ... merged with the moved ones by position (both ascending, positions distinct).  One lane: backwards, in place.  A team:
		// every element finds its place by itself -- its own rank plus the number of elements of the OTHER list in front of it (a
		// binary search) -- into the item list, which is dead by now and becomes the merged list.
		if (C::width() == 1 || n_moved == 0) {
			int64_t i = n_low - 1, j = n_moved - 1, k = n_low + n_moved - 1;
			while (j >= 0) {
				if (i >= 0 && merged[i].pos > moved[j].pos) merged[k--] = merged[i--];
				else merged[k--] = moved[j--];
			}
		} else {
			C::sync();
			MPA_COOP_FOR(C, i, n_low) {
				const int64_t pos = merged[i].pos;
				int64_t lo = 0, hi = n_moved;
				while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (moved[mid].pos < pos) lo = mid + 1; else hi = mid; }
				items[i + lo] = merged[i];
			}
			MPA_COOP_FOR(C, j, n_moved) {
				const int64_t pos = moved[j].pos;
				int64_t lo = 0, hi = n_low;
				while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (merged[mid].pos < pos) lo = mid + 1; else hi = mid; }
				items[j + lo] = moved[j];
			}
			merged = items;
		}
		n_merged = n_low + n_moved;
	}
	C::sync();
	C::mark_time(7);
	for (int d = 0; d < 256; ++d) cnt_hi[d] = d ? cnt8[d] : 0u;   // (the level-2 passes below reuse the first 768 words of S.hist)
	C::sync();
	{ uint32_t big = 0; for (int d = 1; d < 256; ++d) big = cnt_hi[d] > big ? cnt_hi[d] : big; C::note(3, big), C::note(4, n_merged); }
	int64_t n_out = 0;
	if (c0 > 64) { replay_low_digit<C>(c0, p.kmer, merged, n_merged, out, S.hist); n_out = n_merged; }
	else {                                                  // a short bucket is insertion-sorted: stable
		for (int64_t i = 1; i < n_merged; ++i) {
			const SparseItem x = merged[i];
			int64_t j = i;
			for (; j > 0 && x.e.x < merged[j - 1].e.x; --j) merged[j] = merged[j - 1];
			merged[j] = x;
		}
		for (int64_t i = 0; i < n_merged; ++i) out[n_out++] = merged[i].e;
	}
	// ---- level 2 inside the other buckets (all of it is present): digit pass or insertion sort, as the reference recurses
	// (the buckets are independent: the short ones -- insertion sort, no shared state -- go one to a lane, side by side; the long ones
	// one after the other by the whole team.  start[] / cnt_hi[] sit in the part of S.hist the digit passes do not touch.)
	for (int d = 1 + C::lane(); d < 256; d += C::width()) {
		Pair64 *b = tail8 + start[d];
		if (cnt_hi[d] > 1 && cnt_hi[d] <= 64) pairs_insertion(b, b + cnt_hi[d]);
	}
	C::sync();
	for (int d = 1; d < 256; ++d) {
		Pair64 *b = tail8 + start[d];
		if (cnt_hi[d] > 64) team_digit_level0<C>(b, b + cnt_hi[d], S.hist);
	}
	C::sync();
	MPA_COOP_FOR(C, i, m_hi) out[n_out + i] = tail8[i];   // (buckets 1..255 lie back to back from tail8[0] on)
	C::sync();
	return n_out + m_hi;
}

// Can the view be extracted without the full list of chain ends (one entry per anchor of the problem, absent ones included)?
MPA_HD inline bool chain_ends_sparse_possible(const ChainParams &p, const ChainView &v)
{
	return !(p.min_cnt <= 1 || p.min_sc > p.kmer || v.n_total <= 64 || p.kmer < 0 || p.kmer > 255);
}

// chain.c:8-110 on the result of the forward pass: extraction (best chain end first), then the kept chains laid out in
// ascending anchor order and sorted by their first target position.  a_out: [m], u: [m] (one entry per kept chain,
// score << 32 | anchors).  set_only: the caller wants the anchors of the kept chains as a SET, ascending (the pre-chain,
// map.c:186-192, whose chains are dissolved again by the sort that follows it): no chain list, no layout, and no sort either
// -- the view is in ascending anchor order, so marking the kept anchors and reading the view through gives them sorted.
// Returns 0, or 1 when the full list of chain ends is needed and S.ends_cap is too small for it (nothing written).
template<class C> MPA_HD inline int chain_extract_core(const ChainParams &p, const ChainView &v, const ExtractScratch &S, uint64_t *a_out, int64_t *n_a_out, uint64_t *u, int64_t *n_u_out,
                                                       bool set_only)
{
	int64_t n_u = 0, n_order = 0;
	*n_a_out = 0, *n_u_out = 0;
	C::mark_time(0);
	const int32_t max_drop = p.is_spliced ? INT32_MAX : p.bw;
	const int32_t *f = v.f, *pred = v.pred;
	int32_t *mark = S.mark, *order = S.order;
	Pair64 *ends = S.ends;
	int64_t n_ends = sorted_chain_ends_sparse<C>(p, v, S, ends);
	if (n_ends < 0) {
		// the full list: absent anchors are roots (score kmer); they can only be left out when min_cnt rejects them anyway
		if (v.n_total > S.ends_cap) return 1;
		n_ends = 0;
		int64_t vi = 0;
		for (int64_t q = 0; q < v.n_total; ++q) {
			if (vi < v.m && view_pos(v, vi) == q) {
				if (f[vi] >= p.min_sc) ends[n_ends++] = Pair64{ (uint64_t)(int64_t)f[vi], (uint64_t)vi };
				++vi;
			} else if (p.kmer >= p.min_sc) ends[n_ends++] = Pair64{ (uint64_t)(int64_t)p.kmer, MPA_ABSENT };
		}
		C::sync();
		sort_pairs_by_x_core(ends, ends + n_ends, S.stack, S.hist);
		C::sync();
	}
	C::mark_time(1);
	if (n_ends == 0) return 0;
	// ---- the greedy extraction (chain.c:37-75): chain ends best first; a chain walks back until it meets a claimed anchor.
	// Chains only ever touch anchors of their own predecessor TREE, so ends of different trees do not depend on each other:
	// the team takes C::width() consecutive ends at a time, one per lane, and only lanes whose ends sit in the same tree take
	// turns, in the order of the list (the lowest lane goes first).  With a team of one this is the sequential loop.
	const bool team = C::width() > 1;
	int32_t *root = (int32_t*)S.items, *owner = root + v.m;      // (the item lists of the sort replay are dead by now)
	MPA_COOP_FOR(C, i, v.m) {
		mark[i] = 0;
		if (team) owner[i] = INT32_MAX, root[i] = pred[i] < 0 ? (int32_t)i : pred[i];
	}
	C::sync();
	if (team) {                                               // the root of every anchor's tree, by pointer jumping
		for (;;) {
			bool changed = false;
			MPA_COOP_FOR(C, i, v.m) { const int32_t r = root[i], rr = root[r]; if (rr != r) root[i] = rr, changed = true; }
			C::sync();
			if (!C::any(changed)) break;
		}
	}
	C::mark_time(2);
	for (int64_t kb = n_ends - 1; kb >= 0; kb -= C::width()) {
		const int64_t k = kb - C::lane();
		const bool have = k >= 0 && ends[k >= 0 ? k : 0].y != MPA_ABSENT;
		const int64_t tail = have ? (int64_t)ends[k].y : 0;
		const int32_t tail_f = have ? (int32_t)ends[k].x : 0;
		const int32_t my_root = team && have ? root[tail] : 0;
		bool pending = have, accepted = false;
		int64_t cnt = 0;
		int32_t sc = 0;
		while (C::any(pending)) {
			if (team) { if (pending) C::atomic_min(&owner[my_root], (int32_t)C::lane()); C::sync(); }
			const bool win = pending && (!team || owner[my_root] == (int32_t)C::lane());
			// An unclaimed anchor without predecessor is a one-anchor chain.  When min_cnt rejects those, the only trace it
			// would leave is its own claim mark, which nothing reads any more: every anchor chained onto it scores above
			// kmer == f[tail] and has therefore been visited already.
			if (win && mark[tail] == 0 && !(pred[tail] < 0 && p.min_cnt > 1)) {
				// where does the best-scoring suffix of this chain start?  walk back until a claimed anchor or the chain's root
				int64_t cut = tail, walk = tail, stop = -1;
				int32_t best_suffix = 0;
				do {
					mark[walk] = 2;
					stop = walk = pred[walk];
					const int32_t sfx = walk < 0 ? tail_f : tail_f - f[walk];
					if (sfx > best_suffix) best_suffix = sfx, cut = walk;
					else if (best_suffix - sfx > max_drop) break;
				} while (walk >= 0 && mark[walk] == 0);
				for (int64_t x = tail; x >= 0 && x != stop; x = pred[x]) mark[x] = 0;
				// claim the anchors of the suffix (a rejected chain keeps them claimed)
				int64_t x = tail;
				for (; x != cut; x = pred[x]) mark[x] = 1, ++cnt;
				sc = x < 0 ? tail_f : tail_f - f[x];
				accepted = sc >= p.min_sc && cnt > 0 && cnt >= p.min_cnt;
			}
			C::sync();
			if (team && win) owner[my_root] = INT32_MAX;
			pending = pending && !win;
			C::sync();
		}
		// the kept chains of this group in the order of the list (= lane order): u, and their anchors from the END backwards
		const uint64_t acc = C::ballot(accepted);
		int64_t tot = 0;
		const int64_t off = C::scan_excl(accepted ? cnt : 0, &tot);
		if (accepted) {
			u[n_u + C::rank(acc)] = (uint64_t)sc << 32 | (uint64_t)cnt;
			int64_t x = tail;
			for (int64_t j = 0; j < cnt; ++j, x = pred[x]) order[n_order + off + j] = (int32_t)x;
		}
		n_u += C::popc(acc), n_order += tot;
	}
	C::sync();
	C::mark_time(3);
	if (n_u == 0) return 0;
	if (set_only) {
		// (mark: 1 = claimed; the anchors of the kept chains are exactly those in `order`)
		uint8_t *kept = S.kept;
		MPA_COOP_FOR(C, i, v.m) kept[i] = 0;
		C::sync();
		MPA_COOP_FOR(C, i, n_order) kept[order[i]] = 1;
		C::sync();
		*n_a_out = coop_compact<C>(v.m, [&](int64_t i) { return kept[i] != 0; }, [&](int64_t i, int64_t k) { a_out[k] = v.a[i]; });
		C::mark_time(4);
		return 0;
	}
	// ---- lay the chains out in ascending anchor order, chains sorted by their first target position (chain.c:77-110)
	uint64_t *packed = S.packed, *u_sorted = S.u_sorted;
	Pair64 *first = S.first;
	{
		int64_t k = 0;
		for (int64_t c = 0; c < n_u; ++c) {                      // (the chains one after the other, a chain's anchors shared out over the lanes)
			const int64_t cnt = (int64_t)(uint32_t)u[c];
			MPA_COOP_FOR(C, j, cnt) packed[k + j] = v.a[order[k + cnt - 1 - j]];
			if (C::lane() == 0) first[c] = Pair64{ v.a[order[k + cnt - 1]] >> 32, (uint64_t)k << 32 | (uint64_t)c };
			k += cnt;
		}
	}
	C::sync();
	sort_pairs_by_x_core(first, first + n_u, S.stack, S.hist);
	C::sync();
	int64_t k = 0;
	for (int64_t c = 0; c < n_u; ++c) {
		const uint32_t which = (uint32_t)first[c].y;
		const int64_t cnt = (int64_t)(uint32_t)u[which], from = (int64_t)(first[c].y >> 32);
		if (C::lane() == 0) u_sorted[c] = u[which];
		MPA_COOP_FOR(C, j, cnt) a_out[k + j] = packed[from + j];
		k += cnt;
	}
	C::sync();
	MPA_COOP_FOR(C, c, n_u) u[c] = u_sorted[c];
	C::sync();
	*n_a_out = k, *n_u_out = n_u;
	C::mark_time(4);
	return 0;
}

} // namespace mpa
