// host_map.cpp -- the batched replacement of mp_map() (map.c:143-240) and mp_align() (align.c:239-342).
//
// The reference maps one protein per thread from seeding to CIGAR.  Here a whole batch of proteins moves
// through the pipeline stage by stage so that every spliced-DP call of the batch can be issued to the GPU
// together:
//
//   stage A (host, threaded over queries)   sketch -> index lookup -> sort -> pre-chain -> chain ->
//                                           regions -> refine (k=5 re-seeding + re-chaining) -> select ->
//                                           extension limits -> per-region alignment plan
//   DP round 1 (GPU)   left + right extensions of every region            ns_global_gs16b EXT_LEFT/EXT_RIGHT
//                      + every gap between kept anchors with traceback (they do not depend on the extensions,
//                        so they run concurrently with the long extension windows)
//   DP round 2 (GPU)   the io_end retries of align.c:290-296,324-330
//   DP round 3 (GPU)   the two spans accepted by the extensions, with traceback
//   stage B (host, threaded)                CIGAR assembly, mp_extra_* statistics, final ranking
//
// The object is a plain stage machine (mpa_batch_*): it hands out DP tasks and takes DP results, and does
// not care who executes them.  mpa_map_batch() drives it with the HIP executor; the CPU-only tests drive
// it with the oracle to check the host logic against the reference's output.
#include <algorithm>
#include <atomic>
#include <cassert>
#include <cstring>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <functional>
#include "host_core.h"

namespace mpa {

struct Segment {                     // one mp_align_seq() call (align.c:62-80)
	int32_t ne0, ne1, ae0, ae1;      // window-relative nt span (w.r.t. vs0) and protein span
	int32_t task = -1;               // DP task index within its round, -1 = ungapped shortcut
	int32_t score = 0;
	std::vector<uint32_t> cigar;     // filled when the DP result arrives
};

struct AlignPlan {                   // the state of one mp_align() call
	int32_t reg = -1;
	int64_t as = 0, ae = 0, vs0 = 0, vs1 = 0;
	int32_t i0 = 0, as1 = 0;
	int32_t t_left = -1, t_left2 = -1, t_right = -1, t_right2 = -1;
	int32_t l_nt = 0, l_aa = 0, r_nt = 0, r_aa = 0;
	int64_t mid_ve = 0;              // r->ve / r->qe after the gap-patching loop
	int32_t mid_qe = 0;
	bool has_right = false;
	Segment left_span, right_span;   // the spans accepted by the two extensions, re-aligned with traceback (round 3)
	bool has_left_span = false, has_right_span = false;
	std::vector<Segment> gaps;       // gaps between consecutive kept anchors: independent of the extensions (round 1)
};

struct QueryState {
	int32_t qid = 0, qlen = 0;
	const char *seq = nullptr;
	std::vector<Region> regs;
	std::vector<AlignPlan> plans;
	std::vector<uint64_t> seeds;         // kept seeds (query position << 32 | index bucket), ascending; only between the seeding sub-stages
	std::vector<uint64_t> ext_refine;    // per region: window extension limits of the refinement (between the planning sub-stages)
	int64_t win0 = 0;                    // first refinement window of this query in the batch's list (device refinement)
	int64_t n_anchor = 0;                // anchors those seeds expand to
	std::vector<mpa_dp_task_t> local1;   // this query's round-1 DP tasks (plan fields index into it)
	int64_t base1 = 0;                   // where local1 starts in the batch's round-1 task array
};

} // namespace mpa

using namespace mpa;

struct mpa_result_s {
	int32_t n_seq = 0;
	std::vector<mpa_hit_t> hits;
	std::vector<int64_t> hit_off;
	std::vector<uint32_t> cigars;
	std::vector<mpa_feat_t> feats;
};

struct mpa_batch_s {
	const mpa_idx_s *mi = nullptr;
	mpa_mapopt_t opt;
	mpa_dpopt_t dpopt;
	mpa_qbatch_t q;
	int n_threads = 1;
	int round = 0;                   // 0: before round 1; 1, 2: tasks of that round are out (extensions + gaps; accepted spans); 4: done
	std::vector<QueryState> qs;
	std::vector<mpa_dp_task_t> tasks;
	bool seeded_on_device = false;   // between the two seeding phases: `sparse` holds the device's pre-chain result
	PrechainSparse sparse;
	// between the host half of seeding (sketch, lookup, cut-off: batch_sketch_phase) and its device half: the kept seeds of every
	// query as occurrence lists (one job per seed) and the first anchor of every query; empty = the batch is seeded on the host
	std::vector<SeedJob> seed_jobs;
	std::vector<int64_t> seed_qfirst;
};

namespace mpa {

// A small persistent worker pool: the stage machine calls parallel_for several times per batch and thread
// creation would otherwise cost more than some of the stages.
class WorkerPool {
public:
	// never destroyed: workers outlive main().  One pool per lane, so that the pipeline stages of mpa_map_batches()
	// (seeding of the next batch / DP-round bookkeeping / flattening + formatting of the previous one) can each run
	// their own parallel region at the same time.
	static const int kPools = 24 * 8;                   // 24 lanes per pipeline (mpa_map_batches), up to eight pipelines per process (mpa_map_batches_multi)
	static WorkerPool &get(int lane) { static WorkerPool *p = new WorkerPool[kPools]; return p[lane % kPools]; }
	void run(int n_threads, int64_t n, const std::function<void(int64_t)> &fn) {
		if (n_threads <= 1 || n <= 1) { for (int64_t i = 0; i < n; ++i) fn(i); return; }
		std::unique_lock<std::mutex> run_lock(run_mu_);              // one parallel region at a time
		ensure(n_threads - 1);
		{
			std::lock_guard<std::mutex> g(mu_);
			fn_ = &fn, n_ = n, next_.store(0), active_ = std::min<int>(n_threads - 1, (int)workers_.size()), pending_ = active_, ++epoch_;
		}
		cv_.notify_all();
		work();
		std::unique_lock<std::mutex> g(mu_);
		done_cv_.wait(g, [&] { return pending_ == 0; });
		fn_ = nullptr;
	}
private:
	void ensure(int k) {
		while ((int)workers_.size() < k) {
			const int id = (int)workers_.size();
			workers_.emplace_back([this, id] { loop(id); });
			workers_.back().detach();
		}
	}
	void work() { CpuSpan cs(label_); for (;;) { int64_t i = next_.fetch_add(1); if (i >= n_) break; (*fn_)(i); } }
public:
	const char *label_ = "pool (outside a stream)";     // what the CPU time of this pool's parallel regions is booked under (MPA_TIMING)
private:
	void loop(int id) {
		uint64_t seen = 0;
		for (;;) {
			std::unique_lock<std::mutex> g(mu_);
			cv_.wait(g, [&] { return epoch_ != seen && id < active_; });
			seen = epoch_;
			g.unlock();
			work();
			g.lock();
			if (--pending_ == 0) done_cv_.notify_all();
		}
	}
	std::mutex mu_, run_mu_;
	std::condition_variable cv_, done_cv_;
	std::vector<std::thread> workers_;
	const std::function<void(int64_t)> *fn_ = nullptr;
	int64_t n_ = 0;
	std::atomic<int64_t> next_{0};
	int active_ = 0, pending_ = 0;
	uint64_t epoch_ = 0;
};

// MPA_TRACE=1: one line per pipeline event of mpa_map_batches() with a time stamp (ms since the call began) -- tools/gantt.py
// turns them into a per-stage timeline
static double g_trace_t0 = 0;
static bool trace_on() { static int on = -1; if (on < 0) { const char *e = getenv("MPA_TRACE"); on = e && *e && *e != '0'; } return on != 0; }
static void trace(const char *stage, int32_t batch, const char *what)
{
	if (trace_on()) fprintf(stderr, "[mpa-trace] %10.3f %-8s batch %3d %s\n", now_ms() - g_trace_t0, stage, batch, what);
}

static thread_local int tl_pool_lane = 0;
static thread_local int tl_thread_div = 1;          // pipeline stages with light host work take a fraction of the threads
template<typename F> static void parallel_for(int n_threads, int64_t n, F fn)
{
	std::function<void(int64_t)> f = fn;
	WorkerPool::get(tl_pool_lane).run(std::max(1, n_threads / tl_thread_div), n, f);
}

// ------------------------------------------------------------------------------------------------
// stage A
// ------------------------------------------------------------------------------------------------

// per-substage CPU-time accumulators of stage A (thread CPU time summed over the workers; printed with MPA_TIMING=1)
static std::atomic<int64_t> g_acc[16];
// wall-clock microseconds and calls of the stream pipeline's stages since the last reset (mpa_stage_clocks): 0 seeding, 1 planning,
// 2 DP rounds of a batch (all of them, host round trips included), 3 output (flatten + text), 4 sketch (host half of seeding)
static std::atomic<int64_t> g_stage_us[5], g_stage_n[5];
struct StageClock {
	int k; double t0;
	explicit StageClock(int k_) : k(k_), t0(now_ms()) {}
	~StageClock() { g_stage_us[k] += (int64_t)((now_ms() - t0) * 1000.0), ++g_stage_n[k]; }
};
static const char *const kAccName[16] = { "A.sketch+lookup", "A.sort anchors", "A.pre-chain", "A.chain", "A.regions", "A.refine", "A.plan", "A.round-1 plan",
	"  refine: fetch window", "  refine: sketch nt4", "  refine: sketch prot+sort", "  refine: pairs+sort", "  refine: chain", "  refine: n regions", "  anchors (x1000, all queries / threads)", "  pre-chain survivors (x1000)" };
// (thread CPU time, not wall time: under a CPU quota the wall time of a thread says how often it was throttled, not what it cost)
struct AccTimer {
	int k; double t0; bool on;
	AccTimer(int k_) : k(k_), t0(0), on(timing_on()) { if (on) t0 = thread_cpu_ms(); }
	~AccTimer() { if (on) g_acc[k] += (int64_t)((thread_cpu_ms() - t0) * 1e6); }
};

// mp_cal_max_occ (map.c:126-141): boxplot-style cutoff on the occurrence counts of the query's seeds
static int32_t occurrence_cutoff(const mpa_idx_s *mi, const std::vector<uint64_t> &seeds)
{
	const int64_t n = (int64_t)seeds.size(), n_bucket = (int64_t)mi->ki.size();
	std::vector<uint64_t> cnt(n);
	for (int64_t i = 0; i < n; ++i) {
		const int64_t b = (int64_t)(seeds[i] >> 32);
		const int64_t en = b + 1 < n_bucket ? mi->ki[b + 1] : mi->n_kb;
		cnt[i] = (uint64_t)(en - mi->ki[b]);
	}
	sort_u64(cnt.data(), cnt.data() + n);
	const uint64_t q25 = cnt[(int64_t)(n * .25 + .499)], q75 = cnt[(int64_t)(n * .75 + .499)];
	return (int32_t)(q75 + (q75 - q25) * 1.5 + 10.);
}

// mp_refine_reg (map.c:32-111): re-seed the region's window with all 5-mers at base resolution, re-chain
// The query side of the refinement seeding, built once per query: its k-mers sorted by hash and a bitmap over the
// hash space (2^(4k) bits; one per worker thread, reused) with the query's hashes set.
struct RefineQuery {
	std::vector<uint64_t> qk;
	std::vector<uint64_t> *filter = nullptr;
	int32_t kmer = 0;
	static std::vector<uint64_t> &thread_filter(int32_t kmer) {
		static thread_local std::vector<uint64_t> f;
		const size_t words = ((size_t)1 << (4 * kmer)) / 64 + 1;
		if (f.size() < words) f.assign(words, 0);
		return f;
	}
	std::vector<uint32_t> words;                      // packed k-mer words of the query (bitmap keys)
	RefineQuery(const char *aa, int32_t l_aa, int32_t k, bool want_filter = true) : kmer(k) {
		sketch_protein(aa, l_aa, k, 0, qk);
		sort_u64(qk.data(), qk.data() + qk.size());
		if (k <= 6 && want_filter) {
			filter = &thread_filter(k);
			const uint8_t *aa13 = tab_aa13();
			const uint32_t mask = (1U << 4 * k) - 1;
			uint32_t w = 0;
			for (int32_t i = 0, run = 0; i < l_aa; ++i) {
				const uint32_t c = aa13[(uint8_t)aa[i]];
				if (c >= 14) { run = 0, w = 0; continue; }
				w = (w << 4 | c) & mask;
				if (++run >= k) words.push_back(w), (*filter)[w >> 6] |= 1ULL << (w & 63);
			}
		}
	}
	~RefineQuery() { if (filter) for (uint32_t w : words) (*filter)[w >> 6] = 0; }
};

// The anchors of a region's refinement (map.c:36-79): every (window position, query position) pair of equal k = kmer2 k-mers,
// sorted.  dev_hits != nullptr: the window was scanned on the device (dev_refine_scan); its k-mer hits are (hash << 32 | position).
static void refine_region_pairs(const mpa_idx_s *mi, const mpa_mapopt_t &opt, const RefineQuery &rq, const Region &r, int32_t extl, int32_t extr,
                                const uint64_t *dev_hits, int64_t n_dev_hits, std::vector<uint64_t> &a)
{
	const int32_t kmer = opt.kmer2;
	const int64_t ctg_len = mi->ctg[r.vid >> 1].len;
	const int64_t as = r.vs > extl ? r.vs - extl : 0;
	const int64_t ae = r.ve + extr < ctg_len ? r.ve + extr : ctg_len;
	static thread_local std::vector<uint8_t> nt;
	static thread_local std::vector<uint64_t> sd;
	g_acc[13] += 1000000LL * 1;                                // (count of refined regions, printed /1e6/threads)
	if (dev_hits) {                                            // the window was scanned on the device
		AccTimer tm(11);
		sd.assign(dev_hits, dev_hits + n_dev_hits);
		refine_pairs_from_hits(sd, rq.qk, opt.max_ava, a);
		return;
	}
	nt.resize((size_t)(ae - as));
	{ AccTimer tm(8); fetch_nt(mi, (int32_t)r.vid, as, ae, nt.data()); }
	if (rq.filter) {
		AccTimer tm(9);
		refine_seed_pairs(nt.data(), ae - as, mi->opt.min_aa_len, kmer, rq.qk, rq.filter->data(), opt.max_ava, sd, a);
	} else {
		// large k: the reference's own formulation -- sort the window's and the query's (bit 31 set) k-mers together;
		// one sort of the merged list gives the same groups (query entries never equal reference entries)
		AccTimer tm(10);
		sketch_nt4(nt.data(), ae - as, mi->opt.min_aa_len, kmer, 0, 0, 0, sd, false);
		for (uint64_t x : rq.qk) sd.push_back(x | 1ULL << 31);
		sort_u64(sd.data(), sd.data() + sd.size());
		sd.erase(std::unique(sd.begin(), sd.end()), sd.end());
		a.clear();
		const size_t n = sd.size();
		for (size_t k = 0, i = 1; i <= n; ++i) {
			if (i < n && sd[k] >> 32 == sd[i] >> 32) continue;
			size_t j = k;
			while (j < i && !(sd[j] >> 31 & 1)) ++j;
			const int64_t n1 = (int64_t)(j - k), n2 = (int64_t)(i - j);
			if (n1 > 0 && n2 > 0 && (int32_t)n1 * (int32_t)n2 <= opt.max_ava)
				for (size_t i1 = k; i1 < j; ++i1)
					for (size_t i2 = j; i2 < i; ++i2)
						a.push_back((uint64_t)(uint32_t)sd[i1] << 32 | ((uint32_t)sd[i2] << 1 >> 1));
			k = i;
		}
		sort_u64(a.data(), a.data() + a.size());
	}
}

static inline ChainParams refine_chain_params(const mpa_mapopt_t &opt)
{
	return ChainParams{ opt.max_intron, opt.max_gap, opt.bw, opt.max_chn_max_skip, opt.max_chn_iter, opt.min_chn_cnt, opt.min_chn_sc,
	                    opt.chn_coef_log, !(opt.flag & MPA_MF_NO_SPLICE), opt.kmer2, 0 };
}

static void refine_region_from_chains(const mpa_idx_s *mi, const mpa_mapopt_t &opt, Region &r, int32_t extl, const uint64_t *u, int64_t n_u, const uint64_t *a);
// The refinement chain of a region over those anchors and what mp_refine_reg() makes of it (map.c:81-111).  f/pred != nullptr:
// the forward pass of the chain ran on the device (dev_chain_forward).
static void refine_region_chain(const mpa_idx_s *mi, const mpa_mapopt_t &opt, Region &r, int32_t extl, std::vector<uint64_t> &a,
                                const int32_t *f = nullptr, const int32_t *pred = nullptr)
{
	const ChainParams cp = refine_chain_params(opt);
	std::vector<uint64_t> u;
	{ AccTimer tm(12); if (f) chain_anchors_after_forward(cp, f, pred, a, u); else chain_anchors(cp, a, u); }
	refine_region_from_chains(mi, opt, r, extl, u.data(), (int64_t)u.size(), a.data());
}

// ... what mp_refine_reg() makes of the window's chains (map.c:89-111): u[n_u] = score << 32 | anchors per chain, a = the chains' anchors
static void refine_region_from_chains(const mpa_idx_s *mi, const mpa_mapopt_t &opt, Region &r, int32_t extl, const uint64_t *u, int64_t n_u, const uint64_t *a)
{
	const int32_t kmer = opt.kmer2;
	const int64_t as = r.vs > extl ? r.vs - extl : 0;
	if (n_u == 0) { r.cnt = 0, r.off = -1, r.a.clear(); return; }
	// the first chain with the highest score
	size_t best = 0, skip = 0;
	for (size_t i = 1; i < (size_t)n_u; ++i) if ((int32_t)(u[best] >> 32) < (int32_t)(u[i] >> 32)) best = i;
	for (size_t i = 0; i < best; ++i) skip += (uint32_t)u[i];
	const size_t n_a = (uint32_t)u[best];
	r.a.assign(a + skip, a + skip + n_a);
	r.chn_sc = (int32_t)(u[best] >> 32);
	r.cnt = (int32_t)n_a, r.off = 0;
	r.qs = (int32_t)(uint32_t)r.a[0] - (kmer - 1);
	r.qe = (int32_t)(uint32_t)r.a[n_a - 1] + 1;
	r.vs = as + (int64_t)(r.a[0] >> 32) + 1 - 3 * kmer;
	r.ve = as + (int64_t)(r.a[n_a - 1] >> 32) + 1;
	for (uint64_t &x : r.a) x = (uint64_t)((int64_t)(x >> 32) + as - r.vs) << 32 | (x << 32 >> 32);
	r.chn_sc_ungap = chain_score_ungapped(r.a.data(), r.cnt, kmer);
}

static void refine_region(const mpa_idx_s *mi, const mpa_mapopt_t &opt, const RefineQuery &rq, Region &r, int32_t extl, int32_t extr,
                          const uint64_t *dev_hits = nullptr, int64_t n_dev_hits = 0)
{
	static thread_local std::vector<uint64_t> a;
	refine_region_pairs(mi, opt, rq, r, extl, extr, dev_hits, n_dev_hits, a);
	refine_region_chain(mi, opt, r, extl, a);
}

// mp_filter_seed (align.c:6-31): mark (bit 31) anchors inside tight, in-frame runs, trimmed at both ends
static void mark_reliable_anchors(std::vector<uint64_t> &a, int32_t max_aa_dist, int32_t min_cnt, int32_t kmer2, int32_t trim)
{
	const int32_t cnt = (int32_t)a.size();
	for (int32_t i = 0; i < cnt; ++i) {
		int32_t j = i + 1;
		for (; j < cnt; ++j) {
			const int32_t dx = (int32_t)(a[j] >> 32) - (int32_t)(a[j - 1] >> 32), dy = (int32_t)a[j] - (int32_t)a[j - 1];
			if (dx % 3 != 0 || dx > max_aa_dist * 3 || dy > max_aa_dist) break;
		}
		if (j - i < min_cnt) continue;
		int32_t k = j - 2, t = (int32_t)a[j - 1];
		while (k >= i && t - (int32_t)a[k] < trim) --k;
		t = (int32_t)a[i] + 1 - kmer2;
		while (i < k && (int32_t)a[i] + 1 - t < trim) ++i;
		for (; i <= k; ++i) a[i] |= 1ULL << 31;
		i = j - 1;
	}
}

static inline uint8_t codon_aa(const uint8_t *nt)
{
	return nt[0] > 3 || nt[1] > 3 || nt[2] > 3 ? 21 : tab_codon()[nt[0] << 4 | nt[1] << 2 | nt[2]];
}

// first half of mp_align (align.c:239-301,316-323): window, left/right extension tasks
static bool plan_alignment(const mpa_batch_s *b, QueryState &qs, int32_t ridx, int32_t extl0, int32_t extr0, AlignPlan &pl)
{
	const mpa_mapopt_t &opt = b->opt;
	Region &r = qs.regs[ridx];
	mark_reliable_anchors(r.a, 6, 3, opt.kmer2, opt.kmer2 + 1);
	int32_t i0 = 0;
	while (i0 < r.cnt && !(r.a[i0] >> 31 & 1)) ++i0;
	if (i0 == r.cnt) { r.cnt = 0; return false; }
	int32_t extl = opt.max_ext, extr = opt.max_ext;
	if (r.qs >= 10) extl = opt.max_intron / 2;
	if (qs.qlen - r.qe >= 10) extr = opt.max_intron / 2;
	if (extl0 > 0) extl = std::min(extl, extl0);
	if (extr0 > 0) extr = std::min(extr, extr0);
	const int64_t ctg_len = b->mi->ctg[r.vid >> 1].len;
	pl.reg = ridx, pl.i0 = i0;
	pl.as = r.vs > extl ? r.vs - extl : 0;
	pl.ae = r.ve + extr < ctg_len ? r.ve + extr : ctg_len;
	pl.vs0 = r.vs;
	pl.vs1 = pl.vs0 + (int64_t)(r.a[i0] >> 32) + 1;
	pl.as1 = (int32_t)((uint32_t)r.a[i0] << 1 >> 1) + 1;
	// where the gap-patching loop will leave r->ve / r->qe: the end of the last kept anchor
	int32_t ne = 0, ae = 0;
	for (int32_t i = i0; i < r.cnt; ++i)
		if (r.a[i] >> 31 & 1) ne = (int32_t)(r.a[i] >> 32) + 1, ae = (int32_t)((uint32_t)r.a[i] << 1 >> 1) + 1;
	pl.mid_ve = ne + pl.vs0, pl.mid_qe = ae;
	pl.has_right = pl.mid_qe < qs.qlen && pl.mid_ve < pl.ae;
	return true;
}

static mpa_dp_task_t make_task(const QueryState &qs, const Region &r, int64_t nt_off, int64_t nl, int32_t aa_off, int32_t al, int32_t flag, int32_t io)
{
	mpa_dp_task_t t;
	t.nt_off = nt_off, t.vid = (int32_t)r.vid, t.nl = (int32_t)nl, t.qid = qs.qid, t.aa_off = aa_off, t.al = al, t.flag = flag, t.io = io, t.tag = 0;
	return t;
}

// mp_extra_stop / mp_extra_start (align.c:214-237)
static int32_t dist_to_stop(const mpa_idx_s *mi, const Region &r, int64_t ae)
{
	uint8_t c[3];
	for (int64_t j = r.ve; j + 2 < ae; j += 3) {
		fetch_nt(mi, (int32_t)r.vid, j, j + 3, c);
		if (codon_aa(c) == 20) return (int32_t)(j - r.ve);
	}
	return -1;
}

static int32_t dist_to_start(const mpa_idx_s *mi, const Region &r, int64_t as, int64_t ae)
{
	uint8_t c[3];
	for (int64_t j = r.vs; j >= as && j + 2 < ae; j -= 3) {
		fetch_nt(mi, (int32_t)r.vid, j, j + 3, c);
		const uint8_t aa = codon_aa(c);
		if (aa == 20) break;
		if (aa == 12) return (int32_t)(r.vs - j);
	}
	return -1;
}

// mp_extra_cal (align.c:82-201): walk the CIGAR once, accumulate identity/score statistics and exon features
static void summarize_alignment(const mpa_idx_s *mi, const mpa_mapopt_t &opt, Region &r, const char *aa_full, int32_t qlen, int64_t win_end)
{
	const int64_t l_nt = win_end - r.vs;
	std::vector<uint8_t> ntv((size_t)std::max<int64_t>(l_nt, 1));
	fetch_nt(mi, (int32_t)r.vid, r.vs, win_end, ntv.data());
	const uint8_t *nt = ntv.data(), *aa20 = tab_aa20();
	const char *aa = aa_full + r.qs;
	const char *nt_char = "ACGTN";
	const bool has_stop = r.qe == qlen && r.dist_stop == 0;
	int32_t n_intron = 0;
	for (uint32_t c : r.cigar) { uint32_t op = c & 0xf; n_intron += (op == 3 || op == 12 || op == 13); }
	r.n_exon = n_intron + 1;
	r.feat.assign((size_t)r.n_exon + (has_stop ? 1 : 0), Feat());
	r.blen = r.n_iden = r.n_plus = r.n_fs = r.n_stop = r.dp_max = 0;
	int32_t nl = 0, al = 0, ft = 0;
	int32_t blen0 = 0, iden0 = 0, score0 = 0, fs0 = 0, stop0 = 0, phase0 = 0, qs0 = r.qs;
	int64_t vs0 = r.vs;
	char acc0[2] = { 0, 0 };
	auto score_codon = [&](uint8_t nt_aa, int32_t aa_idx) {
		const uint8_t q = aa20[(uint8_t)aa[aa_idx]];
		const int32_t s = opt.mat[nt_aa * opt.asize + q];
		r.n_stop += nt_aa == 20, r.n_iden += nt_aa == q, r.n_plus += s > 0, r.dp_max += s;
	};
	auto close_exon = [&](Feat &f, int64_t ve) {
		f.type = 0, f.vs = vs0, f.ve = ve, f.qs = qs0, f.qe = r.qs + al, f.phase = (int16_t)phase0;
		f.blen = r.blen - blen0, f.n_iden = r.n_iden - iden0, f.n_fs = r.n_fs - fs0, f.n_stop = r.n_stop - stop0, f.score = r.dp_max - score0;
		if (ft > 0) f.acceptor[0] = acc0[0], f.acceptor[1] = acc0[1];
	};
	for (uint32_t c : r.cigar) {
		const int32_t op = c & 0xf, len = (int32_t)(c >> 4);
		if (op == 0) {                                             // M
			for (int32_t l = 0; l < len; ++l) score_codon(codon_aa(nt + nl + 3 * l), al + l);
			nl += len * 3, al += len, r.blen += len * 3;
		} else if (op == 1) {                                      // I
			r.dp_max -= opt.go + opt.ge * len;
			al += len, r.blen += len * 3;
		} else if (op == 2) {                                      // D: in-frame stop codons inside deletions count
			for (int32_t l = 0; l < len; ++l) r.n_stop += codon_aa(nt + nl + 3 * l) == 20;
			r.dp_max -= opt.go + opt.ge * len;
			nl += len * 3, r.blen += len * 3;
		} else if (op == 10) {                                     // F
			r.dp_max -= opt.fs;
			nl += len, r.blen += len, r.n_fs++;
		} else if (op == 11) {                                     // G
			r.dp_max -= opt.fs;
			nl += len, ++al, r.blen += 3, r.n_fs++;
		} else if (op == 3 || op == 12 || op == 13) {              // N / U / V introns
			if (op != 3) {                                         // the codon split by a phase-1/2 intron
				uint8_t cod[3];
				if (op == 12) cod[0] = nt[nl], cod[1] = nt[nl + len - 2], cod[2] = nt[nl + len - 1];
				else cod[0] = nt[nl], cod[1] = nt[nl + 1], cod[2] = nt[nl + len - 1];
				score_codon(codon_aa(cod), al);
				r.blen += 3;
			}
			Feat &f = r.feat[ft];
			int64_t ve;
			if (op == 3) ve = r.vs + nl;
			else if (op == 12) ve = r.vs + nl + 1;
			else ve = r.vs + nl + 2;
			close_exon(f, ve);
			++ft;
			if (op == 3) vs0 = r.vs + nl + len, phase0 = 0;
			else if (op == 12) vs0 = r.vs + nl + len - 2, phase0 = 2;
			else vs0 = r.vs + nl + len - 1, phase0 = 1;
			f.donor[0] = f.ve - r.vs < l_nt ? nt_char[nt[f.ve - r.vs]] : '.';
			f.donor[1] = f.ve - r.vs + 1 < l_nt ? nt_char[nt[f.ve - r.vs + 1]] : '.';
			qs0 = f.qe, fs0 = r.n_fs, stop0 = r.n_stop, score0 = r.dp_max, blen0 = r.blen, iden0 = r.n_iden;
			acc0[0] = vs0 - r.vs >= 2 ? nt_char[nt[vs0 - r.vs - 2]] : '.';
			acc0[1] = vs0 - r.vs >= 1 ? nt_char[nt[vs0 - r.vs - 1]] : '.';
			nl += len, al += op != 3;
		}
	}
	close_exon(r.feat[ft], r.vs + nl);
	++ft;
	if (has_stop) {
		Feat &f = r.feat[ft++];
		f.type = 1, f.vs = r.ve, f.ve = r.ve + 3, f.qs = f.qe = r.qe + al, f.phase = 0, f.n_fs = 0, f.blen = 3, f.n_iden = 0;
	}
	assert(nl == r.ve - r.vs && al == r.qe - r.qs);
}

static void append_cigar(std::vector<uint32_t> &cig, uint32_t op, int32_t len)   // ns_push_cigar, nasw.h:141-152
{
	if (!cig.empty() && (cig.back() & 0xf) == op && op != 10 && op != 11) cig.back() += (uint32_t)len << 4;
	else cig.push_back((uint32_t)len << 4 | op);
}

// mp_map() in sub-stages.  (1) the query's seeds that are not too frequent (map.c:152-170)
static void stage_seeds(mpa_batch_s *b, QueryState &qs)
{
	const mpa_idx_s *mi = b->mi;
	const mpa_mapopt_t &opt = b->opt;
	const int64_t n_bucket = (int64_t)mi->ki.size();
	static thread_local std::vector<uint64_t> sd;
	AccTimer tm(0);
	sketch_protein(qs.seq, qs.qlen, mi->opt.kmer, mi->opt.mod_bit, sd);
	sort_u64(sd.data(), sd.data() + sd.size());
	int32_t max_occ = opt.max_occ;
	if (sd.size() >= 8) max_occ = std::min(max_occ, occurrence_cutoff(mi, sd));
	// kept in ascending query position so that the anchor sort only has to look at the block ids
	qs.seeds.clear(), qs.n_anchor = 0;
	for (uint64_t s : sd) {
		const int64_t bkt = (int64_t)(s >> 32), st = mi->ki[bkt], en = bkt + 1 < n_bucket ? mi->ki[bkt + 1] : mi->n_kb;
		if (en - st <= max_occ && en > st) qs.n_anchor += en - st, qs.seeds.push_back((uint64_t)(uint32_t)s << 32 | (uint64_t)bkt);
	}
	std::sort(qs.seeds.begin(), qs.seeds.end());
}

static inline ChainParams prechain_params(const mpa_idx_s *mi, const mpa_mapopt_t &opt)
{
	const int32_t w = 1 << mi->opt.bbit;
	return ChainParams{ w, w, w, opt.max_chn_max_skip, opt.max_chn_iter, 2, 0, opt.chn_coef_log, !(opt.flag & MPA_MF_NO_SPLICE), mi->opt.kmer, mi->opt.bbit };
}
static inline bool prechain_enabled(const mpa_mapopt_t &opt) { return !(opt.flag & MPA_MF_NO_PRE_CHAIN) && !(opt.flag & MPA_MF_NO_SPLICE); }

// (2, host) anchors of the kept seeds (map.c:171-178), sorted; the pre-chain keeps those with a neighbour within one block
static void stage_anchors_host(mpa_batch_s *b, QueryState &qs, std::vector<uint64_t> &a)
{
	const mpa_idx_s *mi = b->mi;
	const mpa_mapopt_t &opt = b->opt;
	const int64_t n_bucket = (int64_t)mi->ki.size();
	static thread_local std::vector<uint64_t> u;
	a.clear();
	{
		AccTimer tm(0);
		a.reserve((size_t)qs.n_anchor);
		for (uint64_t kq : qs.seeds) {
			const int64_t bkt = (int64_t)(uint32_t)kq, st = mi->ki[bkt], en = bkt + 1 < n_bucket ? mi->ki[bkt + 1] : mi->n_kb;
			const uint64_t qpos = kq >> 32;
			for (int64_t j = st; j < en; ++j) a.push_back((uint64_t)mi->kb.at((size_t)j) << 32 | qpos);   // (at(): a mapped .mpi may leave kb misaligned)
		}
	}
	g_acc[14] += (int64_t)a.size() * 1000;
	{ AccTimer tm(1); sort_anchors_by_block(a); }
	if (prechain_enabled(opt)) {
		AccTimer tm(2);
		chain_anchors_set(prechain_params(mi, opt), a);          // (ascending: the sort of map.c:192 is implied)
		g_acc[15] += (int64_t)a.size() * 1000;
	}
}

// (2, GPU) the same from the sparse result of dev_prechain_forward(): extraction only
static void stage_anchors_from_device(mpa_batch_s *b, QueryState &qs, const PrechainSparse &ps, std::vector<uint64_t> &a)
{
	if (!ps.on_host.empty() && ps.on_host[(size_t)qs.qid]) { stage_anchors_host(b, qs, a); return; }   // (the device declined this query: degenerate input)
	AccTimer tm(2);
	const int64_t c0 = ps.cfirst[qs.qid], m = ps.cfirst[qs.qid + 1] - c0;
	a.clear();
	g_acc[14] += qs.n_anchor * 1000;
	if (m == 0) return;
	// (the device hands the predecessors over as indices into the query's part of the view)
	const ChainView v{ qs.n_anchor, m, ps.pos + c0, ps.f + c0, ps.pred + c0, ps.a + c0 };
	chain_extract_set(prechain_params(b->mi, b->opt), v, a);      // (ascending already: no sort)
	g_acc[15] += (int64_t)a.size() * 1000;
}

// (3) everything else up to and including the alignment plans (map.c:185-226)
static inline ChainParams main_chain_params(const mpa_idx_s *mi, const mpa_mapopt_t &opt)
{
	return ChainParams{ opt.max_intron, opt.max_gap, opt.bw, opt.max_chn_max_skip, opt.max_chn_iter, opt.min_chn_cnt, opt.min_chn_sc,
	                    opt.chn_coef_log, !(opt.flag & MPA_MF_NO_SPLICE), mi->opt.kmer, mi->opt.bbit };
}

static void stage_windows_from_chains(mpa_batch_s *b, QueryState &qs, const std::vector<uint64_t> &u, const std::vector<uint64_t> &a);
static void stage_chain_to_windows(mpa_batch_s *b, QueryState &qs, std::vector<uint64_t> &a)
{
	const mpa_idx_s *mi = b->mi;
	const mpa_mapopt_t &opt = b->opt;
	const int32_t is_splice = !(opt.flag & MPA_MF_NO_SPLICE);
	static thread_local std::vector<uint64_t> u;
	u.clear();
	const ChainParams cp = main_chain_params(mi, opt);
	(void)is_splice;
	{ AccTimer tm(3); chain_anchors(cp, a, u); }
	stage_windows_from_chains(b, qs, u, a);
}

// ... from the main chains (u: score << 32 | anchors per chain; a: their anchors, chain by chain): regions and their windows
static void stage_windows_from_chains(mpa_batch_s *b, QueryState &qs, const std::vector<uint64_t> &u, const std::vector<uint64_t> &a)
{
	const mpa_idx_s *mi = b->mi;
	const mpa_mapopt_t &opt = b->opt;
	std::vector<Region> &regs = qs.regs;
	AccTimer tm(4);
	regions_from_chains(mi, u, a, regs);
	sort_regions(regs);
	assign_parents(opt.mask_level, opt.mask_len, regs, mi->opt.kmer);
	select_secondary(opt.pri_ratio * opt.pri_ratio, mi->opt.kmer * 2, opt.best_n, regs);
	// refinement at base resolution (map.c:205-222): its windows
	extension_limits(nullptr, regs, &a, 100, opt.max_ext, qs.ext_refine);
}

// the window mp_refine_reg() re-seeds (map.c:36-41)
static inline void refine_window(const mpa_idx_s *mi, const Region &r, uint64_t ext, int64_t *as, int64_t *ae)
{
	const int64_t ctg_len = mi->ctg[r.vid >> 1].len;
	const int32_t extl = (int32_t)(ext >> 32), extr = (int32_t)ext;
	*as = r.vs > extl ? r.vs - extl : 0;
	*ae = r.ve + extr < ctg_len ? r.ve + extr : ctg_len;
}

// (4) refinement of every region (with the device's scan results if there are any), then the alignment plans
// rc != nullptr: the whole refinement ran on the device (dev_refine_chains): the chains of refinement window w = qs.win0 + region
static void stage_refine_to_plan(mpa_batch_s *b, QueryState &qs, const RefineHits *rh, const RefineChains *rc = nullptr)
{
	const mpa_idx_s *mi = b->mi;
	const mpa_mapopt_t &opt = b->opt;
	std::vector<Region> &regs = qs.regs;
	std::vector<uint64_t> &ext = qs.ext_refine;
	{
		AccTimer tm(5);
		std::vector<Region> kept;
		if (rc) {
			for (size_t i = 0; i < regs.size(); ++i) {
				const size_t w = (size_t)qs.win0 + i;
				g_acc[13] += 1000000LL;
				refine_region_from_chains(mi, opt, regs[i], (int32_t)(ext[i] >> 32), rc->U + rc->u_first[w], rc->u_first[w + 1] - rc->u_first[w], rc->A + rc->a_first[w]);
				if (regs[i].cnt > 0) kept.push_back(std::move(regs[i]));
			}
		} else {
			RefineQuery rq(qs.seq, qs.qlen, opt.kmer2, rh == nullptr);
			for (size_t i = 0; i < regs.size(); ++i) {
				if (rh) {
					const int64_t w = qs.win0 + (int64_t)i;
					refine_region(mi, opt, rq, regs[i], (int32_t)(ext[i] >> 32), (int32_t)ext[i], rh->hits.data() + rh->first[w], rh->first[w + 1] - rh->first[w]);
				} else refine_region(mi, opt, rq, regs[i], (int32_t)(ext[i] >> 32), (int32_t)ext[i]);
				if (regs[i].cnt > 0) kept.push_back(std::move(regs[i]));
			}
		}
		regs.swap(kept);
	}
	AccTimer tm6(6);
	sort_regions(regs);
	assign_parents(opt.mask_level, opt.mask_len, regs, mi->opt.kmer);
	select_secondary(opt.pri_ratio * opt.pri_ratio, mi->opt.kmer * 2, opt.best_n, regs);
	if (opt.flag & MPA_MF_NO_ALIGN) return;
	// alignment plans (map.c:223-227, align.c:239-267)
	extension_limits(mi, regs, nullptr, 100, opt.max_intron / 2, ext);
	for (size_t i = 0; i < regs.size(); ++i) {
		AlignPlan pl;
		if (plan_alignment(b, qs, (int32_t)i, (int32_t)(ext[i] >> 32), (int32_t)ext[i], pl)) qs.plans.push_back(std::move(pl));
	}
}

// the tail of mp_map() after mp_align() (map.c:228-237)
static void stage_finish(mpa_batch_s *b, QueryState &qs)
{
	const mpa_mapopt_t &opt = b->opt;
	if (opt.flag & MPA_MF_NO_ALIGN) return;
	std::vector<Region> kept;
	for (Region &r : qs.regs) if (r.aligned) kept.push_back(std::move(r));
	qs.regs.swap(kept);
	sort_regions(qs.regs);
	prefer_multi_exon(qs.regs, opt.io);
	assign_parents(opt.mask_level, opt.mask_len, qs.regs, b->mi->opt.kmer);
	select_secondary(opt.pri_ratio, b->mi->opt.kmer * 2, opt.best_n, qs.regs);
}

// ------------------------------------------------------------------------------------------------
// DP rounds
// ------------------------------------------------------------------------------------------------
static int32_t ungapped_score(const mpa_idx_s *mi, const mpa_mapopt_t &opt, const Region &r, int64_t nt_off, int32_t alen, const char *aa)
{
	std::vector<uint8_t> nt((size_t)alen * 3);
	fetch_nt(mi, (int32_t)r.vid, nt_off, nt_off + (int64_t)alen * 3, nt.data());
	const uint8_t *aa20 = tab_aa20();
	int32_t sc = 0;
	// NB: the reference's loop (align.c:36) advances its nucleotide index by 3 but bounds it by the AMINO-ACID
	// length, so only the first ceil(alen/3) codons contribute to AS:i.  Reproduced for byte-identical output.
	for (int32_t i = 0, j = 0; i < alen; i += 3, ++j) sc += opt.mat[codon_aa(&nt[i]) * opt.asize + aa20[(uint8_t)aa[j]]];
	return sc;
}

// one mp_align_seq() call: either the ungapped shortcut (align.c:65-67) or a DP task with traceback
static void make_segment(mpa_batch_s *b, const QueryState &qs, const Region &r, const AlignPlan &pl, int32_t ne0, int32_t ne1, int32_t ae0, int32_t ae1,
                         int32_t qi, int32_t pi, std::vector<mpa_dp_task_t> &tasks, Segment &s)
{
	const mpa_mapopt_t &opt = b->opt;
	s.ne0 = ne0, s.ne1 = ne1, s.ae0 = ae0, s.ae1 = ae1;
	const int32_t nlen = ne1 - ne0, alen = ae1 - ae0;
	if (nlen == alen * 3 && alen <= opt.kmer2) {
		s.task = -1;
		s.score = ungapped_score(b->mi, opt, r, pl.vs0 + ne0, alen, qs.seq + ae0);
		s.cigar.assign(1, (uint32_t)alen << 4);
	} else {
		s.task = (int32_t)tasks.size();
		tasks.push_back(make_task(qs, r, pl.vs0 + ne0, nlen, ae0, alen, MPA_F_CIGAR, opt.io));
	}
}

static void store_result(Segment &s, const mpa_dp_rst_t *rst, const uint32_t *pool)
{
	if (s.task < 0) return;
	const mpa_dp_rst_t &o = rst[s.task];
	s.score = o.score;
	s.cigar.assign(pool + o.cigar_off, pool + o.cigar_off + o.n_cigar);
}

// round-1 tasks of one query: both extensions of every region plus the gaps between kept anchors
// (align.c:305-313, all iterations but the first).  Runs inside the threaded stage A.
// The reference repeats an extension that did not reach the end of the protein with the cheaper intron penalty of a terminal exon
// (io_end) on a window of at most max_ext rows (align.c:290-296, 324-331).  Whether it will is known only after the first call, but
// WHAT it would compute is not: window, protein slice and penalty depend on the plan alone.  So the repeat is issued speculatively
// next to the first call -- at most max_ext (1 000) rows against windows of max_intron / 2, about 4 % more extension rows at
// config 3 -- and a batch needs two DP rounds instead of three: one launch, one host round trip and one tail fewer.
static void plan_round1(mpa_batch_s *b, QueryState &qs)
{
	const mpa_mapopt_t &opt = b->opt;
	const bool retry = opt.io > opt.io_end;
	for (size_t pi = 0; pi < qs.plans.size(); ++pi) {
		AlignPlan &pl = qs.plans[pi];
		const Region &r = qs.regs[pl.reg];
		pl.t_left = (int32_t)qs.local1.size();
		qs.local1.push_back(make_task(qs, r, pl.as, pl.vs1 - pl.as, 0, pl.as1, MPA_F_EXT_LEFT | MPA_F_SS_SKIP0, opt.io));   // starts at the window's first position
		if (retry) {                                          // 5'-end exon: the same with a cheaper intron, on the window's last max_ext rows
			const int64_t as_alt = pl.vs1 - pl.as > opt.max_ext ? pl.vs1 - opt.max_ext : pl.as;
			pl.t_left2 = (int32_t)qs.local1.size();
			qs.local1.push_back(make_task(qs, r, as_alt, pl.vs1 - as_alt, 0, pl.as1, MPA_F_EXT_LEFT | (as_alt == pl.as ? MPA_F_SS_SKIP0 : 0), opt.io_end));
		}
		if (pl.has_right) {
			pl.t_right = (int32_t)qs.local1.size();
			qs.local1.push_back(make_task(qs, r, pl.mid_ve, pl.ae - pl.mid_ve, pl.mid_qe, qs.qlen - pl.mid_qe, MPA_F_EXT_RIGHT, opt.io));
			if (retry) {                                      // 3'-end exon
				const int64_t l_ext = std::min<int64_t>(pl.ae - pl.mid_ve, opt.max_ext);
				pl.t_right2 = (int32_t)qs.local1.size();
				qs.local1.push_back(make_task(qs, r, pl.mid_ve, l_ext, pl.mid_qe, qs.qlen - pl.mid_qe, MPA_F_EXT_RIGHT, opt.io_end));
			}
		}
		int32_t ne0 = (int32_t)(pl.vs1 - pl.vs0), ae0 = pl.as1;
		for (int32_t i = pl.i0 + 1; i < r.cnt; ++i) {
			if (!(r.a[i] >> 31 & 1)) continue;
			const int32_t ne1 = (int32_t)(r.a[i] >> 32) + 1, ae1 = (int32_t)((uint32_t)r.a[i] << 1 >> 1) + 1;
			pl.gaps.emplace_back();
			make_segment(b, qs, r, pl, ne0, ne1, ae0, ae1, qs.qid, (int32_t)pi, qs.local1, pl.gaps.back());
			ne0 = ne1, ae0 = ae1;
		}
	}
}

static void emit_round1(mpa_batch_s *b)
{
	int64_t n = 0;
	for (QueryState &qs : b->qs) qs.base1 = n, n += (int64_t)qs.local1.size();
	b->tasks.resize((size_t)n);
	parallel_for(b->n_threads, (int64_t)b->qs.size(), [&](int64_t qi) {
		QueryState &qs = b->qs[qi];
		if (!qs.local1.empty()) memcpy(&b->tasks[qs.base1], qs.local1.data(), qs.local1.size() * sizeof(mpa_dp_task_t));
	});
}

// round 1 in (rst == nullptr: there was nothing to compute), round 2 out: the spans the two extensions accepted, re-aligned with
// traceback (the first and the last mp_align_seq() call of align.c:297-300, 331)
static void take_round1_emit_round2(mpa_batch_s *b, const mpa_dp_rst_t *rst, const uint32_t *pool)
{
	const mpa_mapopt_t &opt = b->opt;
	if (rst) parallel_for(b->n_threads, (int64_t)b->qs.size(), [&](int64_t qi) {
		QueryState &qs = b->qs[qi];
		for (size_t pi = 0; pi < qs.plans.size(); ++pi) {
			AlignPlan &pl = qs.plans[pi];
			const mpa_dp_rst_t *mine = rst + qs.base1;                 // this query's slice of the round-1 results
			for (Segment &g : pl.gaps) store_result(g, mine, pool);
			pl.l_nt = mine[pl.t_left].nt_len, pl.l_aa = mine[pl.t_left].aa_len;
			// the terminal-exon repeat counts iff the reference would have made it AND it reaches the end of the protein (align.c:290-296)
			if (pl.t_left2 >= 0 && pl.l_aa != pl.as1 && pl.l_nt < opt.max_ext && mine[pl.t_left2].aa_len == pl.as1)
				pl.l_nt = mine[pl.t_left2].nt_len, pl.l_aa = mine[pl.t_left2].aa_len;
			if (pl.has_right) {
				pl.r_nt = mine[pl.t_right].nt_len, pl.r_aa = mine[pl.t_right].aa_len;
				if (pl.t_right2 >= 0 && pl.r_aa < qs.qlen - pl.mid_qe && pl.r_nt < opt.max_ext && mine[pl.t_right2].aa_len == qs.qlen - pl.mid_qe)
					pl.r_nt = mine[pl.t_right2].nt_len, pl.r_aa = mine[pl.t_right2].aa_len;
			}
		}
	});
	std::vector<mpa_dp_task_t> next;
	for (size_t qi = 0; qi < b->qs.size(); ++qi) {
		QueryState &qs = b->qs[qi];
		for (size_t pi = 0; pi < qs.plans.size(); ++pi) {
			AlignPlan &pl = qs.plans[pi];
			Region &r = qs.regs[pl.reg];
			// region start after the left extension (align.c:297-300); its span is the first mp_align_seq() call
			r.vs = pl.vs1 - pl.l_nt;
			r.qs = pl.as1 - pl.l_aa;
			pl.has_left_span = true;
			make_segment(b, qs, r, pl, (int32_t)(r.vs - pl.vs0), (int32_t)(pl.vs1 - pl.vs0), r.qs, pl.as1, (int32_t)qi, (int32_t)pi, next, pl.left_span);
			if (pl.has_right && pl.r_nt > 0 && pl.r_aa > 0) {             // the accepted right-extension span (align.c:331)
				const int32_t ne0 = (int32_t)(pl.mid_ve - pl.vs0);
				pl.has_right_span = true;
				make_segment(b, qs, r, pl, ne0, ne0 + pl.r_nt, pl.mid_qe, pl.mid_qe + pl.r_aa, (int32_t)qi, (int32_t)pi, next, pl.right_span);
			}
		}
	}
	b->tasks.swap(next);
}

static void take_round3(mpa_batch_s *b, const mpa_dp_rst_t *rst, const uint32_t *pool)
{
	parallel_for(b->n_threads, (int64_t)b->qs.size(), [&](int64_t qi) {
		QueryState &qs = b->qs[qi];
		for (AlignPlan &pl : qs.plans) {
			Region &r = qs.regs[pl.reg];
			int32_t score = 0;
			r.cigar.clear();
			if (pl.has_left_span) store_result(pl.left_span, rst, pool);
			if (pl.has_right_span) store_result(pl.right_span, rst, pool);
			auto add = [&](const Segment &s) {
				for (uint32_t c : s.cigar) append_cigar(r.cigar, c & 0xf, (int32_t)(c >> 4));
				score += s.score;
			};
			if (pl.has_left_span) add(pl.left_span);
			for (const Segment &g : pl.gaps) add(g);
			if (pl.has_right_span) add(pl.right_span);
			r.ve = pl.mid_ve, r.qe = pl.mid_qe;
			if (pl.has_right && pl.r_nt > 0 && pl.r_aa > 0) r.ve += pl.r_nt, r.qe += pl.r_aa;
			r.aligned = true;
			r.dp_score = score, r.dp_max2 = 0;
			r.dist_stop = dist_to_stop(b->mi, r, pl.ae);
			r.dist_start = dist_to_start(b->mi, r, pl.as, pl.ae);
			summarize_alignment(b->mi, b->opt, r, qs.seq, qs.qlen, pl.ae);
		}
		stage_finish(b, qs);
	});
}

} // namespace mpa

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" {

// MPA_GPU_SEED: 0 = seeding always on the host, 1 = always on the device, unset = by the size of the batch
static thread_local int tl_seed_mode = -2;          // (the mpa_dbg_* test hooks force a mode for their own call, on their own thread)
static int gpu_seeding_mode()
{
	if (tl_seed_mode != -2) return tl_seed_mode;
	const char *e = getenv("MPA_GPU_SEED");             // (read per call: the tests flip it)
	return e ? (atoi(e) != 0 ? 1 : 0) : -1;
}
struct SeedModeScope { int keep; explicit SeedModeScope(int m) : keep(tl_seed_mode) { tl_seed_mode = m; } ~SeedModeScope() { tl_seed_mode = keep; } };
// Below this many anchors per mini-batch the host's own sort + forward pass is faster than a round trip to a GPU that is
// busy with DP rounds (measured: 2.7 M anchors at config 2 -> host, 25 M at 600 Mbp and 240 M at config 3 -> device).
static const int64_t kDeviceSeedingMinAnchors = 8000000;

// Phase 1a of a batch, host only: the seeds of every query (sketch, bucket lookup, occurrence cut-off: map.c:126-177 up to the
// anchor loop) and, if the batch is worth seeding on the device, its seed jobs.  In the stream pipeline this is a stage of its own.
static mpa_batch_t *batch_sketch_phase(bool have_device, const mpa_idx_t *mi, const mpa_mapopt_t *opt, const mpa_qbatch_t *q, int n_threads)
{
	if (mi->kb.empty() && mi->n_kb == 0) { set_error("the index has no k-mer table (genome-only index): cannot map"); return nullptr; }
	mpa_batch_s *b = new mpa_batch_s();
	b->mi = mi, b->opt = *opt, b->q = *q, b->n_threads = n_threads > 1 ? n_threads : 1;
	mpa_dpopt_from_mapopt(opt, &b->dpopt);
	b->qs.resize(q->n_seq);
	for (int32_t i = 0; i < q->n_seq; ++i) {
		b->qs[i].qid = i, b->qs[i].seq = q->seqs + q->q_off[i], b->qs[i].qlen = (int32_t)(q->q_off[i + 1] - q->q_off[i]);
	}
	const double t0 = now_ms();
	parallel_for(b->n_threads, q->n_seq, [&](int64_t i) { stage_seeds(b, b->qs[i]); });
	timing_note("  A1: seeds of all queries", now_ms() - t0);
	const int mode = gpu_seeding_mode();
	if (!have_device || mode == 0 || !prechain_enabled(*opt) || mi->opt.bbit <= 0 || q->n_seq == 0) return b;
	std::vector<int64_t> &qfirst = b->seed_qfirst;
	std::vector<size_t> jfirst((size_t)q->n_seq + 1, 0);
	qfirst.assign((size_t)q->n_seq + 1, 0);
	for (int32_t i = 0; i < q->n_seq; ++i) qfirst[i + 1] = qfirst[i] + b->qs[i].n_anchor, jfirst[i + 1] = jfirst[i] + b->qs[i].seeds.size();
	if ((mode < 0 && qfirst[q->n_seq] < kDeviceSeedingMinAnchors) || jfirst[q->n_seq] == 0) { qfirst.clear(); return b; }
	b->seed_jobs.resize(jfirst[q->n_seq]);
	const int64_t n_bucket = (int64_t)mi->ki.size();
	SeedJob *const jobs_p = b->seed_jobs.data();
	const int64_t *const qfirst_p = qfirst.data();
	const size_t *const jfirst_p = jfirst.data();
	parallel_for(b->n_threads, q->n_seq, [&, jobs_p, qfirst_p, jfirst_p](int64_t i) {
		int64_t dst = qfirst_p[i];
		size_t j = jfirst_p[i];
		for (uint64_t kq : b->qs[i].seeds) {
			const int64_t bkt = (int64_t)(uint32_t)kq, st = mi->ki[bkt], en = bkt + 1 < n_bucket ? mi->ki[bkt + 1] : mi->n_kb;
			jobs_p[j++] = SeedJob{ st, dst, (int32_t)(en - st), (int32_t)(kq >> 32), (int32_t)i };
			dst += en - st;
		}
	});
	timing_note("  A1: seed jobs", now_ms() - t0);
	return b;
}

// Phase 1b: the anchors of the seed jobs, the pre-chain and the main chain on the device (seed_exec.hip).  false = error.
static bool batch_device_seed_phase(mpa_ctx_t *seed_ctx, mpa_batch_s *b, bool want_chains, SeedHold *hold)
{
	if (!seed_ctx || b->seed_jobs.empty()) return true;
	const double t1 = now_ms();
	const mpa_idx_t *mi = b->mi;
	// (with the main chain's parameters the device carries on through both chaining rounds, unless the caller only wants the pre-chain)
	const ChainParams main_cp = main_chain_params(mi, b->opt);
	const int rc = dev_prechain_forward(seed_ctx, const_cast<mpa_idx_s*>(mi), prechain_params(mi, b->opt), b->q.n_seq, b->seed_qfirst.data(), b->seed_jobs.data(),
	                                    (int64_t)b->seed_jobs.size(), b->sparse, want_chains ? &main_cp : nullptr, hold);
	std::vector<SeedJob>().swap(b->seed_jobs);
	std::vector<int64_t>().swap(b->seed_qfirst);
	if (rc == MPA_ERR_UNSUPPORTED) {                    // e.g. the batch does not fit the device: seed on the host
		if (timing_on()) fprintf(stderr, "[mpa-timing]   device seeding declined (%s): seeding on the host\n", mpa_last_error());
		return true;
	}
	if (rc != MPA_OK) return false;
	b->seeded_on_device = true;
	timing_note("  seeding on the GPU (sift + pre-chain + both chaining rounds)", now_ms() - t1);
	return true;
}

// Phase 1 of a batch: seeds of every query; with a device context and a batch that is worth it, also the anchors, the pre-chain
// and the main chain on the device.
static mpa_batch_t *batch_seed_phase(mpa_ctx_t *seed_ctx, const mpa_idx_t *mi, const mpa_mapopt_t *opt, const mpa_qbatch_t *q, int n_threads, bool want_chains = true, SeedHold *hold = nullptr)
{
	mpa_batch_t *b = batch_sketch_phase(seed_ctx != nullptr, mi, opt, q, n_threads);
	if (b && !batch_device_seed_phase(seed_ctx, b, want_chains, hold)) { delete b; return nullptr; }
	return b;
}

// Phase 2: pre-chain (host: anchors, sort, forward pass, extraction; after device seeding: extraction only), chaining,
// regions, refinement, alignment plans.  b->sparse (pinned memory of the seeding context) is not needed afterwards.
static int gpu_refine_mode()
{
	const char *e = getenv("MPA_GPU_REFINE");           // 0 / 1 / unset = by the size of the batch
	return e ? (atoi(e) != 0 ? 1 : 0) : -1;
}
static const int64_t kDeviceRefineMinBases = 100000000;     // below this the host's own scan is as fast as the round trip

// the packed k-mer words of a query (RefineQuery::words without the bitmap)
static void query_words(const char *aa, int32_t l_aa, int32_t k, std::vector<uint32_t> &w)
{
	const uint8_t *aa13 = tab_aa13();
	const uint32_t mask = (1U << 4 * k) - 1;
	uint32_t x = 0;
	w.clear();
	for (int32_t i = 0, run = 0; i < l_aa; ++i) {
		const uint32_t c = aa13[(uint8_t)aa[i]];
		if (c >= 14) { run = 0, x = 0; continue; }
		x = (x << 4 | c) & mask;
		if (++run >= k) w.push_back(x);
	}
}

// The query side of the device refinement: the query's k-mers (packed word, index of the last residue) grouped by word -- in
// ascending hash order, positions ascending inside a group, as the pairing of map.c:53-79 meets them in the sorted list.
struct QueryGroups { std::vector<uint32_t> gword, gcount, qpos; };
static void query_groups(const char *aa, int32_t l_aa, int32_t k, QueryGroups &out)
{
	const uint8_t *aa13 = tab_aa13();
	const uint32_t mask = (1U << 4 * k) - 1;
	static thread_local std::vector<uint64_t> hk;           // hash << 32 | position
	static thread_local std::vector<uint32_t> wd;           // word of the k-mer ending at position i (by position)
	hk.clear(), wd.assign((size_t)l_aa, 0);
	uint32_t x = 0;
	for (int32_t i = 0, run = 0; i < l_aa; ++i) {
		const uint32_t c = aa13[(uint8_t)aa[i]];
		if (c >= 14) { run = 0, x = 0; continue; }
		x = (x << 4 | c) & mask;
		if (++run >= k) hk.push_back((uint64_t)hash32_mask(x, mask) << 32 | (uint32_t)i), wd[(size_t)i] = x;
	}
	sort_u64(hk.data(), hk.data() + hk.size());
	out.gword.clear(), out.gcount.clear(), out.qpos.clear();
	for (size_t i = 0; i < hk.size(); ++i) {
		if (i == 0 || hk[i] >> 32 != hk[i - 1] >> 32) out.gword.push_back(wd[(size_t)(uint32_t)hk[i]]), out.gcount.push_back(0);
		++out.gcount.back();
		out.qpos.push_back((uint32_t)hk[i]);
	}
}

static void batch_plan_phase(mpa_batch_s *b, mpa_ctx_t *rctx)
{
	const double t0 = now_ms();
	const int64_t n_q = (int64_t)b->qs.size();
	parallel_for(b->n_threads, n_q, [&](int64_t i) {
		static thread_local std::vector<uint64_t> a, u;
		const PrechainSparse &ps = b->sparse;
		if (b->seeded_on_device && ps.has_chains && !(!ps.on_host.empty() && ps.on_host[(size_t)i])) {
			// both chaining rounds ran on the device: its chains are what mp_chain() returns for this query (map.c:195)
			g_acc[14] += b->qs[i].n_anchor * 1000;
			u.assign(ps.U + ps.u_first[(size_t)i], ps.U + ps.u_first[(size_t)i + 1]);
			a.assign(ps.A + ps.a_first[(size_t)i], ps.A + ps.a_first[(size_t)i + 1]);
			stage_windows_from_chains(b, b->qs[i], u, a);
			return;
		}
		if (b->seeded_on_device && !ps.has_chains) stage_anchors_from_device(b, b->qs[i], ps, a);
		else stage_anchors_host(b, b->qs[i], a);
		stage_chain_to_windows(b, b->qs[i], a);
	});
	b->sparse = PrechainSparse();
	timing_note("  plan: anchors..windows (wall)", now_ms() - t0);
	const double t_a = now_ms();
	// the refinement scan of all windows on the device, if the batch is worth the round trip
	RefineHits rh;
	RefineChains rchains;
	bool on_device = false, chains_on_device = false;
	const int mode = gpu_refine_mode();
	if (rctx && mode != 0 && b->opt.kmer2 <= 6) {
		static thread_local std::vector<RefineWindow> wins;
		static thread_local std::vector<int64_t> qw_first;
		static thread_local std::vector<uint32_t> qwords;
		wins.clear();
		int64_t n_bases = 0;
		for (int64_t i = 0; i < n_q; ++i) {
			QueryState &qs = b->qs[i];
			qs.win0 = (int64_t)wins.size();
			for (size_t r = 0; r < qs.regs.size(); ++r) {
				int64_t as, ae;
				refine_window(b->mi, qs.regs[r], qs.ext_refine[r], &as, &ae);
				wins.push_back(RefineWindow{ as, (int32_t)i, (int32_t)qs.regs[r].vid, (int32_t)(ae - as) });
				n_bases += ae - as;
			}
		}
		if (!wins.empty() && (mode == 1 || n_bases >= kDeviceRefineMinBases)) {
			// the whole refinement on the device: groups of every query (in parallel), then scan + pairing + sort + chains
			static thread_local std::vector<QueryGroups> per;
			static thread_local RefineGroupsHost G;
			per.resize((size_t)n_q);
			QueryGroups *perp = per.data();
			parallel_for(b->n_threads, n_q, [&, perp](int64_t i) { query_groups(b->qs[i].seq, b->qs[i].qlen, b->opt.kmer2, perp[i]); });
			G.qg_first.assign((size_t)n_q + 1, 0);
			size_t n_kmer = 0;
			for (int64_t i = 0; i < n_q; ++i) G.qg_first[(size_t)i + 1] = G.qg_first[(size_t)i] + (int64_t)per[(size_t)i].gword.size(), n_kmer += per[(size_t)i].qpos.size();
			G.gword.resize((size_t)G.qg_first[(size_t)n_q]), G.gcount.resize(G.gword.size()), G.gfirst.resize(G.gword.size()), G.qpos.resize(n_kmer);
			{
				size_t at = 0;
				for (int64_t i = 0; i < n_q; ++i) {
					const QueryGroups &p = per[(size_t)i];
					const size_t g0 = (size_t)G.qg_first[(size_t)i];
					size_t k0 = at;
					for (size_t g = 0; g < p.gword.size(); ++g) G.gword[g0 + g] = p.gword[g], G.gcount[g0 + g] = p.gcount[g], G.gfirst[g0 + g] = (uint32_t)k0, k0 += p.gcount[g];
					if (!p.qpos.empty()) memcpy(&G.qpos[at], p.qpos.data(), p.qpos.size() * 4);
					at += p.qpos.size();
				}
			}
			const double t1 = now_ms();
			const int rc = dev_refine_chains(rctx, const_cast<mpa_idx_s*>(b->mi), b->opt.kmer2, b->mi->opt.min_aa_len, b->opt.max_ava, refine_chain_params(b->opt), (int32_t)n_q, G,
			                                 (int64_t)wins.size(), wins.data(), rchains);
			timing_note("  refinement on the GPU (scan + pairs + chains)", now_ms() - t1);
			chains_on_device = rc == MPA_OK;                   // anything else: the scan alone below, or the host
		}
		if (!chains_on_device && !wins.empty() && (mode == 1 || n_bases >= kDeviceRefineMinBases)) {
			std::vector<uint32_t> w;
			qw_first.assign((size_t)n_q + 1, 0), qwords.clear();
			for (int64_t i = 0; i < n_q; ++i) {
				query_words(b->qs[i].seq, b->qs[i].qlen, b->opt.kmer2, w);
				qwords.insert(qwords.end(), w.begin(), w.end());
				qw_first[i + 1] = (int64_t)qwords.size();
			}
			const double t1 = now_ms();
			const int rc = dev_refine_scan(rctx, const_cast<mpa_idx_s*>(b->mi), b->opt.kmer2, b->mi->opt.min_aa_len, (int32_t)n_q, qw_first.data(), qwords.data(),
			                               (int64_t)wins.size(), wins.data(), rh);
			timing_note("  refinement scan on the GPU", now_ms() - t1);
			on_device = rc == MPA_OK;                         // anything else: scan on the host
		}
	}
	const RefineHits *rhp = on_device ? &rh : nullptr;
	timing_note("  plan: windows + scan (wall)", now_ms() - t_a);
	const double t_b = now_ms();
	const RefineChains *rcp = chains_on_device ? &rchains : nullptr;
	parallel_for(b->n_threads, n_q, [&, rhp, rcp](int64_t i) {
		stage_refine_to_plan(b, b->qs[i], rhp, rcp);
		AccTimer tm(7);
		plan_round1(b, b->qs[i]);
	});
	timing_note("  plan: refine..plans (wall)", now_ms() - t_b);
	timing_note("stage A (seed..plan)", now_ms() - t0);
	if (timing_on()) for (int k = 0; k < 16; ++k) { timing_note(kAccName[k], (double)g_acc[k].exchange(0) / 1e6); }   // thread CPU ms summed over the workers (13-15: counts)
	if (b->opt.flag & MPA_MF_NO_ALIGN) { b->round = 4; return; }
	// the round-1 task list is part of planning too: in a stream it is ready before a DP lane picks the batch up
	const double t1 = now_ms();
	emit_round1(b);
	timing_note("emit round 1", now_ms() - t1);
	b->round = 1;
	if (b->tasks.empty()) { take_round3(b, nullptr, nullptr); b->round = 4; }   // nothing to align at all
}

static mpa_batch_t *batch_begin_impl(mpa_ctx_t *seed_ctx, const mpa_idx_t *mi, const mpa_mapopt_t *opt, const mpa_qbatch_t *q, int n_threads)
{
	mpa_batch_t *b = batch_seed_phase(seed_ctx, mi, opt, q, n_threads);
	if (b) batch_plan_phase(b, seed_ctx);
	return b;
}

// Test hook (not in include/mpamd.h): the refinement scan of n_win windows [as, as + len) on vid for ONE query, on the host
// (ctx == NULL: refine_seed_pairs' scan) or on the device (k_refine_scan).  first[n_win + 1] / *out (mpa_free): the hits
// (hash << 32 | window position) of every window, sorted.
int64_t mpa_dbg_refine_hits(mpa_ctx_t *ctx, const mpa_idx_t *mi, int32_t kmer, const char *aa, int32_t l_aa, int32_t n_win, const int32_t *vid, const int64_t *as,
                            const int32_t *len, int64_t *first, uint64_t **out)
{
	*out = nullptr;
	std::vector<std::vector<uint64_t>> per((size_t)n_win);
	if (ctx) {
		std::vector<RefineWindow> wins((size_t)n_win);
		for (int32_t k = 0; k < n_win; ++k) wins[k] = RefineWindow{ as[k], 0, vid[k], len[k] };
		std::vector<uint32_t> w;
		query_words(aa, l_aa, kmer, w);
		const int64_t qw_first[2] = { 0, (int64_t)w.size() };
		RefineHits rh;
		const int rc = dev_refine_scan(ctx, const_cast<mpa_idx_s*>(mi), kmer, mi->opt.min_aa_len, 1, qw_first, w.data(), n_win, wins.data(), rh);
		if (rc != MPA_OK) return rc;
		for (int32_t k = 0; k < n_win; ++k) per[k].assign(rh.hits.begin() + rh.first[k], rh.hits.begin() + rh.first[k + 1]);
	} else {
		RefineQuery rq(aa, l_aa, kmer);
		std::vector<uint8_t> nt;
		std::vector<uint64_t> a;
		for (int32_t k = 0; k < n_win; ++k) {
			nt.resize((size_t)std::max(len[k], 1));
			fetch_nt(mi, vid[k], as[k], as[k] + len[k], nt.data());
			refine_seed_pairs(nt.data(), len[k], mi->opt.min_aa_len, kmer, rq.qk, rq.filter->data(), INT32_MAX, per[k], a);
		}
	}
	first[0] = 0;
	for (int32_t k = 0; k < n_win; ++k) std::sort(per[k].begin(), per[k].end()), first[k + 1] = first[k] + (int64_t)per[k].size();
	uint64_t *o = (uint64_t*)malloc((size_t)std::max<int64_t>(first[n_win], 1) * 8);
	for (int32_t k = 0; k < n_win; ++k) memcpy(o + first[k], per[k].data(), per[k].size() * 8);
	*out = o;
	return first[n_win];
}

// Test hook (not in include/mpamd.h): the anchors that survive the pre-chain (map.c:163-192), query by query, computed on the
// host (ctx == NULL) or with the device seeding stage (seed_exec.hip).  off[n_seq + 1] receives the offsets into *out
// (malloc'd, mpa_free).  Returns the total or a negative error code.
// Test hook: the anchors of every query as map.c:163-178 leaves them (block << 32 | query position, sorted) -- the INPUT of the
// reference's pre-chain call (map.c:188), so that a test can hand them to the real mp_chain().  off [n_seq + 1]; *out malloc'd.
int64_t mpa_dbg_anchors(const mpa_idx_t *mi, const mpa_mapopt_t *opt, const mpa_qbatch_t *q, int n_threads, int64_t *off, uint64_t **out)
{
	*out = nullptr;
	mpa_mapopt_t o2 = *opt;
	o2.flag |= MPA_MF_NO_PRE_CHAIN;                       // (stage_anchors_host then stops after the sort)
	mpa_batch_t *b;
	{ SeedModeScope host_only(0); b = batch_seed_phase(nullptr, mi, &o2, q, n_threads, false); }
	if (!b) return MPA_ERR_ARG;
	std::vector<std::vector<uint64_t>> per((size_t)q->n_seq);
	parallel_for(b->n_threads, q->n_seq, [&](int64_t i) { stage_anchors_host(b, b->qs[i], per[i]); });
	off[0] = 0;
	for (int32_t i = 0; i < q->n_seq; ++i) off[i + 1] = off[i] + (int64_t)per[i].size();
	uint64_t *o = (uint64_t*)malloc((size_t)std::max<int64_t>(off[q->n_seq], 1) * 8);
	for (int32_t i = 0; i < q->n_seq; ++i) if (!per[i].empty()) memcpy(o + off[i], per[i].data(), per[i].size() * 8);
	*out = o;
	const int64_t n = off[q->n_seq];
	delete b;
	return n;
}

int64_t mpa_dbg_prechain_survivors(mpa_ctx_t *ctx, const mpa_idx_t *mi, const mpa_mapopt_t *opt, const mpa_qbatch_t *q, int n_threads, int64_t *off, uint64_t **out)
{
	*out = nullptr;
	mpa_batch_t *b;
	{ SeedModeScope forced(ctx ? 1 : 0); b = batch_seed_phase(ctx, mi, opt, q, n_threads, false); }
	if (!b) return MPA_ERR_ARG;
	if (ctx && !b->seeded_on_device && q->n_seq > 0 && prechain_enabled(*opt)) { delete b; set_error("device seeding was not used"); return MPA_ERR_UNSUPPORTED; }
	std::vector<std::vector<uint64_t>> per((size_t)q->n_seq);
	parallel_for(b->n_threads, q->n_seq, [&](int64_t i) {
		if (b->seeded_on_device) stage_anchors_from_device(b, b->qs[i], b->sparse, per[i]);
		else stage_anchors_host(b, b->qs[i], per[i]);
	});
	off[0] = 0;
	for (int32_t i = 0; i < q->n_seq; ++i) off[i + 1] = off[i] + (int64_t)per[i].size();
	uint64_t *o = (uint64_t*)malloc((size_t)std::max<int64_t>(off[q->n_seq], 1) * 8);
	for (int32_t i = 0; i < q->n_seq; ++i) memcpy(o + off[i], per[i].data(), per[i].size() * 8);
	*out = o;
	const int64_t n = off[q->n_seq];
	delete b;
	return n;
}

// Test hook (not in include/mpamd.h): the MAIN chains of every query (what mp_chain() returns at map.c:195: u = score << 32 |
// anchors per chain, a = the chains' anchors), from the host stages (ctx == NULL) or from the device, which runs both chaining
// rounds (k_chain_extract, k_chain_fwd).  off_u / off_a [n_seq + 1]; *out_u / *out_a malloc'd (mpa_free).  Returns the number of
// queries the device handed back to the host (>= 0) or a negative error code.
int64_t mpa_dbg_main_chains(mpa_ctx_t *ctx, const mpa_idx_t *mi, const mpa_mapopt_t *opt, const mpa_qbatch_t *q, int n_threads, int64_t *off_u, uint64_t **out_u,
                            int64_t *off_a, uint64_t **out_a)
{
	*out_u = *out_a = nullptr;
	mpa_batch_t *b;
	{ SeedModeScope forced(ctx ? 1 : 0); b = batch_seed_phase(ctx, mi, opt, q, n_threads, true); }
	if (!b) return MPA_ERR_ARG;
	if (ctx && q->n_seq > 0 && prechain_enabled(*opt) && !(b->seeded_on_device && b->sparse.has_chains)) { delete b; set_error("the device did not chain"); return MPA_ERR_UNSUPPORTED; }
	std::vector<std::vector<uint64_t>> us((size_t)q->n_seq), as((size_t)q->n_seq);
	const ChainParams cp = main_chain_params(mi, *opt);
	int64_t n_back = 0;
	for (int32_t i = 0; i < q->n_seq; ++i) {
		const PrechainSparse &ps = b->sparse;
		if (b->seeded_on_device && ps.has_chains && !(!ps.on_host.empty() && ps.on_host[(size_t)i])) {
			us[i].assign(ps.U + ps.u_first[(size_t)i], ps.U + ps.u_first[(size_t)i + 1]);
			as[i].assign(ps.A + ps.a_first[(size_t)i], ps.A + ps.a_first[(size_t)i + 1]);
			continue;
		}
		n_back += b->seeded_on_device ? 1 : 0;
		stage_anchors_host(b, b->qs[i], as[i]);
		chain_anchors(cp, as[i], us[i]);
	}
	off_u[0] = off_a[0] = 0;
	for (int32_t i = 0; i < q->n_seq; ++i) off_u[i + 1] = off_u[i] + (int64_t)us[i].size(), off_a[i + 1] = off_a[i] + (int64_t)as[i].size();
	uint64_t *ou = (uint64_t*)malloc((size_t)std::max<int64_t>(off_u[q->n_seq], 1) * 8), *oa = (uint64_t*)malloc((size_t)std::max<int64_t>(off_a[q->n_seq], 1) * 8);
	for (int32_t i = 0; i < q->n_seq; ++i) {
		if (!us[i].empty()) memcpy(ou + off_u[i], us[i].data(), us[i].size() * 8);
		if (!as[i].empty()) memcpy(oa + off_a[i], as[i].data(), as[i].size() * 8);
	}
	*out_u = ou, *out_a = oa;
	delete b;
	return n_back;
}

// Test hook: the forward pass of mp_chain (chain.c:181-209) for n_prob chaining problems whose sorted anchors lie back to back
// in a[] (first[n_prob + 1] = boundaries): f / pred (index inside the problem, -1 = none) of every anchor, from the device
// kernel k_chain_fwd (ctx != NULL) or from the host pass that chain_anchors() runs (ctx == NULL).
int mpa_dbg_chain_forward(mpa_ctx_t *ctx, int32_t max_dist_x, int32_t max_dist_y, int32_t bw, int32_t max_skip, int32_t max_iter, float coef_log,
                          int32_t is_spliced, int32_t kmer, int32_t bbit, int32_t n_prob, const int64_t *first, const uint64_t *a, int32_t *f, int32_t *pred)
{
	const ChainParams p{ max_dist_x, max_dist_y, bw, max_skip, max_iter, 1, 0, coef_log, is_spliced, kmer, bbit };
	const int64_t n = first[n_prob];
	if (!ctx) {
		for (int32_t q = 0; q < n_prob; ++q) chain_forward(p, a + first[q], first[q + 1] - first[q], f + first[q], pred + first[q]);
		return MPA_OK;
	}
	ChainIO io;
	int rc = dev_chain_buffers(ctx, n, io);
	if (rc != MPA_OK) return rc;
	if (n > 0) memcpy(io.a, a, (size_t)n * 8);
	if ((rc = dev_chain_forward(ctx, p, n_prob, first, io)) != MPA_OK) return rc;
	if (n > 0) memcpy(f, io.f, (size_t)n * 4), memcpy(pred, io.pred, (size_t)n * 4);
	return MPA_OK;
}

static mpa_batch_t *mpa_batch_begin_impl(const mpa_idx_t *mi, const mpa_mapopt_t *opt, const mpa_qbatch_t *q, int n_threads)
{
	return batch_begin_impl(nullptr, mi, opt, q, n_threads);
}

mpa_batch_t *mpa_batch_begin(const mpa_idx_t *mi, const mpa_mapopt_t *opt, const mpa_qbatch_t *q, int n_threads)
{
	return mpa::guarded<mpa_batch_t *>(nullptr, [&] { return mpa_batch_begin_impl(mi, opt, q, n_threads); });
}

int64_t mpa_batch_dp_tasks(mpa_batch_t *b, const mpa_dp_task_t **tasks, mpa_dpopt_t *opt)
{
	if (opt) *opt = b->dpopt;
	if (b->round == 0) {
		double t0 = now_ms();
		emit_round1(b);
		timing_note("emit round 1", now_ms() - t0);
		b->round = 1;
		if (b->tasks.empty()) { take_round3(b, nullptr, nullptr); b->round = 4; }   // nothing to align at all
	}
	// (round 2 without tasks: every accepted span took the ungapped shortcut)
	if (b->round == 2 && b->tasks.empty()) { take_round3(b, nullptr, nullptr); b->round = 4; }
	if (b->round >= 4) { *tasks = nullptr; return 0; }
	*tasks = b->tasks.data();
	return (int64_t)b->tasks.size();
}

static int mpa_batch_dp_results_impl(mpa_batch_t *b, const mpa_dp_rst_t *rst, const uint32_t *cigar_pool)
{
	double t0 = now_ms();
	if (b->round == 1) { take_round1_emit_round2(b, rst, cigar_pool); b->round = 2; timing_note("take 1 / emit 2", now_ms() - t0); }
	else if (b->round == 2) { take_round3(b, rst, cigar_pool); b->round = 4; b->tasks.clear(); timing_note("take 2 + finish", now_ms() - t0); }
	else { set_error("mpa_batch_dp_results called out of sequence"); return MPA_ERR_ARG; }
	return MPA_OK;
}

int mpa_batch_dp_results(mpa_batch_t *b, const mpa_dp_rst_t *rst, const uint32_t *cigar_pool)
{
	return mpa::guarded<int>(MPA_ERR_HIP, [&] { return mpa_batch_dp_results_impl(b, rst, cigar_pool); });
}

static mpa_result_t *mpa_batch_finish_impl(mpa_batch_t *b)
{
	double t_fin = now_ms();
	mpa_result_s *res = new mpa_result_s();
	res->n_seq = (int32_t)b->qs.size();
	res->hit_off.assign(b->qs.size() + 1, 0);
	for (size_t qi = 0; qi < b->qs.size(); ++qi) {
		QueryState &qs = b->qs[qi];
		res->hit_off[qi] = (int64_t)res->hits.size();
		for (const Region &r : qs.regs) {
			mpa_hit_t h;
			memset(&h, 0, sizeof(h));
			h.qid = qs.qid, h.id = r.id, h.parent = r.parent, h.n_sub = r.n_sub, h.subsc = r.subsc, h.cnt = r.cnt;
			h.n_exon = r.n_exon, h.chn_sc = r.chn_sc, h.chn_sc_ungap = r.chn_sc_ungap, h.vid = r.vid;
			h.qs = r.qs, h.qe = r.qe, h.vs = r.vs, h.ve = r.ve, h.has_aln = r.aligned;
			h.dp_score = r.dp_score, h.dp_max = r.dp_max, h.dp_max2 = r.dp_max2, h.blen = r.blen, h.n_fs = r.n_fs, h.n_stop = r.n_stop;
			h.dist_stop = r.dist_stop, h.dist_start = r.dist_start, h.n_iden = r.n_iden, h.n_plus = r.n_plus;
			h.n_cigar = (int32_t)r.cigar.size(), h.n_feat = (int32_t)r.feat.size();
			h.cigar_off = (int64_t)res->cigars.size(), h.feat_off = (int64_t)res->feats.size();
			res->cigars.insert(res->cigars.end(), r.cigar.begin(), r.cigar.end());
			for (const Feat &f : r.feat) {
				mpa_feat_t o;
				memset(&o, 0, sizeof(o));
				o.vs = f.vs, o.ve = f.ve, o.qs = f.qs, o.qe = f.qe, o.type = f.type, o.phase = f.phase, o.n_fs = f.n_fs, o.n_stop = f.n_stop;
				o.score = f.score, o.n_iden = f.n_iden, o.blen = f.blen;
				memcpy(o.donor, f.donor, 2), memcpy(o.acceptor, f.acceptor, 2);
				res->feats.push_back(o);
			}
			res->hits.push_back(h);
		}
	}
	res->hit_off[b->qs.size()] = (int64_t)res->hits.size();
	parallel_for(b->n_threads, (int64_t)b->qs.size(), [&](int64_t i) { b->qs[i] = QueryState(); });   // free per-query state in parallel
	delete b;
	timing_note("batch_finish (flatten)", now_ms() - t_fin);
	return res;
}

mpa_result_t *mpa_batch_finish(mpa_batch_t *b)
{
	return mpa::guarded<mpa_result_t *>(nullptr, [&] { return mpa_batch_finish_impl(b); });
}

int64_t mpa_result_n_hit(const mpa_result_t *r) { return (int64_t)r->hits.size(); }
const mpa_hit_t *mpa_result_hits(const mpa_result_t *r) { return r->hits.data(); }
const int64_t *mpa_result_hit_off(const mpa_result_t *r) { return r->hit_off.data(); }
const uint32_t *mpa_result_cigars(const mpa_result_t *r) { return r->cigars.data(); }
const mpa_feat_t *mpa_result_feats(const mpa_result_t *r) { return r->feats.data(); }
void mpa_result_destroy(mpa_result_t *r) { delete r; }

// one (sub-)batch through the stage machine with the HIP executor
static int run_dp_rounds(mpa_ctx_t *ctx, const mpa_idx_t *mi, const mpa_qbatch_t *q, mpa_batch_t *b)
{
	const mpa_dp_task_t *tasks;
	mpa_dpopt_t dpopt;
	int64_t n;
	std::vector<mpa_dp_rst_t> rst;
	while ((n = mpa_batch_dp_tasks(b, &tasks, &dpopt)) > 0) {
		uint32_t *pool = nullptr;
		int64_t n_pool = 0;
		rst.resize((size_t)n);
		double t0 = now_ms();
		int rc = mpa_dp_run(ctx, mi, &dpopt, q, n, tasks, rst.data(), &pool, &n_pool);
		timing_note("mpa_dp_run (total)", now_ms() - t0);
		if (rc != MPA_OK) { free(pool); return rc; }
		rc = mpa_batch_dp_results(b, rst.data(), pool);
		free(pool);
		if (rc != MPA_OK) return rc;
	}
	return MPA_OK;
}

static int run_batch_on(mpa_ctx_t *ctx, const mpa_idx_t *mi, const mpa_mapopt_t *opt, const mpa_qbatch_t *q, int n_threads, mpa_result_t **out)
{
	*out = nullptr;
	mpa_batch_t *b = batch_begin_impl(ctx, mi, opt, q, n_threads);
	if (!b) return MPA_ERR_ARG;
	const int rc = run_dp_rounds(ctx, mi, q, b);
	if (rc != MPA_OK) { delete b; return rc; }
	*out = mpa_batch_finish(b);
	return MPA_OK;
}

// Several mini-batches as a three-stage software pipeline: [seed..plan of batch k+1] | [DP rounds of batch k on the GPU]
// | [flatten + format batch k-1].  Each stage is one host thread driving its own worker-pool lane.
// claim != nullptr: the job's batches are handed out by the caller -- claim(user) returns the index (into batches[]) of the next
// batch this call should map, or -1 when the job has none left -- so that several callers (one process per GPU) can share one
// job and each takes work as fast as it gets through it (the kt_for work stealing of map.c:264-271, between processes).
// results / text / text_len are then in CLAIM order, order[j] says which batch slot j holds, *n_mapped how many there are.
static int mpa_map_batches_impl(mpa_ctx_t *ctx, const mpa_idx_t *mi, const mpa_mapopt_t *opt, int32_t n_batches, const mpa_qbatch_t *batches,
                    const char *const *const *names, int n_threads, mpa_result_t **results, int64_t *id_io, char **text, int64_t *text_len,
                    mpa_claim_fn claim = nullptr, void *claim_user = nullptr, int32_t *n_mapped = nullptr, int32_t *order = nullptr, int pipe = 0)
{
	const int lane0 = 24 * (pipe & 7);                    // worker-pool lanes of this pipeline (several pipelines of one process: one per device)
	for (int32_t k = 0; k < n_batches; ++k) { results[k] = nullptr; if (text) text[k] = nullptr; if (text_len) text_len[k] = 0; }
	if (!ctx) { set_error("mpa_map_batches needs a device context: the DP has no CPU fallback"); return MPA_ERR_NO_DEVICE; }
	if (names && (!text || !text_len || !id_io)) { set_error("mpa_map_batches: names given without text/text_len/id_io"); return MPA_ERR_ARG; }
	g_trace_t0 = now_ms();
	struct Slot {
		mpa_batch_t *b = nullptr;
		int32_t g = 0, j = 0;                             // index into batches[]; ordinal of the batch among those this call maps
		mpa_qbatch_t q{};                                 // the batch's queries
		bool sketched = false, seeded = false, begun = false, dp_done = false;
	};
	std::vector<Slot> slot((size_t)n_batches);
	std::vector<uint8_t> taken((size_t)n_batches, 0);   // batches a claim has handed out already
	std::mutex mu;
	std::condition_variable cv;
	int32_t n_dp_done = 0, n_planned = 0, rc_all = MPA_OK, next_seed = 0, next_plan = 0;
	int32_t n_total = (int32_t)slot.size();               // units this call maps: shrinks to the number created when the job runs out (guarded by mu)
	int32_t n_created = 0, n_claimed = 0, next_own = 0;   // units created so far; batches taken; (without a claim function) the next batch
	bool exhausted = false;
	if (n_mapped) *n_mapped = 0;
	std::string err;
	// DP lanes = batches in their DP rounds at the same time.  A round is one k_dp_round launch (one hardware queue), so four
	// lanes fit next to the seeding streams; measured at config 3 (20 000 proteins in 10 batches, two planners, two seeders):
	// 3 lanes 0.91 s, 4 lanes 0.865 s; with one planner and one seeder the host stages bound the stream (1.00 s, any lane count)
	// round 3 (both chaining rounds on the device, sleeping waits: the host stages no longer bound the stream): 5 lanes 93.5 ms per
	// 4 000-protein step, 4 lanes 99.8, 6 lanes 100-112 (config 3, 20 steps)
	int n_lanes = 5;
	if (const char *e = getenv("MPA_DP_LANES")) n_lanes = atoi(e);
	n_lanes = std::max(1, std::min(n_lanes, 8));
	std::vector<mpa_ctx_t*> lane_ctx((size_t)n_lanes, ctx);
	for (int d = 1; d < n_lanes; ++d) if (!(lane_ctx[d] = ctx_sibling(ctx, d))) return MPA_ERR_HIP;
	for (int d = 0; d < n_lanes; ++d) ctx_set_role(ctx, lane_ctx[d], 0);
	// HIP maps streams to its hardware queues by creation order; with 17 streams per context the first side streams of the
	// third context land on the queues of the second one's (measured, rocprofv3 queue ids), three further on they do not
	if (n_lanes > 2) ctx_set_side_offset(lane_ctx[2], 3);
	// Planners: the planning stage (pre-chain extraction, chaining, refinement, plans) is memory-bound host work that gains
	// nothing from more than ~32 threads, so two batches are planned side by side with half the threads each
	// (MPA_PLANNERS, default 2).  The seeder's own streams and buffers, one set per batch that can be between seeding and
	// the end of planning: the result of batch k is consumed while batches k+1 .. are on the device.
	// (round 3: with the whole refinement on the device a planner mostly waits for it -- three of them, a third of the threads each)
	// (round 5, once the contexts' idle side streams no longer took hardware queues away: four planners and three seeders, +5 %)
	int n_plan = 4;
	if (const char *e = getenv("MPA_PLANNERS")) n_plan = atoi(e);
	n_plan = std::max(1, std::min(n_plan, 6));
	// Seeders: the device-seeding stage of a batch is mostly waiting for its kernels, which share the GPU with the DP rounds in
	// flight; two batches are seeded side by side (MPA_SEEDERS, default 2)
	int n_seed = 3;
	if (const char *e = getenv("MPA_SEEDERS")) n_seed = atoi(e);
	n_seed = std::max(1, std::min(n_seed, 4));
	// Depth of the pipeline.  A seeding context is busy from the start of a batch's seeding to the end of its planning, and a
	// planned batch waits for a DP lane: seeding of batch k waits for the plan of batch k - n_seed_ctx, the plan of batch k for
	// the end of the DP of batch k - (lanes + planners).  Deeper coupling was measured to change nothing (rounds 2-3).
	const int plan_ahead = 0;
	// Device contexts (streams + pools): one per seeder and one per planner.  What a seeded batch leaves for its planner is pinned
	// host memory only (SeedHold): n_seed_ctx holders, one per batch that can be between the start of seeding and the end of planning.
	const int n_seed_ctx = n_plan + n_seed;
	std::vector<mpa_ctx_t*> seed_dev((size_t)n_seed, nullptr), plan_dev((size_t)n_plan, nullptr);
	for (int k = 0; k < n_seed; ++k) if (!(seed_dev[k] = ctx_sibling(ctx, n_lanes + k))) return MPA_ERR_HIP;
	for (int k = 0; k < n_plan; ++k) if (!(plan_dev[k] = ctx_sibling(ctx, n_lanes + n_seed + k))) return MPA_ERR_HIP;
	for (int k = 0; k < n_seed; ++k) ctx_set_role(ctx, seed_dev[k], 1);
	for (int k = 0; k < n_plan; ++k) ctx_set_role(ctx, plan_dev[k], 2);
	std::vector<SeedHold*> hold((size_t)n_seed_ctx, nullptr);
	for (int k = 0; k < n_seed_ctx; ++k) hold[k] = ctx_seed_hold(ctx, k);
	static const char *const kSeedName[4] = { "seed0", "seed1", "seed2", "seed3" };
	static const char *const kPlanName[6] = { "plan0", "plan1", "plan2", "plan3", "plan4", "plan5" };
	auto fail = [&](int rc, const std::string &msg) { std::lock_guard<std::mutex> g(mu); if (rc_all == MPA_OK) rc_all = rc, err = msg; cv.notify_all(); };

	// (an exception inside a stage thread -- out of memory, say -- must become an error code, not std::terminate)
	// Sketcher: the host half of seeding (protein sketch, bucket lookup, occurrence cut-off, seed jobs: ~27 ms per 4 000 proteins)
	// as a stage of its own, one or two batches ahead of the seeders, whose stage is then device work only -- it was the stage that
	// bounded the stream (141 ms of wall per batch on two seeders, round 4).  It also claims the job's batches.
	auto sketcher_body = [&]() { try {
		tl_pool_lane = lane0 + 20, tl_thread_div = 4;
		WorkerPool::get(tl_pool_lane).label_ = "sketch: worker pool";
		CpuSpan cs("sketch: stage thread");
		for (;;) {
			int32_t k;
			{	// stay at most n_seed + 1 batches ahead of the seeders
				std::unique_lock<std::mutex> g(mu);
				cv.wait(g, [&] { return rc_all != MPA_OK || n_created - next_seed <= n_seed; });
				if (rc_all != MPA_OK) return;
			}
			{	// (a claim may be a round trip to another process: outside the pipeline's lock)
				const int32_t want = claim ? claim(claim_user) : (next_own < n_batches ? next_own++ : -1);
				std::lock_guard<std::mutex> g(mu);
				if (want < 0 || want >= n_batches) { exhausted = true, n_total = n_created; cv.notify_all(); return; }   // the job has no batch left
				// (a claim function that hands a batch out twice -- a counter that was not reset, a wrong job key -- would run the
				// slots, results[] and order[] past the caller's arrays: an error, not heap corruption)
				if (n_created >= n_batches || taken[(size_t)want]) {
					if (rc_all == MPA_OK) rc_all = MPA_ERR_ARG, err = "mpa_map_batches_claim: claim() returned batch " + std::to_string(want) + " twice (or more batches than the job has)";
					exhausted = true, n_total = n_created; cv.notify_all(); return;
				}
				taken[(size_t)want] = 1;
				k = n_created;
				Slot &u = slot[(size_t)k];
				u.g = want, u.j = n_claimed, u.q = batches[want];
				++n_created, ++n_claimed;
			}
			trace("sketch", k, "begin");
			mpa_batch_t *b;
			{ StageClock sc(4); b = batch_sketch_phase(true, mi, opt, &slot[k].q, n_threads); }
			trace("sketch", k, "end");
			if (!b) { fail(MPA_ERR_ARG, mpa_last_error()); return; }
			std::lock_guard<std::mutex> g(mu);
			slot[k].b = b, slot[k].sketched = true;
			cv.notify_all();
		}
	} catch (const std::exception &e) { fail(MPA_ERR_HIP, std::string("sketch stage: ") + e.what()); } };
	auto seeder_body = [&](int sd) { try {
		tl_pool_lane = lane0 + 10 + sd, tl_thread_div = 4;
		WorkerPool::get(tl_pool_lane).label_ = "seeding: worker pool";
		CpuSpan cs("seeding: stage thread (waits for the device)");
		for (;;) {
			int32_t k;
			{	// a free seeder takes the next sketched batch in input order; its result lives in a holder until the batch is planned:
				// wait for the batch that had this holder before
				std::unique_lock<std::mutex> g(mu);
				cv.wait(g, [&] { return rc_all != MPA_OK || (next_seed < n_created && slot[next_seed].sketched) || (exhausted && next_seed >= n_total); });
				if (rc_all != MPA_OK || !(next_seed < n_created && slot[next_seed].sketched)) return;
				k = next_seed++;
				cv.notify_all();                                      // (the sketcher may go one further)
				cv.wait(g, [&] { return rc_all != MPA_OK || k < n_seed_ctx || slot[k - n_seed_ctx].begun; });
				if (rc_all != MPA_OK) return;
			}
			trace(kSeedName[sd], k, "begin");
			bool ok;
			{ StageClock sc(0); ok = batch_device_seed_phase(seed_dev[sd], slot[k].b, true, hold[k % n_seed_ctx]); }
			trace(kSeedName[sd], k, "end");
			if (!ok) { fail(MPA_ERR_HIP, mpa_last_error()); return; }
			std::lock_guard<std::mutex> g(mu);
			slot[k].seeded = true;
			cv.notify_all();
		}
	} catch (const std::exception &e) { fail(MPA_ERR_HIP, std::string("seeding stage: ") + e.what()); } };
	std::thread sketcher(sketcher_body);
	std::vector<std::thread> seeders;
	for (int sd = 0; sd < n_seed; ++sd) seeders.emplace_back(seeder_body, sd);
	auto planner_body = [&](int pl) { try {
		tl_pool_lane = lane0 + 14 + pl, tl_thread_div = n_plan;
		WorkerPool::get(tl_pool_lane).label_ = "planning: worker pool";
		CpuSpan cs("planning: stage thread (serial parts + waits)");
		for (;;) {
			int32_t k;
			{	// stay a bounded number of batches ahead of the ones in their DP rounds (planned batches hold their windows and tasks)
				std::unique_lock<std::mutex> g(mu);
				if (next_plan >= n_total) return;
				k = next_plan++;
				cv.wait(g, [&] { return rc_all != MPA_OK || k >= n_total || (slot[k].seeded && k <= n_dp_done + n_lanes + n_plan - 1 + plan_ahead); });
				if (rc_all != MPA_OK || k >= n_total) return;       // (k >= n_total: the job ran out before this slot was claimed)
			}
			trace(kPlanName[pl], k, "begin");
			{ StageClock sc(1); batch_plan_phase(slot[k].b, plan_dev[pl]); }
			trace(kPlanName[pl], k, "end");
			std::lock_guard<std::mutex> g(mu);
			slot[k].begun = true, ++n_planned;
			cv.notify_all();
		}
	} catch (const std::exception &e) { fail(MPA_ERR_HIP, std::string("planning stage: ") + e.what()); } };
	std::vector<std::thread> planners;
	for (int pl = 0; pl < n_plan; ++pl) planners.emplace_back(planner_body, pl);
	std::thread finisher([&] { try {
		tl_pool_lane = lane0 + 1, tl_thread_div = 4;
		WorkerPool::get(tl_pool_lane).label_ = "output: worker pool";
		CpuSpan cs("output: stage thread (flatten + format)");
		for (int32_t k = 0; k < (int32_t)slot.size(); ++k) {
			{
				std::unique_lock<std::mutex> g(mu);
				cv.wait(g, [&] { return rc_all != MPA_OK || k >= n_total || slot[k].dp_done; });
				if (k >= n_total || !slot[k].dp_done) return;
			}
			trace("finish", k, "begin");
			const Slot &u = slot[k];
			StageClock sc(3);
			mpa_result_t *r = mpa_batch_finish(slot[k].b);
			slot[k].b = nullptr;
			char *t = nullptr;
			int64_t tl = 0;
			if (names) tl = mpa_format_output(mi, opt, &u.q, names[u.g], r, id_io, &t);   // (batches finish in input order: the hit ids run on)
			results[u.j] = r;
			if (names) text[u.j] = t, text_len[u.j] = tl;
			trace("finish", k, "end");
		}
	} catch (const std::exception &e) { fail(MPA_ERR_HIP, std::string("output stage: ") + e.what()); } });
	// DP lanes: lane d drives the DP rounds of batches d, d + n_lanes, ... on its own device context (streams + buffers).
	// The rounds of one batch are dominated by the tails of a few very long DP calls, during which the GPU is nearly
	// idle; the next batch's rounds fill that space.
	// a lane that is free takes the next batch in input order (a static deal would leave lanes idle behind a slow batch)
	int32_t next_dp = 0;
	auto dp_lane = [&](int d, mpa_ctx_t *cd) { try {
		tl_pool_lane = lane0 + 2 + d, tl_thread_div = 4;
		WorkerPool::get(tl_pool_lane).label_ = "DP lanes: worker pool";
		CpuSpan cs("DP lanes: lane thread (round bookkeeping + waits)");
		for (;;) {
			int32_t k;
			{
				std::unique_lock<std::mutex> g(mu);
				if (next_dp >= n_total) return;
				k = next_dp++;
				cv.wait(g, [&] { return rc_all != MPA_OK || k >= n_total || slot[k].begun; });
				if (rc_all != MPA_OK || k >= n_total) return;
			}
			static const char *const lane_name[8] = { "dp0", "dp1", "dp2", "dp3", "dp4", "dp5", "dp6", "dp7" };
			trace(lane_name[d], k, "begin");
			int rc;
			{ StageClock sc(2); rc = run_dp_rounds(cd, mi, &slot[k].q, slot[k].b); }
			trace(lane_name[d], k, "end");
			if (rc != MPA_OK) { fail(rc, mpa_last_error()); return; }
			std::lock_guard<std::mutex> g(mu);
			slot[k].dp_done = true, ++n_dp_done;
			cv.notify_all();
		}
	} catch (const std::exception &e) { fail(MPA_ERR_HIP, std::string("DP lane: ") + e.what()); } };
	std::vector<std::thread> lanes;
	for (int d = 1; d < n_lanes; ++d) lanes.emplace_back(dp_lane, d, lane_ctx[d]);
	dp_lane(0, ctx);
	for (auto &t : lanes) t.join();
	tl_thread_div = 1, tl_pool_lane = 0;                   // (lane 0 ran on the caller's thread)
	finisher.join();
	sketcher.join();
	for (auto &t : seeders) t.join();
	for (auto &t : planners) t.join();
	ctx_absorb_sibling_stats(ctx);
	cpu_report_and_reset("mpa_map_batches", now_ms() - g_trace_t0);
	if (rc_all == MPA_OK) {
		if (n_mapped) *n_mapped = n_claimed;
		if (order) for (int32_t k = 0; k < n_total; ++k) order[slot[k].j] = slot[k].g;
	}
	if (rc_all != MPA_OK) {
		for (Slot &sl : slot) delete sl.b;
		for (int32_t k = 0; k < n_batches; ++k) {
			delete results[k], results[k] = nullptr;
			if (text && text[k]) free(text[k]), text[k] = nullptr;
		}
		set_error(err);
	}
	return rc_all;
}

void mpa_stage_clocks(double ms[5], int64_t calls[5], int reset)
{
	for (int k = 0; k < 5; ++k) {
		if (ms) ms[k] = (double)g_stage_us[k].load() / 1000.0;
		if (calls) calls[k] = g_stage_n[k].load();
		if (reset) g_stage_us[k] = 0, g_stage_n[k] = 0;
	}
}

int mpa_map_batches(mpa_ctx_t *ctx, const mpa_idx_t *mi, const mpa_mapopt_t *opt, int32_t n_batches, const mpa_qbatch_t *batches,
                    const char *const *const *names, int n_threads, mpa_result_t **results, int64_t *id_io, char **text, int64_t *text_len)
{
	return mpa::guarded<int>(MPA_ERR_HIP, [&] { return mpa_map_batches_impl(ctx, mi, opt, n_batches, batches, names, n_threads, results, id_io, text, text_len); });
}

int mpa_map_batches_claim(mpa_ctx_t *ctx, const mpa_idx_t *mi, const mpa_mapopt_t *opt, int32_t n_batches, const mpa_qbatch_t *batches,
                          const char *const *const *names, int n_threads, mpa_result_t **results, int64_t *id_io, char **text, int64_t *text_len,
                          mpa_claim_fn claim, void *user, int32_t *n_mapped, int32_t *order)
{
	if (!claim || !n_mapped || !order) { set_error("mpa_map_batches_claim: claim, n_mapped and order are required"); return MPA_ERR_ARG; }
	return mpa::guarded<int>(MPA_ERR_HIP, [&] { return mpa_map_batches_impl(ctx, mi, opt, n_batches, batches, names, n_threads, results, id_io, text, text_len, claim, user, n_mapped, order); });
}

// One job over the GPUs of ONE process: the kt_for of mp_map_file (map.c:264-271, 291) across devices.  A pipeline per device
// context, each claiming the job's next mini-batch from one atomic counter whenever it has room (a faster or less loaded GPU
// maps more of the job); no data-path exchange between them -- the index is resident on every device, the results come back
// to the host anyway.  PAF text is formatted inside the pipelines; GFF3 / GTF text needs the running hit id of the whole file
// (map.c:306), so those jobs are formatted afterwards, in input order, from the exclusive prefix sum of mpa_result_n_output().
static int mpa_map_batches_multi_impl(int n_ctx, mpa_ctx_t *const *ctxs, mpa_idx_t *mi, const mpa_mapopt_t *opt, int32_t n_batches, const mpa_qbatch_t *batches,
                                      const char *const *const *names, int n_threads, mpa_result_t **results, int64_t *id_io, char **text, int64_t *text_len)
{
	if (n_ctx <= 0 || !ctxs) { set_error("mpa_map_batches_multi: no device context"); return MPA_ERR_NO_DEVICE; }
	if (n_ctx > 8) { set_error("mpa_map_batches_multi: at most eight device contexts"); return MPA_ERR_ARG; }
	if (names && (!text || !text_len || !id_io)) { set_error("mpa_map_batches_multi: names given without text/text_len/id_io"); return MPA_ERR_ARG; }
	for (int p = 0; p < n_ctx; ++p) {
		if (!ctxs[p]) { set_error("mpa_map_batches_multi: null context"); return MPA_ERR_NO_DEVICE; }
		const int rc = dev_upload_index(ctxs[p], mi);
		if (rc != MPA_OK) return rc;
	}
	if (n_ctx == 1) return mpa_map_batches_impl(ctxs[0], mi, opt, n_batches, batches, names, n_threads, results, id_io, text, text_len);
	for (int32_t k = 0; k < n_batches; ++k) { results[k] = nullptr; if (text) text[k] = nullptr; if (text_len) text_len[k] = 0; }
	const bool needs_ids = names && (opt->flag & (MPA_MF_GFF | MPA_MF_GTF));
	struct Pipe {
		std::vector<mpa_result_t*> res;
		std::vector<char*> text;
		std::vector<int64_t> tlen;
		std::vector<int32_t> order;
		int32_t n_mapped = 0;
		int64_t id = 0;
		int rc = MPA_OK;
		std::string err;
	};
	std::vector<Pipe> pipes((size_t)n_ctx);
	std::atomic<int32_t> next{0};
	struct Claim { std::atomic<int32_t> *next; int32_t n; };
	Claim cl{ &next, n_batches };
	auto claim = [](void *u) -> int32_t { Claim *c = (Claim*)u; const int32_t k = c->next->fetch_add(1); return k < c->n ? k : -1; };
	const int threads_each = std::max(1, n_threads / n_ctx);
	std::vector<std::thread> th;
	for (int p = 0; p < n_ctx; ++p) {
		Pipe &P = pipes[(size_t)p];
		P.res.assign((size_t)n_batches, nullptr), P.text.assign((size_t)n_batches, nullptr), P.tlen.assign((size_t)n_batches, 0), P.order.assign((size_t)n_batches, -1);
		th.emplace_back([&, p] {
			Pipe &Q = pipes[(size_t)p];
			Q.rc = mpa::guarded<int>(MPA_ERR_HIP, [&] {
				return mpa_map_batches_impl(ctxs[p], mi, opt, n_batches, batches, needs_ids ? nullptr : names, threads_each, Q.res.data(), &Q.id, needs_ids ? nullptr : Q.text.data(),
				                            needs_ids ? nullptr : Q.tlen.data(), claim, &cl, &Q.n_mapped, Q.order.data(), p);
			});
			if (Q.rc != MPA_OK) { Q.err = mpa_last_error(); next.store(n_batches); }   // (a failed pipeline ends the job for the others too)
		});
	}
	for (auto &t : th) t.join();
	int rc = MPA_OK;
	for (Pipe &P : pipes) if (P.rc != MPA_OK && rc == MPA_OK) rc = P.rc, set_error(P.err);
	for (Pipe &P : pipes)
		for (int32_t j = 0; j < P.n_mapped; ++j) {
			const int32_t k = P.order[(size_t)j];
			if (rc == MPA_OK && k >= 0 && k < n_batches) {
				results[k] = P.res[(size_t)j];
				if (names && !needs_ids) text[k] = P.text[(size_t)j], text_len[k] = P.tlen[(size_t)j];
			} else { delete P.res[(size_t)j]; if (P.text[(size_t)j]) free(P.text[(size_t)j]); }
		}
	if (rc != MPA_OK) {
		for (int32_t k = 0; k < n_batches; ++k) { delete results[k], results[k] = nullptr; if (text && text[k]) free(text[k]), text[k] = nullptr; }
		return rc;
	}
	for (int32_t k = 0; k < n_batches; ++k) if (!results[k]) { set_error("mpa_map_batches_multi: a batch was not mapped"); return MPA_ERR_HIP; }
	if (needs_ids) {
		std::vector<int64_t> id0((size_t)n_batches + 1, *id_io);
		for (int32_t k = 0; k < n_batches; ++k) id0[(size_t)k + 1] = id0[(size_t)k] + mpa_result_n_output(opt, &batches[k], results[k]);
		for (int32_t k = 0; k < n_batches; ++k) {               // (mpa_format_output threads its own parallel region over the queries)
			int64_t id = id0[(size_t)k];
			text_len[k] = mpa_format_output(mi, opt, &batches[k], names[k], results[k], &id, &text[k]);
		}
		*id_io = id0[(size_t)n_batches];
	} else if (names) {
		for (int32_t k = 0; k < n_batches; ++k) *id_io += mpa_result_n_output(opt, &batches[k], results[k]);
	}
	return MPA_OK;
}

int mpa_map_batches_multi(int n_ctx, mpa_ctx_t *const *ctxs, mpa_idx_t *mi, const mpa_mapopt_t *opt, int32_t n_batches, const mpa_qbatch_t *batches,
                          const char *const *const *names, int n_threads, mpa_result_t **results, int64_t *id_io, char **text, int64_t *text_len)
{
	return mpa::guarded<int>(MPA_ERR_HIP, [&] { return mpa_map_batches_multi_impl(n_ctx, ctxs, mi, opt, n_batches, batches, names, n_threads, results, id_io, text, text_len); });
}

static int mpa_map_batch_impl(mpa_ctx_t *ctx, const mpa_idx_t *mi, const mpa_mapopt_t *opt, const mpa_qbatch_t *q, int n_threads, mpa_result_t **out)
{
	*out = nullptr;
	if (!ctx) { set_error("mpa_map_batch needs a device context: the DP has no CPU fallback"); return MPA_ERR_NO_DEVICE; }
	return run_batch_on(ctx, mi, opt, q, n_threads, out);
}

int mpa_map_batch(mpa_ctx_t *ctx, const mpa_idx_t *mi, const mpa_mapopt_t *opt, const mpa_qbatch_t *q, int n_threads, mpa_result_t **out)
{
	return mpa::guarded<int>(MPA_ERR_HIP, [&] { return mpa_map_batch_impl(ctx, mi, opt, q, n_threads, out); });
}

} // extern "C"
