// dp_exec.hip -- host-side executor of the batched spliced DP: the implementation of mpa_dp_run().
//
// A batch of ns_global_gs16b() calls (nasw.h:135; call sites align.c:73,288,293,322,327) is turned into
//   1. one k_prep_rows / k_prep_prof launch (per-row records + query profiles, written once to HBM),
//   2. one k_ext_narrow launch (16/32/64 lanes per call, up to 8 calls packed per wavefront) and one k_ext_wide<NW>
//      launch per wide shape class for the score-only extension calls, calls sorted by window length so that
//      co-resident calls finish together (k_ext_huge + k_ext_replay for calls wider than 1024 columns),
//   3. k_glob_narrow / k_glob_wide<NW> launches for the calls that need a CIGAR, chunked by traceback memory, each
//      followed by k_backtrack,
// every launch on its own HIP stream of the context, bracketed by HIP events (mpa_dp_last_stats feeds bench.py's
// roofline record).  The seeding-stage drivers (dev_prechain_forward, dev_refine_scan; kernels in seed_exec.hip) live
// here too.  There is no CPU fallback here by design.
#include <hip/hip_runtime.h>
#include <time.h>
#include <algorithm>
#include <functional>
#include <mutex>
#include <thread>
#include <atomic>
#include <cstring>
#include <cstdlib>
#include <string>
#include <vector>
#include "mpa_internal.h"
#include "host_core.h"
#include "dp_device.h"
#include "chain_core.h"
#include "dp_kernels.hip"
// (layout invariants of ext_wide_body, checked here so that the kernel source's md5 -- what profiles/*_pmc_summary.json is tied to -- stays put)
static_assert(EXT_WIDE_LDS(2) % 16 == 0 && EXT_WIDE_LDS(4) % 16 == 0 && (2 * 64 * PROF_COL_STRIDE) % 16 == 0, "the groups of a workgroup and the record ring behind the profiles start on 16 bytes");
static_assert(EXT_WIDE_RING % 12 == 0 && EXT_WIDE_XIN % 48 == 0 && EXT_WIDE_KROWS % 12 == 0 && EXT_WIDE_KROWS >= 64 + 12 + 12 + 3, "rings in whole 12-row blocks; the key ring holds a flush of 64 rows, a block, and the first wave's lead");
#include "dp_antidiag.hip"
#include "seed_exec.hip"

namespace mpa {

#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { \
	set_error(std::string(#expr) + ": " + hipGetErrorString(e_)); return MPA_ERR_HIP; } } while (0)

// bytes of HBM this process holds through the pools below and the resident index, and how often a pool had to grow
// (mpa_device_bytes / mpa_pool_growths: bench.py's hbm_resident_gb and pool_growth_events)
static std::atomic<long long> g_dev_bytes{0}, g_pool_growths{0};
static thread_local bool tl_alloc_failed = false;         // the last pool request of this thread could not be met (device seeding then declines instead of failing)

struct DevBuf {
	void *p = nullptr;
	size_t cap = 0;
	// high-water mark of this pool over all contexts that play the same part in the stream pipeline (DP lane, seeder, planner):
	// the batches of a job are alike, so what one lane needed for its pool the others will need too -- a context that has to
	// (re)allocate sizes the pool for the largest request any of them has seen, and the first batches of a stream do the growing
	// once for everybody instead of once per context (a growth is a hipFree: it waits for the whole device)
	std::atomic<size_t> *hint = nullptr;
	int ensure(size_t bytes) {
		const size_t asked = bytes;
		if (hint) {
			size_t h = hint->load(std::memory_order_relaxed);
			while (bytes > h && !hint->compare_exchange_weak(h, bytes, std::memory_order_relaxed)) {}
			if (bytes <= cap) return MPA_OK;
			// (the siblings' high-water mark is a guess about batches to come, not a need: it is taken only while it stays within
			// twice the request -- one outlier batch on one lane must not make every lane's pool that large for the rest of the job)
			if (h > bytes && h <= 2 * bytes) bytes = h;
		}
		if (bytes <= cap) return MPA_OK;
		const double t0 = now_ms();
		// (free, then allocate: measured -- round 3, call 18 -- a pool that keeps its old block until the stream is over and only
		// hipMalloc()s pays 25 ms per growth instead of 6: the allocator hands the block just freed straight back, a fresh one is
		// mapped)
		if (p) { (void)hipFree(p); g_dev_bytes -= (long long)cap; ++g_pool_growths; }
		p = nullptr, cap = 0;
		// (round 5 tried an arena -- a few 8-GB chunks carved up on the host instead of ~180 hipMallocs in a cold run's first second:
		// no gain, the cost of a cold start is the VOLUME of device memory the driver maps, ~100 GB in ~2.5 s, however it is asked for:
		// profiles/r05_cli_cold_start.txt)
		// (generous: growing a pool is a hipFree, which waits for the whole device and stalls every pipeline stage; the batches of a
		// job are alike, so a third of slack makes the first allocation of a pool its last in nearly all cases -- but slack and hint
		// are wishes: when the device cannot give that much, the bare request is tried before the call fails)
		size_t want = bytes;
		want += std::max<size_t>(want / 8, std::min<size_t>(want / 3, (size_t)256 << 20)) + 4096;   // a third of slack up to 256 MB, an eighth beyond (round 4: every pool carried a third: 100 GB per rank)
		size_t free_b = 0, total_b = 0;
		if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && want > free_b - (free_b >> 4)) want = asked + 4096;
		hipError_t e = hipMalloc(&p, want);
		if (e != hipSuccess && want > asked + 4096) { (void)hipGetLastError(); want = asked + 4096; e = hipMalloc(&p, want); }
		if (e != hipSuccess) { (void)hipGetLastError(); p = nullptr, tl_alloc_failed = true; set_error(std::string("hipMalloc(") + std::to_string(want) + "): " + hipGetErrorString(e)); return MPA_ERR_HIP; }
		cap = want;
		g_dev_bytes += (long long)cap;
		timing_note("    pool growth (device)", now_ms() - t0);
		return MPA_OK;
	}
	// exactly `bytes` (the caller has added its own slack)
	int ensure_exact(size_t bytes) {
		if (bytes <= cap) return MPA_OK;
		if (p) { (void)hipFree(p); g_dev_bytes -= (long long)cap; ++g_pool_growths; }
		p = nullptr, cap = 0;
		if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); p = nullptr, tl_alloc_failed = true; set_error("hipMalloc(" + std::to_string(bytes) + ") failed"); return MPA_ERR_HIP; }
		cap = bytes;
		g_dev_bytes += (long long)cap;
		return MPA_OK;
	}
	void release() { if (p) { (void)hipFree(p); g_dev_bytes -= (long long)cap; } p = nullptr, cap = 0; }
	template<typename T> T *as() { return (T*)p; }
};

struct DeviceIndex {
	int device = -1;
	uint8_t *seq = nullptr;
	int64_t *ctg_off = nullptr, *ctg_len = nullptr;
	uint32_t *kb = nullptr;                   // k-mer occurrence lists (block ids), uploaded on first GPU seeding call
	size_t kb_bytes = 0;
	uint8_t *spsc = nullptr;                  // splice-score track (--spsc), uploaded with the genome when the index has one
	size_t seq_bytes = 0, spsc_bytes = 0;     // bytes counted into g_dev_bytes for the genome and the track
};

struct HostPinned {
	void *p = nullptr;
	size_t cap = 0;
	int ensure(size_t bytes) {
		if (bytes <= cap) return MPA_OK;
		if (p) (void)hipHostFree(p);
		p = nullptr, cap = 0;
		const size_t want = bytes * 3 / 2 + 4096;   // (re-pinning host memory is slow: grow in big steps)
		if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) { set_error("hipHostMalloc failed"); return MPA_ERR_HIP; }
		cap = want;
		return MPA_OK;
	}
	void release() { if (p) (void)hipHostFree(p); p = nullptr, cap = 0; }
	template<typename T> T *as() { return (T*)p; }
};

// What a seeding call leaves for the planning stage: pinned host memory only.  In the stream pipeline the device pools belong to
// the SEEDER (two of them), the results to the batch (one holder per batch between the start of its seeding and the end of its
// planning), so that a batch waiting to be planned does not pin down a full set of device pools.
struct SeedHold { HostPinned h_pos, h_f, h_pred, h_a, h_U, h_A; };

struct SeedBufs {
	DevBuf jobs, f, pred, mark, flag, idx, tmp, cfirst;
	HostPinned h_jobs;
	SeedHold own;                                                          // results of a call without a holder of its own (blocking path, refinement)
	DevBuf r_win, r_chunk, r_words, r_hits, r_count;      // refinement scan
	HostPinned h_rhits;
	DevBuf pf_qfirst2, val64[2];                                            // first kept anchor of every query; the kept anchors' values
	DevBuf s_meta, s_cur, s_cur2, s_kept, s_base, s_out, s_flag, dkey;      // k_seed_sift: segments + per-query tables, list cursors, per-segment results, dense keys
	HostPinned h_meta, h_back;                                             // ... their staging (up) and qfirst2 / flags / cfirst (down)
	DevBuf x_all;                                                          // device chaining: views, extraction scratch, survivors, main-chain state, chains (carved up per call)
	DevBuf rx_all, rx_keys;                                                // device refinement: pairing tables, pair keys (two buffers), chain state (carved up per call)
	HostPinned h_xoff;                                                     // ... offsets of the chains of every query (down)
	DevBuf c_a, c_f, c_pred, c_mark, c_flag, c_first, c_long;        // chain forward pass (k_chain_fwd, k_chain_fwd_wave: list of long runs + its counter)
	HostPinned hc_a, hc_f, hc_pred;
};


} // namespace mpa

using namespace mpa;

struct mpa_ctx_s {
	int device = 0;
	hipStream_t stream = nullptr;
	static const int kSide = 16;              // side streams: every kernel class of a batch runs concurrently
	hipStream_t side[kSide] = {};
	hipEvent_t ev[6] = {};
	hipEvent_t fork_ev = nullptr;
	hipEvent_t lev[2 * kSide] = {};           // start/stop pair per side-stream launch
	DevBuf tasks, waves, chunks, qseq, rec, prof, tb, cig, ncig, score, extout, bnd, list, rowkey, cigd, cigoff, hkey, xg, units;
	DevBuf lite, ckpt, wlist;                 // checkpointed traceback (dp_device.h): extension-bit words, checkpoints, the calls the walk takes
	HostPinned h_up, h_down, h_pool;          // staging of a DP round's descriptors (host -> device) and of its results: no pageable copies, one wait
	mpa_dp_stats_t stats = {};
	mpa_dp_stats_t total = {};
	size_t tb_budget = (size_t)8 << 30;       // bytes of traceback matrix per k_glob launch
	std::vector<mpa_ctx_s*> siblings;         // extra contexts on the same device for concurrent sub-batches (owned)
	SeedBufs seed;                            // buffers of the GPU seeding stage (seed_exec.hip)
	hipEvent_t wait_ev = nullptr;             // blocking-sync event: a host thread that waits for the device SLEEPS (wait_stream)
	int side_off = 0;                         // first side stream a round uses (lets the DP lanes of a stream of batches sit on different hardware queues)
	hipStream_t seed_stream = nullptr;        // high-priority stream of the seeding kernels: short, and must not queue behind DP tails
	bool no_split = false;                    // this mpa_dp_run() repeats a round whose workgroup hand-off timed out: 512/1024-column calls go to k_ext_huge
	int64_t handoff_retries = 0;              // how often that has happened on this context (mpa_dp_handoff_retries)
	std::vector<SeedHold*> holds;             // result holders of the stream pipeline's batches (owned; ctx_seed_hold)
	struct PoolHints { std::atomic<size_t> dev[3][96]; };
	PoolHints *hints = nullptr;               // (root context only, owned) high-water marks per pipeline role and pool
	bool antidiag = false;                    // (measurement) the 32-column extension class runs on the anti-diagonal prototype, k_ext_antidiag (mpa_dbg_antidiag)
	// ---- DP worker pool (dp_kernels.hip, k_dp_worker).  The pool itself belongs to the ROOT context of a device ...
	mpa_ctx_s *root = nullptr;                // the context this one is a sibling of (nullptr: a root)
	DpPool *dp_pool = nullptr;                // (root) slots + arguments of every lane, one block of device memory
	std::mutex pool_mu;                       // (root) guards pool creation, slot numbers and the interval list
	int pool_slots = 0;                       // (root) slots handed out
	hipEvent_t pool_base = nullptr;           // (root) time zero of the worker launches' intervals
	std::vector<std::pair<float, float>> pool_iv;   // (root) [start, end) of every finished worker launch of the device, ms since pool_base
	// ... a slot, a generation counter, a word of pinned host memory and a worker stream belong to every context that runs DP rounds
	int dp_slot = -1;
	unsigned int dp_gen = 0;
	int32_t *dp_done = nullptr;               // pinned: receives the generation of a round when its last unit has finished
	hipStream_t worker_stream = nullptr;      // the lane's worker launches (never waited for by a round: its workers may be busy with other lanes' units)
	hipEvent_t arm_ev = nullptr;
	struct WorkerLaunch { hipEvent_t e0, e1; };
	std::vector<WorkerLaunch> wl_busy, wl_free;   // event pairs of worker launches not yet harvested / free for reuse
	DevBuf dp_trace;                          // (MPA_DP_TRACE) per-unit start/end ticks of the current round
};

namespace mpa {

// every device pool of a context, in a fixed order (the index is the pool's identity across contexts)
template<typename F> static void ctx_each_devbuf(mpa_ctx_s *ctx, F f)
{
	SeedBufs &B = ctx->seed;
	DevBuf *all[] = { &ctx->tasks, &ctx->waves, &ctx->chunks, &ctx->qseq, &ctx->rec, &ctx->prof, &ctx->tb, &ctx->cig, &ctx->ncig,
	                  &ctx->score, &ctx->extout, &ctx->bnd, &ctx->list, &ctx->rowkey, &ctx->cigd, &ctx->cigoff, &ctx->hkey, &ctx->xg, &ctx->units,
	                  &B.jobs, &B.f, &B.pred, &B.mark, &B.flag, &B.idx, &B.tmp, &B.cfirst,
	                  &B.r_win, &B.r_chunk, &B.r_words, &B.r_hits, &B.r_count,
	                  &B.c_a, &B.c_f, &B.c_pred, &B.c_mark, &B.c_flag, &B.c_first, &B.c_long,
	                  &B.pf_qfirst2, &B.val64[0], &B.val64[1],
	                  &B.s_meta, &B.s_cur, &B.s_cur2, &B.s_kept, &B.s_base, &B.s_out, &B.s_flag, &B.dkey, &B.x_all, &B.rx_all, &B.rx_keys,
	                  &ctx->lite, &ctx->ckpt, &ctx->wlist };
	int k = 0;
	for (DevBuf *b : all) f(*b, k++);
}

// Wait for everything enqueued on a stream -- asleep.  hipStreamSynchronize() spins on the completion signal by default; a
// pipeline keeps eight or nine host threads waiting for the device at any time (DP lanes, seeders, planners), and on a host
// that gives the process a CPU quota (16 cores per GPU on the boxes this was measured on) spinning waiters eat the very cores
// the host stages need.  An event created with hipEventBlockingSync makes the runtime block on the signal instead.
// Measured (round 3): even the "blocking" hipEventSynchronize costs a waiting thread about half a core, so the wait polls the
// event and SLEEPS 100 us between polls after a short burst of immediate ones.
static hipError_t wait_stream(mpa_ctx_t *ctx, hipStream_t s)
{
	if (!ctx->wait_ev) return hipStreamSynchronize(s);
	hipError_t e = hipEventRecord(ctx->wait_ev, s);
	if (e != hipSuccess) return e;
	for (int polls = 0;; ++polls) {
		e = hipEventQuery(ctx->wait_ev);
		if (e != hipErrorNotReady) return e;
		if (polls >= 8) {
			static const long nap_ns = [] { const char *e = getenv("MPA_POLL_US"); const long v = e ? atol(e) : 100; return (v < 1 ? 1 : v > 5000 ? 5000 : v) * 1000L; }();
			struct timespec ts = { 0, nap_ns };
			nanosleep(&ts, nullptr);
		}
	}
}

// A large host array into device memory.  The index arrays are views into the mapped .mpi (page cache) or pageable vectors: a
// plain hipMemcpy stages them through the runtime's own bounce buffer on ONE thread (measured, round 4: the 7.6 GB of a 3 Gbp
// index cost most of the 4.3 s a cold command-line run spends before it maps anything).  Here four host threads copy 32-MB slices
// into two pinned buffers in turn while the DMA engine drains the other one.
static hipError_t upload_large(void *dst, const void *src, size_t bytes, hipStream_t s)
{
	const size_t kSlice = (size_t)32 << 20;
	if (bytes < 4 * kSlice) return hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice);
	void *pin[2] = { nullptr, nullptr };
	hipEvent_t done[2] = { nullptr, nullptr };
	hipError_t e = hipSuccess;
	for (int k = 0; k < 2 && e == hipSuccess; ++k) { e = hipHostMalloc(&pin[k], kSlice, hipHostMallocDefault); if (e == hipSuccess) e = hipEventCreateWithFlags(&done[k], hipEventDisableTiming); }
	if (e == hipSuccess) {
		const int kThreads = 4;
		size_t at = 0;
		for (int k = 0; at < bytes && e == hipSuccess; ++k, at += kSlice) {
			const int b = k & 1;
			const size_t n = std::min(kSlice, bytes - at);
			if (k >= 2) e = hipEventSynchronize(done[b]);              // the copy that last used this buffer has left it
			if (e != hipSuccess) break;
			std::thread th[kThreads];
			const size_t part = (n + kThreads - 1) / kThreads;
			for (int t = 0; t < kThreads; ++t)
				th[t] = std::thread([=] { const size_t o = (size_t)t * part; if (o < n) memcpy((char*)pin[b] + o, (const char*)src + at + o, std::min(part, n - o)); });
			for (auto &t : th) t.join();
			e = hipMemcpyAsync((char*)dst + at, pin[b], n, hipMemcpyHostToDevice, s);
			if (e == hipSuccess) e = hipEventRecord(done[b], s);
		}
		if (e == hipSuccess) e = hipStreamSynchronize(s);
	}
	for (int k = 0; k < 2; ++k) { if (done[k]) (void)hipEventDestroy(done[k]); if (pin[k]) (void)hipHostFree(pin[k]); }
	if (e != hipSuccess) { (void)hipGetLastError(); return hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice); }   // (no pinned memory to be had: the plain copy)
	return hipSuccess;
}

// The seeding / refinement kernels of a context run on a stream of their own, created with the device's highest priority: they are
// short and a pipeline stage waits for each of them (MPA_PRIO_SEED=0: normal priority; MPA_PRIO_MAIN=1: the contexts' main streams
// -- the DP lanes' prep kernels, walks and copies -- get the high priority too).
static void ensure_seed_stream(mpa_ctx_t *ctx)
{
	if (ctx->seed_stream) return;
	static const bool high = [] { const char *e = getenv("MPA_PRIO_SEED"); return !e || atoi(e) != 0; }();
	int least = 0, greatest = 0;
	(void)hipDeviceGetStreamPriorityRange(&least, &greatest);
	if (hipStreamCreateWithPriority(&ctx->seed_stream, hipStreamNonBlocking, high ? greatest : least) != hipSuccess) ctx->seed_stream = ctx->stream;
}

int dev_upload_index(mpa_ctx_t *ctx, mpa_idx_s *mi)
{
	if (ctx->device < 0 || ctx->device >= mpa_idx_s::kMaxDevices) { set_error("device number beyond what an index keeps copies for"); return MPA_ERR_UNSUPPORTED; }
	static std::mutex mu[mpa_idx_s::kMaxDevices];             // one per device: the pipelines of several devices upload side by side
	std::lock_guard<std::mutex> g(mu[ctx->device]);
	if (mi->dev[ctx->device]) return MPA_OK;
	HIP_TRY(hipSetDevice(ctx->device));
	DeviceIndex *d = new DeviceIndex();
	d->device = ctx->device;
	// (an upload that fails half-way gives everything back: the caller may retry, e.g. on the host path, and must not leak HBM)
	struct Undo { DeviceIndex *d; ~Undo() { if (!d) return; (void)hipFree(d->seq); (void)hipFree(d->ctg_off); (void)hipFree(d->ctg_len); (void)hipFree(d->spsc); delete d; } } undo{ d };
	const size_t n = mi->ctg.size();
	std::vector<int64_t> off(n), len(n);
	for (size_t i = 0; i < n; ++i) off[i] = mi->ctg[i].off, len[i] = mi->ctg[i].len;
	HIP_TRY(hipMalloc((void**)&d->seq, mi->seq.size() + 16));
	HIP_TRY(hipMalloc((void**)&d->ctg_off, n * 8 + 8));
	HIP_TRY(hipMalloc((void**)&d->ctg_len, n * 8 + 8));
	{ const double t0 = now_ms(); HIP_TRY(upload_large(d->seq, mi->seq.data(), mi->seq.size(), ctx->stream)); timing_note("index upload: packed genome", now_ms() - t0); }
	HIP_TRY(hipMemcpy(d->ctg_off, off.data(), n * 8, hipMemcpyHostToDevice));
	HIP_TRY(hipMemcpy(d->ctg_len, len.data(), n * 8, hipMemcpyHostToDevice));
	if (!mi->spsc.empty()) {
		HIP_TRY(hipMalloc((void**)&d->spsc, mi->spsc.size() + 16));
		HIP_TRY(hipMemcpy(d->spsc, mi->spsc.data(), mi->spsc.size(), hipMemcpyHostToDevice));
	}
	d->seq_bytes = mi->seq.size() + 16, d->spsc_bytes = mi->spsc.empty() ? 0 : mi->spsc.size() + 16;   // (what was added is what dev_free_index takes off again)
	mi->dev[ctx->device] = d;
	undo.d = nullptr;
	g_dev_bytes += (long long)(d->seq_bytes + d->spsc_bytes);
	return MPA_OK;
}

void dev_free_index(mpa_idx_s *mi)
{
	for (DeviceIndex *&d : mi->dev) {
		if (!d) continue;
		(void)hipSetDevice(d->device);
		(void)hipFree(d->seq); (void)hipFree(d->ctg_off); (void)hipFree(d->ctg_len);
		g_dev_bytes -= (long long)(d->seq_bytes + d->spsc_bytes);
		if (d->kb) { (void)hipFree(d->kb); g_dev_bytes -= (long long)d->kb_bytes; }
		if (d->spsc) (void)hipFree(d->spsc);
		delete d;
		d = nullptr;
	}
}

// ns_log2 (nasw-sse.c:330-338) and the extension-length penalty of nasw-sse.c:426 tabulated as a step
// function.  Evaluated on the host with the same float operations as the reference (no FMA contraction),
// so the kernel needs no floating point at all.
static float log2_poly(float x)
{
	union { float f; uint32_t i; } z = { x };
	float r = (float)((int32_t)((z.i >> 23) & 255) - 128);
	z.i &= ~(255u << 23);
	z.i += 127u << 23;
	r += (-0.34484843f * z.f + 2.02466578f) * z.f - 0.67487759f;
	return r;
}

static int build_pen_table(float coef, int32_t max_x, PenTable *pt)
{
	int32_t n = 0, cur = 0;
	pt->x[n] = INT32_MIN, pt->val[n] = 0, ++n;
	for (int32_t x = 2; x <= max_x; ++x) {
		int32_t v = (int32_t)(coef * log2_poly((float)x) + .5f);
		if (v != cur) {
			if (n >= MPA_PEN_MAX - 1) return -1;
			pt->x[n] = x, pt->val[n] = v, ++n, cur = v;
		}
	}
	pt->n = n;
	for (int32_t k = n; k < MPA_PEN_MAX; ++k) pt->x[k] = INT32_MAX, pt->val[k] = cur;
	return 0;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per device: remember it per (kernel, device), under a lock -- several
// DP lanes and seeders get here at once, and a process may hold contexts on several devices
static hipError_t ensure_dynamic_lds(const void *fn, int device, size_t bytes)
{
	static std::mutex mu;
	static std::vector<std::pair<const void*, int>> done;
	std::lock_guard<std::mutex> g(mu);
	for (auto &d : done) if (d.first == fn && d.second == device) return hipSuccess;
	const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
	if (e == hipSuccess) done.emplace_back(fn, device);
	return e;
}

struct ExtClass { int G, NB; };
// classes 0..2: k_ext_narrow, several calls per wave.  classes 3..6: k_ext_wide<NW>, NW waves per pair of calls.
static const ExtClass kExtClasses[] = { {16, 1}, {32, 1}, {64, 1}, {64, 2}, {64, 4}, {64, 8}, {64, 16} };
static const int kNumExtClasses = 7;

static int ext_class_of(int32_t ncol)
{
	for (int k = 0; k < kNumExtClasses; ++k)
		if (ncol <= kExtClasses[k].G * kExtClasses[k].NB) return k;
	return -1;
}

template<int NW> static hipError_t launch_glob_wide(const GlobArgs &a, int n_groups, hipStream_t s, bool wide_ge = false)
{
	const size_t lds = (size_t)NW * 22 * 64 * 2 + 2 * NW * 16 + 64 * 4;
	if (wide_ge) hipLaunchKernelGGL((k_glob_wide<NW, true>), dim3(n_groups), dim3(NW * 64), lds, s, a);
	else hipLaunchKernelGGL((k_glob_wide<NW, false>), dim3(n_groups), dim3(NW * 64), lds, s, a);
	return hipGetLastError();
}

} // namespace mpa

extern "C" {

// traceback classes 0,1,2 (16/32/64 lanes) and 7 (block-major, > 1024 columns) in one launch; a.waves = whole array
// (the traceback chunks after the first, which do not ride in the round's launch)
static hipError_t launch_glob_narrow(const GlobArgs &a, const int *first, const int *cnt, hipStream_t s, bool wide_ge = false)
{
	const size_t lds = (size_t)22 * 64 * 2 + (size_t)4 * 32 * 4;
	NarrowMap m{};
	const int cls[4] = { 0, 1, 2, 7 };
	int total = 0;
	for (int k = 0; k < 4; ++k) m.first[k] = first[cls[k]], m.cnt[k] = cnt[cls[k]], total += cnt[cls[k]];
	if (wide_ge) hipLaunchKernelGGL(k_glob_narrow<true>, dim3((unsigned)total), dim3(64), lds, s, a, m);
	else hipLaunchKernelGGL(k_glob_narrow<false>, dim3((unsigned)total), dim3(64), lds, s, a, m);
	return hipGetLastError();
}

} // namespace mpa

extern "C" {

int mpa_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

mpa_ctx_t *mpa_ctx_create(int device)
{
	int n = mpa_device_count();
	if (n <= 0 || device < 0 || device >= n) {
		set_error("no usable HIP device (the MI355X DP kernels have no CPU fallback)");
		return nullptr;
	}
	if (hipSetDevice(device) != hipSuccess) { set_error("hipSetDevice failed"); return nullptr; }
	mpa_ctx_s *ctx = new mpa_ctx_s();
	ctx->device = device;
	static const bool main_high = [] { const char *e = getenv("MPA_PRIO_MAIN"); return e && atoi(e) != 0; }();
	int least = 0, greatest = 0;
	(void)hipDeviceGetStreamPriorityRange(&least, &greatest);
	bool ok = (main_high ? hipStreamCreateWithPriority(&ctx->stream, hipStreamNonBlocking, greatest) : hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) == hipSuccess;
	for (auto &e : ctx->ev) ok = ok && hipEventCreate(&e) == hipSuccess;
	for (auto &e : ctx->lev) ok = ok && hipEventCreate(&e) == hipSuccess;
	ok = ok && hipEventCreate(&ctx->fork_ev) == hipSuccess;
	ok = ok && hipEventCreateWithFlags(&ctx->wait_ev, hipEventBlockingSync | hipEventDisableTiming) == hipSuccess;
	if (!ok) {                                            // a null handle would silently alias the legacy default stream
		set_error("creating the context's HIP streams/events failed");
		mpa_ctx_destroy(ctx);
		return nullptr;
	}
	if (const char *s = getenv("MPA_TB_BUDGET_MB")) ctx->tb_budget = (size_t)atoll(s) << 20;
	return ctx;
}

void mpa_ctx_destroy(mpa_ctx_t *ctx)
{
	if (!ctx) return;
	for (mpa_ctx_s *sb : ctx->siblings) mpa_ctx_destroy(sb);
	ctx->siblings.clear();
	(void)hipSetDevice(ctx->device);
	if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
	if (ctx->worker_stream) { (void)hipStreamSynchronize(ctx->worker_stream); (void)hipStreamDestroy(ctx->worker_stream); }   // (workers exit by themselves once no slot has a unit left)
	for (auto *v : { &ctx->wl_busy, &ctx->wl_free }) { for (auto &w : *v) { (void)hipEventDestroy(w.e0); (void)hipEventDestroy(w.e1); } v->clear(); }
	if (ctx->arm_ev) (void)hipEventDestroy(ctx->arm_ev);
	if (ctx->dp_done) (void)hipHostFree(ctx->dp_done);
	if (ctx->pool_base) (void)hipEventDestroy(ctx->pool_base);
	if (ctx->dp_pool) (void)hipFree(ctx->dp_pool);
	ctx->dp_trace.release();
	SeedBufs &B = ctx->seed;
	ctx_each_devbuf(ctx, [](DevBuf &b, int) { b.release(); });
	for (HostPinned *h : { &B.h_jobs, &B.h_rhits, &B.hc_a, &B.hc_f, &B.hc_pred, &B.h_meta, &B.h_back, &B.h_xoff, &ctx->h_up, &ctx->h_down, &ctx->h_pool }) h->release();
	auto drop_hold = [](SeedHold &H) { for (HostPinned *h : { &H.h_pos, &H.h_f, &H.h_pred, &H.h_a, &H.h_U, &H.h_A }) h->release(); };
	drop_hold(B.own);
	for (SeedHold *H : ctx->holds) { drop_hold(*H); delete H; }
	ctx->holds.clear();
	delete ctx->hints, ctx->hints = nullptr;
	for (auto &e : ctx->ev) if (e) (void)hipEventDestroy(e);
	for (auto &e : ctx->lev) if (e) (void)hipEventDestroy(e);
	if (ctx->fork_ev) (void)hipEventDestroy(ctx->fork_ev);
	if (ctx->wait_ev) (void)hipEventDestroy(ctx->wait_ev);
	for (auto &st : ctx->side) if (st) (void)hipStreamDestroy(st);
	if (ctx->seed_stream && ctx->seed_stream != ctx->stream) (void)hipStreamDestroy(ctx->seed_stream);
	if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
	delete ctx;
}

int mpa_idx_to_device(mpa_ctx_t *ctx, mpa_idx_t *mi) { return dev_upload_index(ctx, mi); }

int mpa_idx_build_kmers_device(mpa_ctx_t *ctx, mpa_idx_t *mi)
{
	if (!ctx) { set_error("no device context"); return MPA_ERR_NO_DEVICE; }
	return mpa::guarded<int>(MPA_ERR_HIP, [&] { return dev_index_build(ctx, mi); });
}

} // extern "C"

namespace mpa {
// k-th sibling of a context: same device, own streams and buffers, created on first use
mpa_ctx_t *ctx_sibling(mpa_ctx_t *ctx, int k)
{
	if (k <= 0) return ctx;
	while ((int)ctx->siblings.size() < k) {
		mpa_ctx_t *sb = mpa_ctx_create(ctx->device);
		if (!sb) return nullptr;
		sb->tb_budget = ctx->tb_budget;
		sb->root = ctx;
		ctx->siblings.push_back(sb);
	}
	return ctx->siblings[k - 1];
}
void ctx_set_side_offset(mpa_ctx_t *ctx, int off) { ctx->side_off = off; }
// `ctx` (the root itself or one of its siblings) plays part `role` of the root's stream pipeline: 0 DP lane, 1 seeder, 2 planner
void ctx_set_role(mpa_ctx_t *root, mpa_ctx_t *ctx, int role)
{
	if (!root->hints) {
		root->hints = new mpa_ctx_s::PoolHints();
		for (auto &r : root->hints->dev) for (auto &h : r) h.store(0);
	}
	mpa_ctx_s::PoolHints *H = root->hints;
	ctx_each_devbuf(ctx, [&](DevBuf &b, int k) { b.hint = k < 96 ? &H->dev[role][k] : nullptr; });
}
SeedHold *ctx_seed_hold(mpa_ctx_t *ctx, int k)
{
	while ((int)ctx->holds.size() <= k) ctx->holds.push_back(new SeedHold());
	return ctx->holds[(size_t)k];
}

// (MPA_TIMING) the device pools of a root context and its siblings, largest first: where the HBM of a pipeline goes
void ctx_pool_report(mpa_ctx_t *root)
{
	static const char *const kName[] = { "tasks", "waves", "chunks", "qseq", "rec", "prof", "tb", "cig", "ncig", "score", "extout", "bnd", "list", "rowkey", "cigd", "cigoff", "hkey", "xg", "units",
		"s.jobs", "s.f", "s.pred", "s.mark", "s.flag", "s.idx", "s.tmp", "s.cfirst", "s.r_win", "s.r_chunk", "s.r_words", "s.r_hits", "s.r_count",
		"s.c_a", "s.c_f", "s.c_pred", "s.c_mark", "s.c_flag", "s.c_first", "s.c_long", "s.pf_qfirst2", "s.val64_0", "s.val64_1",
		"s.s_meta", "s.s_cur", "s.s_cur2", "s.s_kept", "s.s_base", "s.s_out", "s.s_flag", "s.dkey", "s.x_all", "s.rx_all", "s.rx_keys", "lite", "ckpt", "wlist" };
	std::vector<mpa_ctx_t*> all{ root };
	for (mpa_ctx_t *sb : root->siblings) all.push_back(sb);
	size_t grand = 0;
	for (size_t c = 0; c < all.size(); ++c) {
		size_t tot = 0;
		std::vector<std::pair<size_t, int>> big;
		ctx_each_devbuf(all[c], [&](DevBuf &b, int k) { tot += b.cap; if (b.cap >= ((size_t)64 << 20)) big.push_back({ b.cap, k }); });
		std::sort(big.rbegin(), big.rend());
		fprintf(stderr, "[mpa-pools] context %zu: %.2f GB;", c, tot / 1e9);
		for (auto &x : big) fprintf(stderr, " %s %.2f", x.second < (int)(sizeof(kName) / sizeof(kName[0])) ? kName[x.second] : "?", x.first / 1e9);
		fprintf(stderr, "\n");
		grand += tot;
	}
	fprintf(stderr, "[mpa-pools] all contexts of the pipeline: %.2f GB of pools (+ the resident index)\n", grand / 1e9);
}

void pool_harvest(mpa_ctx_t *ctx, bool wait);
void ctx_absorb_sibling_stats(mpa_ctx_t *ctx)
{
	if (timing_on()) ctx_pool_report(ctx->root ? ctx->root : ctx);
	pool_harvest(ctx, true);
	for (mpa_ctx_s *sb : ctx->siblings) {
		pool_harvest(sb, true);
		mpa_dp_stats_t &t = ctx->total, &u = sb->total;
		t.n_ext += u.n_ext, t.n_glob += u.n_glob, t.cells_ext += u.cells_ext, t.cells_glob += u.cells_glob, t.rows_prep += u.rows_prep;
		t.alg_bytes_ext += u.alg_bytes_ext, t.alg_bytes_glob += u.alg_bytes_glob;
		t.ms_prep += u.ms_prep, t.ms_ext += u.ms_ext, t.ms_glob += u.ms_glob, t.ms_backtrack += u.ms_backtrack, t.ms_total += u.ms_total;
		t.launches_ext += u.launches_ext, t.launches_glob += u.launches_glob;
		t.cells_ext_round += u.cells_ext_round, t.cells_glob_round += u.cells_glob_round, t.ms_round += u.ms_round, t.launches_round += u.launches_round;
		u = mpa_dp_stats_t();
		ctx->handoff_retries += sb->handoff_retries, sb->handoff_retries = 0;
	}
}
} // namespace mpa

namespace mpa {
// dev_prechain_forward() with k_seed_sift (the default).  The caller has uploaded the jobs.  Per-anchor memory: 16 bytes of
// staging; everything behind the sift is sized by the kept anchors.  The result arrays are written by k_seed_compact straight
// into pinned host memory (no copy kernels, no second pass over HBM).
static int dev_chains_on_device(mpa_ctx_t *ctx, int32_t n_query, int64_t m, int64_t n2, int nb, const uint64_t *key, const uint64_t *val, const int64_t *d_qfirst,
                                const int32_t *h_flag, const ChainParams &pre, const ChainParams &mainp, PrechainSparse &out, SeedHold &H);

static int dev_prechain_forward_sift(mpa_ctx_t *ctx, DeviceIndex *d, uint32_t n_block, const PreParams &pp, int nb, int32_t n_query, const int64_t *qfirst,
                                     const SeedJob *jobs, int64_t n_jobs, PrechainSparse &out, double t_begin, const ChainParams *pre_cp, const ChainParams *main_cp, SeedHold &H)
{
	SeedBufs &B = ctx->seed;
	hipStream_t s = ctx->seed_stream;
	if (n_block >= 0x7fffffffu) { set_error("GPU seeding: more than 2^31 blocks"); return MPA_ERR_UNSUPPORTED; }
	// ---- segments: a query's block space in pieces of ~seg_target anchors (evenly, the kernel adapts inside a segment)
	const int64_t seg_target = [] { const char *e = getenv("MPA_SIFT_SEG"); const int64_t v = e ? atoll(e) : 49152; return v < 256 ? (int64_t)256 : v; }();   // (read per call: the tests flip it)
	static thread_local std::vector<SiftSeg> segs;
	static thread_local std::vector<int64_t> jfirst;
	static thread_local std::vector<int32_t> qseg;
	segs.clear();
	int64_t n_cur = 0;                                         // cursors: one per (segment, list of its query)
	static thread_local std::vector<int64_t> sfirst;          // first staging slot of every query (sift_stage_slots)
	jfirst.assign((size_t)n_query + 1, 0), qseg.assign((size_t)n_query + 1, 0), sfirst.assign((size_t)n_query + 1, 0);
	for (int32_t q = 0; q < n_query; ++q) sfirst[(size_t)q + 1] = sfirst[(size_t)q] + sift_stage_slots(qfirst[q + 1] - qfirst[q]);
	const int64_t n_stage = sfirst[(size_t)n_query];
	for (int64_t j = 0; j < n_jobs; ++j) ++jfirst[(size_t)jobs[j].qid + 1];
	for (int32_t q = 0; q < n_query; ++q) jfirst[(size_t)q + 1] += jfirst[(size_t)q];
	for (int32_t q = 0; q < n_query; ++q) {
		const int64_t na = qfirst[q + 1] - qfirst[q];
		qseg[(size_t)q] = (int32_t)segs.size();
		if (na == 0) continue;
		if (na >= (int64_t)1 << 31) { set_error("GPU seeding: a query with more than 2^31 anchors"); return MPA_ERR_UNSUPPORTED; }
		const int64_t nl_q = jfirst[(size_t)q + 1] - jfirst[(size_t)q];
		if (nl_q > (1 << 20)) { set_error("GPU seeding: a query with more than 2^20 seeds"); return MPA_ERR_UNSUPPORTED; }
		const int64_t ns = std::min<int64_t>((na + seg_target - 1) / seg_target, n_block);
		for (int64_t k = 0; k < ns; ++k) {
			const uint32_t lo = (uint32_t)((uint64_t)n_block * (uint64_t)k / (uint64_t)ns), hi = (uint32_t)((uint64_t)n_block * (uint64_t)(k + 1) / (uint64_t)ns);
			if (hi > lo) {
				if (n_cur > INT32_MAX - nl_q) { set_error("GPU seeding: too many (segment, seed) cursors in one batch"); return MPA_ERR_UNSUPPORTED; }
				segs.push_back(SiftSeg{ q, lo, hi, (int32_t)n_cur });
				n_cur += nl_q;
			}
		}
	}
	qseg[(size_t)n_query] = (int32_t)segs.size();
	const int32_t n_seg = (int32_t)segs.size();
	if (n_seg == 0) return MPA_OK;
	// one pinned block up: qfirst | jfirst | sfirst | segments | qseg
	const size_t meta_q = ((size_t)n_query + 1) * 8, seg_bytes = (size_t)n_seg * sizeof(SiftSeg);
	const size_t off_jf = meta_q, off_sf = 2 * meta_q, off_seg = 3 * meta_q, off_qs = off_seg + seg_bytes, meta_bytes = off_qs + ((size_t)n_query + 1) * 4;
	int rc;
	// (round 6: the two staging arrays of the sift -- 8 B per staging slot each, dead once k_sift_copy has packed the kept anchors --
	// live at the front of the chaining block, which is carved up only behind that copy: 4.4 GB less per seeder at genome scale)
	const size_t stage_bytes = ((size_t)n_stage * 8 + 64 + 255) & ~(size_t)255;
	if ((rc = B.h_meta.ensure(meta_bytes))) return rc;
	char *hm = B.h_meta.as<char>();
	memcpy(hm, qfirst, meta_q), memcpy(hm + off_jf, jfirst.data(), meta_q), memcpy(hm + off_sf, sfirst.data(), meta_q), memcpy(hm + off_seg, segs.data(), seg_bytes), memcpy(hm + off_qs, qseg.data(), ((size_t)n_query + 1) * 4);
	if ((rc = B.s_meta.ensure(meta_bytes)) || (rc = B.s_cur.ensure((size_t)n_cur * 4 + 16)) || (rc = B.s_cur2.ensure((size_t)n_cur * 4 + 16)) ||
	    (rc = B.s_kept.ensure((size_t)n_seg * 4)) || (rc = B.s_base.ensure((size_t)n_seg * 8)) || (rc = B.s_out.ensure(((size_t)n_seg + 1) * 8)) ||
	    (rc = B.s_flag.ensure((size_t)n_query * 4 + 16)) || (rc = B.pf_qfirst2.ensure(meta_q)) || (rc = B.cfirst.ensure(meta_q)) ||
	    (rc = B.x_all.ensure(2 * stage_bytes)) || (rc = B.h_back.ensure(2 * meta_q + (size_t)n_query * 4 + 64))) return rc;
	uint64_t *const stage0 = B.x_all.as<uint64_t>(), *const stage1 = (uint64_t*)(B.x_all.as<char>() + stage_bytes);
	HIP_TRY(hipMemcpyAsync(B.s_meta.p, hm, meta_bytes, hipMemcpyHostToDevice, s));
	HIP_TRY(hipMemsetAsync(B.s_flag.p, 0, (size_t)n_query * 4 + 16, s));
	const char *dm = B.s_meta.as<char>();
	const int64_t *d_qfirst = (const int64_t*)dm, *d_jfirst = (const int64_t*)(dm + off_jf), *d_sfirst = (const int64_t*)(dm + off_sf);
	const SiftSeg *d_segs = (const SiftSeg*)(dm + off_seg);
	const int32_t *d_qseg = (const int32_t*)(dm + off_qs);
	// (MPA_SIFT_CAP=2048, measurement: ranges of half the size need 18 KB of LDS instead of 37 KB -- a workgroup then fits next to
	// four DP workgroups on a CU -- and touch the lists twice as often)
	static const int sift_cap = [] { const char *e = getenv("MPA_SIFT_CAP"); return e ? atoi(e) : 4096; }();
	if (sift_cap == 2048)
		hipLaunchKernelGGL(k_seed_sift<2048>, dim3((unsigned)n_seg), dim3(SIFT_THREADS), 0, s, d_segs, B.jobs.as<SeedJobDev>(), d_jfirst, d_qfirst, d_sfirst, d->kb, n_block, nb,
		                   B.s_cur.as<int32_t>(), B.s_cur2.as<int32_t>(), stage0, stage1, B.s_kept.as<uint32_t>(), B.s_base.as<int64_t>(),
		                   B.s_flag.as<int32_t>());
	else
	hipLaunchKernelGGL(k_seed_sift<4096>, dim3((unsigned)n_seg), dim3(SIFT_THREADS), 0, s, d_segs, B.jobs.as<SeedJobDev>(), d_jfirst, d_qfirst, d_sfirst, d->kb, n_block, nb,
	                   B.s_cur.as<int32_t>(), B.s_cur2.as<int32_t>(), stage0, stage1, B.s_kept.as<uint32_t>(), B.s_base.as<int64_t>(),
	                   B.s_flag.as<int32_t>());
	hipLaunchKernelGGL(k_sift_offsets, dim3(1), dim3(256), 0, s, d_segs, n_seg, n_query, d_qseg, B.s_flag.as<int32_t>(), B.s_kept.as<uint32_t>(), B.s_out.as<int64_t>(),
	                   B.pf_qfirst2.as<int64_t>());
	HIP_TRY(hipGetLastError());
	int64_t *h_qfirst2 = B.h_back.as<int64_t>(), *h_cfirst = h_qfirst2 + (n_query + 1);
	int32_t *h_flag = (int32_t*)(h_cfirst + (n_query + 1));
	HIP_TRY(hipMemcpyAsync(h_qfirst2, B.pf_qfirst2.p, meta_q, hipMemcpyDeviceToHost, s));
	HIP_TRY(hipMemcpyAsync(h_flag, B.s_flag.p, (size_t)n_query * 4, hipMemcpyDeviceToHost, s));
	HIP_TRY(wait_stream(ctx, s));
	const double t_sift = now_ms();
	timing_note("    seed: segments + sift (wait)", t_sift - t_begin);
	int32_t n_declined = 0;
	for (int32_t q = 0; q < n_query; ++q) n_declined += h_flag[q] != 0;
	if (n_declined) {
		out.on_host.assign((size_t)n_query, 0);
		for (int32_t q = 0; q < n_query; ++q) out.on_host[(size_t)q] = h_flag[q] != 0;
	}
	const int64_t n2 = h_qfirst2[n_query];
	if (n2 == 0) return MPA_OK;
	if ((rc = B.dkey.ensure((size_t)n2 * 8)) || (rc = B.val64[0].ensure((size_t)n2 * 8)) || (rc = B.f.ensure((size_t)n2 * 4)) || (rc = B.pred.ensure((size_t)n2 * 4)) ||
	    (rc = B.mark.ensure((size_t)n2 * 4)) || (rc = B.flag.ensure((size_t)n2 * 4)) || (rc = B.idx.ensure((size_t)n2 * 4))) return rc;
	const unsigned nblk = (unsigned)((n2 + 255) / 256);
	const uint64_t *key = B.dkey.as<uint64_t>();
	const uint64_t *val = B.val64[0].as<uint64_t>();
	hipLaunchKernelGGL(k_sift_copy, dim3((unsigned)n_seg), dim3(256), 0, s, d_segs, B.s_flag.as<int32_t>(), B.s_kept.as<uint32_t>(), B.s_base.as<int64_t>(), B.s_out.as<int64_t>(),
	                   stage0, stage1, B.dkey.as<uint64_t>(), B.val64[0].as<uint64_t>());
	hipLaunchKernelGGL(k_seed_fill, dim3(nblk), dim3(256), 0, s, n2, pp.kmer, B.f.as<int32_t>(), B.pred.as<int32_t>(), B.mark.as<int32_t>(), B.flag.as<uint32_t>());
	hipLaunchKernelGGL(k_prechain_fwd<uint64_t>, dim3(nblk), dim3(256), 0, s, key, val, n2, nb, B.pf_qfirst2.as<int64_t>(), pp, B.f.as<int32_t>(), B.pred.as<int32_t>(),
	                   B.mark.as<int32_t>(), B.flag.as<uint32_t>());
	HIP_TRY(hipGetLastError());
	size_t scan_bytes = 0;
	HIP_TRY(rocprim::exclusive_scan(nullptr, scan_bytes, B.flag.as<uint32_t>(), B.idx.as<uint32_t>(), 0u, (size_t)n2, rocprim::plus<uint32_t>(), s));
	if ((rc = B.tmp.ensure(scan_bytes + 256))) return rc;
	HIP_TRY(rocprim::exclusive_scan(B.tmp.p, scan_bytes, B.flag.as<uint32_t>(), B.idx.as<uint32_t>(), 0u, (size_t)n2, rocprim::plus<uint32_t>(), s));
	hipLaunchKernelGGL(k_seed_bounds, dim3((unsigned)(n_query / 256 + 1)), dim3(256), 0, s, B.pf_qfirst2.as<int64_t>(), n_query, n2, B.idx.as<uint32_t>(), B.flag.as<uint32_t>(),
	                   B.cfirst.as<int64_t>());
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipMemcpyAsync(h_cfirst, B.cfirst.p, meta_q, hipMemcpyDeviceToHost, s));
	HIP_TRY(wait_stream(ctx, s));
	const double t_kernels = now_ms();
	memcpy(out.cfirst.data(), h_cfirst, meta_q);
	const int64_t m = out.cfirst[n_query];
	out.m = m;
	timing_note("    seed: copy + pre-chain + scan (wait)", t_kernels - t_sift);
	if (m == 0) return MPA_OK;
	// both chaining rounds on the device (main_cp == nullptr: the caller wants the pre-chain's linked anchors, as rounds 1-2 did)
	{
		if (main_cp && pre_cp) {
			rc = dev_chains_on_device(ctx, n_query, m, n2, nb, key, val, d_qfirst, h_flag, *pre_cp, *main_cp, out, H);
			if (rc != MPA_ERR_UNSUPPORTED) { timing_note("    seed: chains on the device", now_ms() - t_kernels); return rc; }
		}
	}
	if ((rc = H.h_pos.ensure((size_t)m * 4)) || (rc = H.h_f.ensure((size_t)m * 4)) || (rc = H.h_pred.ensure((size_t)m * 4)) || (rc = H.h_a.ensure((size_t)m * 8))) return rc;
	hipLaunchKernelGGL((k_seed_compact<uint64_t, true>), dim3(nblk), dim3(256), 0, s, key, val, n2, nb, B.pf_qfirst2.as<int64_t>(), B.flag.as<uint32_t>(), B.idx.as<uint32_t>(),
	                   B.f.as<int32_t>(), B.pred.as<int32_t>(), H.h_pos.as<int32_t>(), H.h_f.as<int32_t>(), H.h_pred.as<int32_t>(), H.h_a.as<uint64_t>());
	HIP_TRY(hipGetLastError());
	HIP_TRY(wait_stream(ctx, s));
	out.pos = H.h_pos.as<int32_t>(), out.f = H.h_f.as<int32_t>(), out.pred = H.h_pred.as<int32_t>(), out.a = H.h_a.as<uint64_t>();
	timing_note("    seed: compact into pinned memory", now_ms() - t_kernels);
	return MPA_OK;
}

// Both chaining rounds of every query on the device, behind the forward pass of the pre-chain (map.c:186-196):
//   k_seed_compact      the linked anchors of every query as a sparse view (position in the full list, f, pred, anchor), in HBM
//   k_chain_extract     pre-chain extraction (set): the survivors of every query, ascending
//   k_chain_fwd         forward pass of the main chain over them (block anchors, max_dist_x = max_intron)
//   k_chain_extract     main-chain extraction: chains (score, count) and their anchors, sorted by first target position
//   k_offsets2 + k_chain_pack   the chains of all queries, densely, into pinned host memory
// What travels back is what mp_reg_gen_from_block() consumes (a few hundred anchors per query) instead of every linked anchor
// (~11 000 per query at 3 Gbp), and the host no longer spends a core-second per mini-batch on chaining.
// key/val: the kept anchors (dense, sorted), with B.f / B.pred / B.flag / B.idx / B.cfirst from the pre-chain's forward pass.
static int dev_chains_on_device(mpa_ctx_t *ctx, int32_t n_query, int64_t m, int64_t n2, int nb, const uint64_t *key, const uint64_t *val, const int64_t *d_qfirst,
                                const int32_t *h_flag, const ChainParams &pre, const ChainParams &mainp, PrechainSparse &out, SeedHold &H)
{
	SeedBufs &B = ctx->seed;
	hipStream_t s = ctx->seed_stream;
	if (mainp.bbit != pre.bbit || mainp.kmer != pre.kmer) { set_error("device chains: pre-chain and main chain disagree on the anchors"); return MPA_ERR_UNSUPPORTED; }
	// ---- one allocation, carved up: everything is indexed like the view (m entries), `ends` and `stack` have extras per problem
	const size_t M = (size_t)m, NQ = (size_t)n_query;
	size_t at = 0;
	auto carve = [&](size_t bytes) { const size_t o = at; at += (bytes + 255) & ~(size_t)255; return o; };
	const size_t o_vpos = carve(M * 8), o_vf = o_vpos + M * 4, o_vpred = carve(M * 4), o_va = carve(M * 8);
	const size_t o_mark = carve(M * 4), o_order = carve(M * 4), o_ends = carve((M + 64 * NQ + 64) * sizeof(Pair64)), o_tail8 = carve(M * sizeof(Pair64)),
	             o_items = carve(M * sizeof(SparseItem)), o_moved = carve(M * sizeof(SparseItem)), o_merged = carve(M * sizeof(SparseItem)),
	             o_kept = carve(M), o_stack = carve((M / 64 + 6 * NQ + 16) * sizeof(SortRange));
	// (the chain layout's scratch -- packed anchors, sorted u, first positions -- is only live after the sort replay and the
	// extraction: k_chain_extract puts it into the problem's own `moved` and `merged` lists)
	const size_t o_pre_a = carve(M * 8), o_pre_u = carve(M * 8), o_pre_na = carve(NQ * 8 + 8), o_pre_nu = carve(NQ * 8 + 8), o_status = carve(NQ * 4 + 16);
	const size_t o_mf = carve(M * 4), o_mpred = carve(M * 4), o_mmark = carve(M * 4);
	const int32_t kSerialRun = 48;                          // longer runs of the main chain get a wavefront each (k_chain_fwd_wave)
	const size_t long_cap = M / (size_t)(kSerialRun + 1) + 16, o_long = carve(long_cap * sizeof(LongRun)), o_nlong = carve(64);
	// (the main chains go where the pre-chain's view was: it is dead once the pre-chain has been extracted)
	const size_t o_out_a = o_va, o_out_u = o_vpos, o_na = carve(NQ * 8 + 8), o_nu = carve(NQ * 8 + 8), o_offa = carve(NQ * 8 + 16), o_offu = carve(NQ * 8 + 16);
	int rc;
	if ((rc = B.x_all.ensure(at))) return rc;                // (ensure() adds a third of slack: a re-allocation is a hipFree, which waits for the whole device)
	char *X = B.x_all.as<char>();
	HIP_TRY(hipMemsetAsync(X + o_status, 0, NQ * 4 + 16, s));
	const unsigned nblk2 = (unsigned)((n2 + 255) / 256), nblkm = (unsigned)((m + 255) / 256);
	// the sparse view of the pre-chain's forward pass
	hipLaunchKernelGGL((k_seed_compact<uint64_t, true>), dim3(nblk2), dim3(256), 0, s, key, val, n2, nb, B.pf_qfirst2.as<int64_t>(), B.flag.as<uint32_t>(), B.idx.as<uint32_t>(),
	                   B.f.as<int32_t>(), B.pred.as<int32_t>(), (int32_t*)(X + o_vpos), (int32_t*)(X + o_vf), (int32_t*)(X + o_vpred), (uint64_t*)(X + o_va));
	ExtractArgs xa;
	xa.first = B.cfirst.as<int64_t>(), xa.cnt = nullptr, xa.ntot_first = d_qfirst;
	xa.v_pos = (const int32_t*)(X + o_vpos), xa.v_f = (const int32_t*)(X + o_vf), xa.v_pred = (const int32_t*)(X + o_vpred), xa.v_a = (const uint64_t*)(X + o_va);
	xa.mark = (int32_t*)(X + o_mark), xa.order = (int32_t*)(X + o_order), xa.ends = (Pair64*)(X + o_ends), xa.tail8 = (Pair64*)(X + o_tail8);
	xa.items = (SparseItem*)(X + o_items), xa.moved = (SparseItem*)(X + o_moved), xa.merged = (SparseItem*)(X + o_merged);
	xa.kept = (uint8_t*)(X + o_kept), xa.stack = (SortRange*)(X + o_stack);
	xa.a_out = (uint64_t*)(X + o_pre_a), xa.u_out = (uint64_t*)(X + o_pre_u), xa.n_a = (int64_t*)(X + o_pre_na), xa.n_u = (int64_t*)(X + o_pre_nu);
	xa.status = (int32_t*)(X + o_status), xa.p = pre, xa.set_only = 1;
	// MPA_TIMING=2 (debug): per-phase wall clock of the extraction kernel, averaged over the problems of the launch
	static const bool prof = [] { const char *e = getenv("MPA_TIMING"); return e && atoi(e) >= 2; }();
	long long *d_prof = nullptr;
	auto prof_begin = [&]() -> int {
		if (!prof) return MPA_OK;
		HIP_TRY(hipMalloc((void**)&d_prof, NQ * 128 + 64));
		HIP_TRY(hipMemsetAsync(d_prof, 0, NQ * 128, s));
		const int n_prof = (int)NQ;
		HIP_TRY(hipMemcpyToSymbolAsync(HIP_SYMBOL(g_extract_prof_n), &n_prof, sizeof(n_prof), 0, hipMemcpyHostToDevice, s));
		HIP_TRY(hipMemcpyToSymbolAsync(HIP_SYMBOL(g_extract_prof), &d_prof, sizeof(d_prof), 0, hipMemcpyHostToDevice, s));
		return MPA_OK;
	};
	auto prof_end = [&](const char *what) -> int {
		if (!prof) return MPA_OK;
		std::vector<long long> h(NQ * 16);
		HIP_TRY(hipMemcpyAsync(h.data(), d_prof, NQ * 128, hipMemcpyDeviceToHost, s));
		HIP_TRY(wait_stream(ctx, s));
		double sum[4] = { 0, 0, 0, 0 }, sub[4] = { 0, 0, 0, 0 }, mx = 0;
		int64_t cnt = 0, cnt2 = 0;
		std::vector<std::pair<double, size_t>> by_time;
		for (size_t q = 0; q < NQ; ++q) {
			const long long *t = &h[q * 16];
			if (!t[4] || !t[0]) continue;
			for (int k = 0; k < 4; ++k) sum[k] += (double)(t[k + 1] - t[k]) * 1e-5;   // 100 MHz ticks -> ms
			mx = std::max(mx, (double)(t[4] - t[0]) * 1e-5), ++cnt;
			by_time.emplace_back((double)(t[4] - t[0]) * 1e-5, q);
			if (t[5] && t[6] && t[7]) {                            // the two-level sort replay: its parts (stamps 5-7 lie between 0 and 1)
				sub[0] += (double)(t[5] - t[0]) * 1e-5, sub[1] += (double)(t[6] - t[5]) * 1e-5;
				sub[2] += (double)(t[7] - t[6]) * 1e-5, sub[3] += (double)(t[1] - t[7]) * 1e-5, ++cnt2;
			}
		}
		fprintf(stderr, "[mpa-extract-prof] %s: %lld problems; mean ms: sort replay %.2f, trees %.2f, extraction %.2f, output %.2f; slowest problem %.2f ms\n", what, (long long)cnt,
		        sum[0] / std::max<int64_t>(cnt, 1), sum[1] / std::max<int64_t>(cnt, 1), sum[2] / std::max<int64_t>(cnt, 1), sum[3] / std::max<int64_t>(cnt, 1), mx);
		if (cnt2) fprintf(stderr, "[mpa-extract-prof]   two-level replay (%lld problems): level-1 placement %.2f, level-1 walk %.2f, merge %.2f, level 2 %.2f ms\n", (long long)cnt2,
		                  sub[0] / cnt2, sub[1] / cnt2, sub[2] / cnt2, sub[3] / cnt2);
		if (!by_time.empty()) {                                   // the distribution, and what the slowest problems look like
			std::sort(by_time.begin(), by_time.end());
			const size_t n = by_time.size();
			fprintf(stderr, "[mpa-extract-prof]   problem ms: p50 %.2f p90 %.2f p99 %.2f max %.2f\n", by_time[n / 2].first, by_time[n * 9 / 10].first, by_time[n * 99 / 100].first, by_time[n - 1].first);
			for (size_t k = 0; k < std::min<size_t>(n, 6); ++k) {
				const size_t q = by_time[n - 1 - k].second;
				const long long *t = &h[q * 16];
				fprintf(stderr, "[mpa-extract-prof]   slow #%zu: %.2f ms (replay %.2f [lvl2 %.2f] trees %.2f extraction %.2f output %.2f); view %lld, non-roots %lld, high scores %lld, largest level-2 bucket %lld, merged %lld\n", k,
				        by_time[n - 1 - k].first, (double)(t[1] - t[0]) * 1e-5, t[7] ? (double)(t[1] - t[7]) * 1e-5 : 0.0, (double)(t[2] - t[1]) * 1e-5, (double)(t[3] - t[2]) * 1e-5, (double)(t[4] - t[3]) * 1e-5,
				        t[8], t[9], t[10], t[11], t[12]);
			}
			const size_t q = by_time[n / 2].second;
			const long long *t = &h[q * 16];
			fprintf(stderr, "[mpa-extract-prof]   median problem: view %lld, non-roots %lld, high scores %lld, largest level-2 bucket %lld, merged %lld\n", t[8], t[9], t[10], t[11], t[12]);
		}
		long long *none = nullptr;
		HIP_TRY(hipMemcpyToSymbolAsync(HIP_SYMBOL(g_extract_prof), &none, sizeof(none), 0, hipMemcpyHostToDevice, s));
		HIP_TRY(wait_stream(ctx, s));
		HIP_TRY(hipDeviceSynchronize());                   // (debug facility: an extraction launched by another context may still be stamping into the buffer)
		(void)hipFree(d_prof), d_prof = nullptr;
		return MPA_OK;
	};
	if ((rc = prof_begin())) return rc;
	hipLaunchKernelGGL(k_chain_extract, dim3((unsigned)n_query), dim3(64), EXTRACT_LDS_BYTES, s, xa, n_query);
	HIP_TRY(hipGetLastError());
	if ((rc = prof_end("pre-chain"))) return rc;
	// the main chain over the survivors: forward pass ...
	PreParams pm;
	pm.max_dist_x = std::max(mainp.max_dist_x, mainp.bw), pm.max_dist_y = mainp.max_dist_y;
	if (pm.max_dist_y < mainp.bw && !mainp.is_spliced) pm.max_dist_y = mainp.bw;
	pm.bw = mainp.bw, pm.max_skip = mainp.max_skip, pm.max_iter = mainp.max_iter, pm.kmer = mainp.kmer, pm.bbit = mainp.bbit;
	pm.is_spliced = mainp.is_spliced, pm.coef_log = mainp.coef_log, pm.max_dblock = pm.max_dist_x >> mainp.bbit;
	hipLaunchKernelGGL(k_seed_fill, dim3(nblkm), dim3(256), 0, s, m, pm.kmer, (int32_t*)(X + o_mf), (int32_t*)(X + o_mpred), (int32_t*)(X + o_mmark), (uint32_t*)(X + o_mark));
	HIP_TRY(hipMemsetAsync(X + o_nlong, 0, 64, s));
	hipLaunchKernelGGL(k_chain_fwd, dim3(nblkm), dim3(256), 0, s, (const uint64_t*)(X + o_pre_a), m, B.cfirst.as<int64_t>(), (const int64_t*)(X + o_pre_na), n_query, pm,
	                   (int32_t*)(X + o_mf), (int32_t*)(X + o_mpred), (int32_t*)(X + o_mmark), kSerialRun, (LongRun*)(X + o_long), (unsigned int*)(X + o_nlong), (unsigned int)long_cap);
	hipLaunchKernelGGL(k_chain_fwd_wave, dim3((unsigned)std::min<size_t>(long_cap, 65536)), dim3(64), 0, s, (const uint64_t*)(X + o_pre_a), (const LongRun*)(X + o_long),
	                   (const unsigned int*)(X + o_nlong), (unsigned int)long_cap, pm, (int32_t*)(X + o_mf), (int32_t*)(X + o_mpred), (int32_t*)(X + o_mmark));
	HIP_TRY(hipGetLastError());
	// ... and extraction: dense views over the survivors
	xa.cnt = (const int64_t*)(X + o_pre_na), xa.ntot_first = nullptr;
	xa.v_pos = nullptr, xa.v_f = (const int32_t*)(X + o_mf), xa.v_pred = (const int32_t*)(X + o_mpred), xa.v_a = (const uint64_t*)(X + o_pre_a);
	xa.a_out = (uint64_t*)(X + o_out_a), xa.u_out = (uint64_t*)(X + o_out_u), xa.n_a = (int64_t*)(X + o_na), xa.n_u = (int64_t*)(X + o_nu);
	xa.p = mainp, xa.set_only = 0;
	if ((rc = prof_begin())) return rc;
	hipLaunchKernelGGL(k_chain_extract, dim3((unsigned)n_query), dim3(64), EXTRACT_LDS_BYTES, s, xa, n_query);
	if ((rc = prof_end("main chain"))) return rc;
	hipLaunchKernelGGL(k_offsets2, dim3(1), dim3(256), 0, s, (const int64_t*)(X + o_na), (const int64_t*)(X + o_nu), n_query, (int64_t*)(X + o_offa), (int64_t*)(X + o_offu));
	HIP_TRY(hipGetLastError());
	// offsets + status down, then the chains themselves straight into pinned memory
	const size_t offb = (NQ + 1) * 8;
	if ((rc = B.h_xoff.ensure(2 * offb + NQ * 4 + 64))) return rc;
	int64_t *h_offa = B.h_xoff.as<int64_t>(), *h_offu = h_offa + (NQ + 1);
	int32_t *h_status = (int32_t*)(h_offu + (NQ + 1));
	HIP_TRY(hipMemcpyAsync(h_offa, X + o_offa, offb, hipMemcpyDeviceToHost, s));
	HIP_TRY(hipMemcpyAsync(h_offu, X + o_offu, offb, hipMemcpyDeviceToHost, s));
	HIP_TRY(hipMemcpyAsync(h_status, X + o_status, NQ * 4, hipMemcpyDeviceToHost, s));
	HIP_TRY(wait_stream(ctx, s));
	const int64_t tot_a = h_offa[n_query], tot_u = h_offu[n_query];
	if ((rc = H.h_A.ensure((size_t)tot_a * 8 + 64)) || (rc = H.h_U.ensure((size_t)tot_u * 8 + 64))) return rc;
	if (tot_a > 0 || tot_u > 0) {
		hipLaunchKernelGGL(k_chain_pack, dim3((unsigned)n_query), dim3(256), 0, s, B.cfirst.as<int64_t>(), (const int64_t*)(X + o_na), (const int64_t*)(X + o_nu),
		                   (const int64_t*)(X + o_offa), (const int64_t*)(X + o_offu), (const uint64_t*)(X + o_out_a), (const uint64_t*)(X + o_out_u),
		                   H.h_A.as<uint64_t>(), H.h_U.as<uint64_t>());
		HIP_TRY(hipGetLastError());
		HIP_TRY(wait_stream(ctx, s));
	}
	out.a_first.assign(h_offa, h_offa + n_query + 1), out.u_first.assign(h_offu, h_offu + n_query + 1);
	out.A = H.h_A.as<uint64_t>(), out.U = H.h_U.as<uint64_t>();
	out.has_chains = true;
	bool any = !out.on_host.empty();
	for (int32_t q = 0; q < n_query && !any; ++q) any = h_status[q] != 0;
	if (any) {
		if (out.on_host.empty()) out.on_host.assign(NQ, 0);
		for (int32_t q = 0; q < n_query; ++q) if (h_status[q] || (h_flag && h_flag[q])) out.on_host[(size_t)q] = 1;
	}
	return MPA_OK;
}

// GPU seeding for one mini-batch: anchors -> sort -> forward pass of the pre-chain -> the anchors that have a neighbour.
// jobs: the kept seeds of all queries (qid ascending, within a query ascending query position, dst = running anchor
// offset); qfirst[n_query + 1]: first anchor of every query.  out: per query a sparse ChainView's arrays
// (pred = index into the query's part of the view, -1 for none).
int dev_prechain_forward(mpa_ctx_t *ctx, mpa_idx_s *mi, const ChainParams &pre, int32_t n_query, const int64_t *qfirst,
                         const SeedJob *jobs, int64_t n_jobs, PrechainSparse &out, const ChainParams *main, SeedHold *hold)
{
	const int64_t n = qfirst[n_query];
	out.cfirst.assign((size_t)n_query + 1, 0);
	out.pos = out.f = out.pred = nullptr, out.a = nullptr, out.m = 0, out.on_host.clear();
	out.has_chains = false, out.U = out.A = nullptr, out.u_first.clear(), out.a_first.clear();
	if (n == 0 || n_jobs == 0) return MPA_OK;
	if (pre.bbit <= 0) { set_error("GPU pre-chain needs block anchors (bbit > 0)"); return MPA_ERR_UNSUPPORTED; }
	HIP_TRY(hipSetDevice(ctx->device));
	if (dev_upload_index(ctx, mi) != MPA_OK) return MPA_ERR_HIP;
	DeviceIndex *d = mi->dev[ctx->device];
	if (!d->kb) {
		static std::mutex mu[mpa_idx_s::kMaxDevices];            // (per device, like dev_upload_index)
		std::lock_guard<std::mutex> g(mu[ctx->device]);
		if (!d->kb) {
			uint32_t *p = nullptr;
			HIP_TRY(hipMalloc((void**)&p, mi->kb.size() * 4 + 16));
			{ const double t0 = now_ms(); HIP_TRY(upload_large(p, mi->kb.data(), mi->kb.size() * 4, ctx->stream)); timing_note("index upload: occurrence lists", now_ms() - t0); }
			d->kb = p, d->kb_bytes = mi->kb.size() * 4 + 16;
			g_dev_bytes += (long long)d->kb_bytes;
		}
	}
	PreParams pp;
	pp.max_dist_x = std::max(pre.max_dist_x, pre.bw), pp.max_dist_y = pre.max_dist_y;
	if (pp.max_dist_y < pre.bw && !pre.is_spliced) pp.max_dist_y = pre.bw;
	pp.bw = pre.bw, pp.max_skip = pre.max_skip, pp.max_iter = pre.max_iter, pp.kmer = pre.kmer, pp.bbit = pre.bbit;
	pp.is_spliced = pre.is_spliced, pp.coef_log = pre.coef_log, pp.max_dblock = pp.max_dist_x >> pre.bbit;
	int nb = 1, qb = 1;
	while ((1ULL << nb) < (uint64_t)mi->n_block + (uint64_t)pp.max_dblock + 2) ++nb;
	while ((1LL << qb) < n_query) ++qb;
	if (nb + qb > 64) { set_error("GPU pre-chain: too many queries x blocks for a 64-bit key"); return MPA_ERR_UNSUPPORTED; }
	if (pp.max_dblock != 1) { set_error("GPU seeding: the sift assumes a pre-chain that reaches one block"); return MPA_ERR_UNSUPPORTED; }
	{	// the working set is ~60 bytes per anchor (sift: 16 of staging, the rest sized by the kept ones); a batch that does not fit
		// stays on the host (the caller falls back)
		size_t free_b = 0, total_b = 0;
		if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
			const SeedBufs &Bc = ctx->seed;
			const size_t have = Bc.f.cap + Bc.pred.cap + Bc.mark.cap + Bc.flag.cap + Bc.idx.cap + Bc.tmp.cap + Bc.x_all.cap + Bc.dkey.cap + Bc.val64[0].cap;
			if ((size_t)n * 40 > have + free_b - (free_b >> 3)) { set_error("GPU seeding: batch too large for device memory"); return MPA_ERR_UNSUPPORTED; }
		}
	}
	SeedBufs &B = ctx->seed;
	ensure_seed_stream(ctx);
	hipStream_t s = ctx->seed_stream;
	const double t_begin = now_ms();
	// ---- upload the seed jobs
	if (B.h_jobs.ensure((size_t)n_jobs * sizeof(SeedJobDev)) != MPA_OK) return MPA_ERR_HIP;
	SeedJobDev *hj = B.h_jobs.as<SeedJobDev>();
	for (int64_t i = 0; i < n_jobs; ++i) hj[i] = SeedJobDev{ jobs[i].kb_off, jobs[i].dst, jobs[i].cnt, jobs[i].qpos, jobs[i].qid, 0 };
	if (B.jobs.ensure((size_t)n_jobs * sizeof(SeedJobDev)) != MPA_OK) return MPA_ERR_HIP;
	HIP_TRY(hipMemcpyAsync(B.jobs.p, hj, (size_t)n_jobs * sizeof(SeedJobDev), hipMemcpyHostToDevice, s));
	// merge the occurrence lists per query in block order, keep what has a neighbour (k_seed_sift, seed_exec.hip)
	tl_alloc_failed = false;
	const int rc = dev_prechain_forward_sift(ctx, d, mi->n_block, pp, nb, n_query, qfirst, jobs, n_jobs, out, t_begin, &pre, main, hold ? *hold : B.own);
	// a pool that could not grow (the admission check above is an estimate): the batch is seeded on the host, as for any batch
	// that does not fit -- nothing has been handed to the caller yet
	if (rc == MPA_ERR_HIP && tl_alloc_failed) { (void)hipStreamSynchronize(s); return MPA_ERR_UNSUPPORTED; }
	return rc;
}
} // namespace mpa

namespace mpa {
// mp_idx_build's k-mer table on the device (index.c:52-136): scan (count, then emit), one radix sort of all keys, unique,
// bucket histogram + scan.  Leaves kb[] resident for the seeding kernels.  MPA_ERR_UNSUPPORTED (the caller builds on the host):
// parameters outside the kernel's range, or not enough device memory for the keys of this genome.
int dev_index_build(mpa_ctx_t *ctx, mpa_idx_s *mi)
{
	const int32_t n_strand = (int32_t)mi->ctg.size() * 2;
	const mpa_idxopt_t &io = mi->opt;
	const int bucket_bits = io.kmer * 4 - io.mod_bit;
	if (n_strand == 0 || io.kmer < 1 || io.kmer > 7 || io.mod_bit < 0 || bucket_bits < 1 || bucket_bits > 28 || io.bbit < 0 || io.bbit > 20 || io.min_aa_len < io.kmer ||
	    io.min_aa_len > 1000) { set_error("index build: parameters outside the device kernel's range"); return MPA_ERR_UNSUPPORTED; }
	HIP_TRY(hipSetDevice(ctx->device));
	if (dev_upload_index(ctx, mi) != MPA_OK) return MPA_ERR_HIP;
	DeviceIndex *d = mi->dev[ctx->device];
	hipStream_t s = ctx->stream;
	std::vector<int64_t> chunk_first((size_t)n_strand + 1, 0);
	for (int32_t j = 0; j < n_strand; ++j) chunk_first[j + 1] = chunk_first[j] + (mi->ctg[j >> 1].len + REFINE_CHUNK - 1) / REFINE_CHUNK;
	const int64_t n_chunk = chunk_first[n_strand];
	if (n_chunk == 0 || n_chunk > 0x7fffffff) { set_error("index build: genome too small or too large for one launch"); return MPA_ERR_UNSUPPORTED; }
	const size_t n_bucket = (size_t)1 << bucket_bits;
	DevBuf b_first, b_bo, b_count, b_off, b_keys, b_keys2, b_flag, b_idx, b_tmp, b_cnt, b_ki;
	auto release_all = [&]() { for (DevBuf *b : { &b_first, &b_bo, &b_count, &b_off, &b_keys, &b_keys2, &b_flag, &b_idx, &b_tmp, &b_cnt, &b_ki }) b->release(); };
	struct Guard { std::function<void()> f; ~Guard() { f(); } } guard{ release_all };
	int rc;
	if ((rc = b_first.ensure(((size_t)n_strand + 1) * 8)) || (rc = b_bo.ensure((size_t)n_strand * 4 + 4)) || (rc = b_count.ensure((size_t)n_chunk * 4 + 4)) ||
	    (rc = b_off.ensure(((size_t)n_chunk + 1) * 8))) return rc;
	HIP_TRY(hipMemcpyAsync(b_first.p, chunk_first.data(), ((size_t)n_strand + 1) * 8, hipMemcpyHostToDevice, s));
	HIP_TRY(hipMemcpyAsync(b_bo.p, mi->bo.data(), (size_t)n_strand * 4, hipMemcpyHostToDevice, s));
	IndexScanArgs a;
	a.g = DevGenome{ d->seq, d->ctg_off, d->ctg_len, nullptr, mi->l_seq };
	a.chunk_first = b_first.as<int64_t>(), a.bo = b_bo.as<uint32_t>();
	a.n_strand = n_strand, a.kmer = io.kmer, a.mod_bit = io.mod_bit, a.bbit = io.bbit, a.min_aa_len = io.min_aa_len, a.halo = (3 * io.min_aa_len + 2 + 15) & ~15;
	for (int c = 0; c < 64; ++c) a.rt.t[c] = tab_codon()[c] >= 20 ? 0xff : tab_codon13()[c];
	const size_t lds = (size_t)REFINE_CHUNK + 2 * (size_t)a.halo;
	const double t0 = now_ms();
	hipLaunchKernelGGL(k_index_scan<false>, dim3((unsigned)n_chunk), dim3(256), lds, s, a, b_count.as<uint32_t>(), (const uint64_t*)nullptr, (uint64_t*)nullptr);
	HIP_TRY(hipGetLastError());
	// exclusive scan of the per-chunk counts (as 64-bit offsets)
	{
		size_t tmp_bytes = 0;
		auto in = rocprim::make_transform_iterator(b_count.as<uint32_t>(), U32ToU64());
		HIP_TRY(rocprim::exclusive_scan(nullptr, tmp_bytes, in, b_off.as<uint64_t>(), (uint64_t)0, (size_t)n_chunk, rocprim::plus<uint64_t>(), s));
		if ((rc = b_tmp.ensure(tmp_bytes + 256))) return rc;
		HIP_TRY(rocprim::exclusive_scan(b_tmp.p, tmp_bytes, in, b_off.as<uint64_t>(), (uint64_t)0, (size_t)n_chunk, rocprim::plus<uint64_t>(), s));
	}
	uint64_t last_off = 0;
	uint32_t last_cnt = 0;
	HIP_TRY(hipMemcpyAsync(&last_off, b_off.as<uint64_t>() + (n_chunk - 1), 8, hipMemcpyDeviceToHost, s));
	HIP_TRY(hipMemcpyAsync(&last_cnt, b_count.as<uint32_t>() + (n_chunk - 1), 4, hipMemcpyDeviceToHost, s));
	HIP_TRY(wait_stream(ctx, s));
	const int64_t n_keys = (int64_t)(last_off + last_cnt);
	if (n_keys == 0) { mi->ki.assign(n_bucket, 0), mi->kb.clear(), mi->n_kb = 0; return MPA_OK; }
	{	// two key buffers, flags, scan, kb: ~40 bytes per key
		size_t free_b = 0, total_b = 0;
		if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && (size_t)n_keys * 44 + n_bucket * 16 > free_b - (free_b >> 3)) {
			set_error("index build: not enough device memory for the k-mer keys of this genome"); return MPA_ERR_UNSUPPORTED;
		}
	}
	if ((rc = b_keys.ensure((size_t)n_keys * 8)) || (rc = b_keys2.ensure((size_t)n_keys * 8))) return rc;
	hipLaunchKernelGGL(k_index_scan<true>, dim3((unsigned)n_chunk), dim3(256), lds, s, a, (uint32_t*)nullptr, b_off.as<uint64_t>(), b_keys.as<uint64_t>());
	HIP_TRY(hipGetLastError());
	int nb = 1;
	while ((1ULL << nb) < (uint64_t)mi->n_block + 1) ++nb;
	{
		size_t tmp_bytes = 0;
		HIP_TRY(rocprim::radix_sort_keys(nullptr, tmp_bytes, b_keys.as<uint64_t>(), b_keys2.as<uint64_t>(), (size_t)n_keys, 0u, 32u + (unsigned)bucket_bits, s));
		if ((rc = b_tmp.ensure(tmp_bytes + 256))) return rc;
		HIP_TRY(rocprim::radix_sort_keys(b_tmp.p, tmp_bytes, b_keys.as<uint64_t>(), b_keys2.as<uint64_t>(), (size_t)n_keys, 0u, 32u + (unsigned)bucket_bits, s));
	}
	(void)nb;
	b_keys.release();
	const uint64_t *sorted = b_keys2.as<uint64_t>();
	if ((rc = b_flag.ensure((size_t)n_keys * 4)) || (rc = b_idx.ensure((size_t)n_keys * 8)) || (rc = b_cnt.ensure(n_bucket * 8)) || (rc = b_ki.ensure(n_bucket * 8))) return rc;
	const unsigned nblk = (unsigned)((n_keys + 255) / 256);
	hipLaunchKernelGGL(k_index_flag, dim3(nblk), dim3(256), 0, s, sorted, n_keys, b_flag.as<uint32_t>());
	{
		size_t tmp_bytes = 0;
		auto in = rocprim::make_transform_iterator(b_flag.as<uint32_t>(), U32ToU64());
		HIP_TRY(rocprim::exclusive_scan(nullptr, tmp_bytes, in, b_idx.as<uint64_t>(), (uint64_t)0, (size_t)n_keys, rocprim::plus<uint64_t>(), s));
		if ((rc = b_tmp.ensure(tmp_bytes + 256))) return rc;
		HIP_TRY(rocprim::exclusive_scan(b_tmp.p, tmp_bytes, in, b_idx.as<uint64_t>(), (uint64_t)0, (size_t)n_keys, rocprim::plus<uint64_t>(), s));
	}
	uint64_t last_idx = 0;
	uint32_t last_flag = 0;
	HIP_TRY(hipMemcpyAsync(&last_idx, b_idx.as<uint64_t>() + (n_keys - 1), 8, hipMemcpyDeviceToHost, s));
	HIP_TRY(hipMemcpyAsync(&last_flag, b_flag.as<uint32_t>() + (n_keys - 1), 4, hipMemcpyDeviceToHost, s));
	HIP_TRY(wait_stream(ctx, s));
	const int64_t n_kb = (int64_t)(last_idx + last_flag);
	uint32_t *d_kb = nullptr;
	HIP_TRY(hipMalloc((void**)&d_kb, (size_t)n_kb * 4 + 16));
	struct KbGuard { uint32_t *&p; hipStream_t s; ~KbGuard() { if (p) { (void)hipStreamSynchronize(s); (void)hipFree(p); } } } kb_guard{ d_kb, s };   // freed on every error path below
	HIP_TRY(hipMemsetAsync(b_cnt.p, 0, n_bucket * 8, s));
	hipLaunchKernelGGL(k_index_compact, dim3(nblk), dim3(256), 0, s, sorted, n_keys, b_flag.as<uint32_t>(), b_idx.as<uint64_t>(), d_kb, b_cnt.as<unsigned long long>());
	{
		size_t tmp_bytes = 0;
		HIP_TRY(rocprim::exclusive_scan(nullptr, tmp_bytes, b_cnt.as<uint64_t>(), b_ki.as<uint64_t>(), (uint64_t)0, n_bucket, rocprim::plus<uint64_t>(), s));
		if ((rc = b_tmp.ensure(tmp_bytes + 256))) return rc;
		HIP_TRY(rocprim::exclusive_scan(b_tmp.p, tmp_bytes, b_cnt.as<uint64_t>(), b_ki.as<uint64_t>(), (uint64_t)0, n_bucket, rocprim::plus<uint64_t>(), s));
	}
	HIP_TRY(hipGetLastError());
	// (into temporaries: a copy that fails must not leave the index with a half-filled table)
	std::vector<int64_t> ki_new(n_bucket);
	std::vector<uint32_t> kb_new((size_t)n_kb);
	HIP_TRY(hipMemcpyAsync(ki_new.data(), b_ki.p, n_bucket * 8, hipMemcpyDeviceToHost, s));
	HIP_TRY(hipMemcpyAsync(kb_new.data(), d_kb, (size_t)n_kb * 4, hipMemcpyDeviceToHost, s));
	HIP_TRY(wait_stream(ctx, s));
	mi->ki.swap(ki_new), mi->kb.swap(kb_new), mi->n_kb = n_kb;
	if (d->kb) { (void)hipFree(d->kb); g_dev_bytes -= (long long)d->kb_bytes; }
	d->kb = d_kb, d->kb_bytes = (size_t)n_kb * 4 + 16;     // stays resident for the seeding kernels
	g_dev_bytes += (long long)d->kb_bytes;
	d_kb = nullptr;                                        // (ownership moved: the guard lets go)
	timing_note("index build on the GPU", now_ms() - t0);
	return MPA_OK;
}
} // namespace mpa

namespace mpa {
// Forward pass of mp_chain for a batch of chaining problems on the device (k_chain_fwd): the main chain of every query of a
// mini-batch, or the refinement chains of its regions.  The caller writes the sorted anchors of all problems, back to back,
// into io.a (pinned memory of the context) and finds f / pred (index inside the problem) in io.f / io.pred afterwards.
int dev_chain_buffers(mpa_ctx_t *ctx, int64_t n, ChainIO &io)
{
	SeedBufs &B = ctx->seed;
	int rc;
	if ((rc = B.hc_a.ensure((size_t)n * 8 + 64)) || (rc = B.hc_f.ensure((size_t)n * 4 + 64)) || (rc = B.hc_pred.ensure((size_t)n * 4 + 64))) return rc;
	io.a = B.hc_a.as<uint64_t>(), io.f = B.hc_f.as<int32_t>(), io.pred = B.hc_pred.as<int32_t>();
	return MPA_OK;
}

int dev_chain_forward(mpa_ctx_t *ctx, const ChainParams &cp, int32_t n_prob, const int64_t *first, const ChainIO &io)
{
	const int64_t n = first[n_prob];
	if (n == 0 || n_prob == 0) return MPA_OK;
	if (n_prob > (1 << 30) || cp.kmer < 0) { set_error("chain forward pass: parameters outside the device kernel's range"); return MPA_ERR_UNSUPPORTED; }
	for (int32_t q = 0; q < n_prob; ++q)
		if (first[q + 1] - first[q] > INT32_MAX - 2) { set_error("chain forward pass: a problem has too many anchors"); return MPA_ERR_UNSUPPORTED; }
	HIP_TRY(hipSetDevice(ctx->device));
	SeedBufs &B = ctx->seed;
	ensure_seed_stream(ctx);
	hipStream_t s = ctx->seed_stream;
	PreParams pp;
	pp.max_dist_x = std::max(cp.max_dist_x, cp.bw), pp.max_dist_y = cp.max_dist_y;
	if (pp.max_dist_y < cp.bw && !cp.is_spliced) pp.max_dist_y = cp.bw;
	pp.bw = cp.bw, pp.max_skip = cp.max_skip, pp.max_iter = cp.max_iter, pp.kmer = cp.kmer, pp.bbit = cp.bbit;
	pp.is_spliced = cp.is_spliced, pp.coef_log = cp.coef_log, pp.max_dblock = pp.max_dist_x >> cp.bbit;
	int rc;
	if ((rc = B.c_a.ensure((size_t)n * 8)) || (rc = B.c_f.ensure((size_t)n * 4)) || (rc = B.c_pred.ensure((size_t)n * 4)) || (rc = B.c_mark.ensure((size_t)n * 4)) ||
	    (rc = B.c_flag.ensure((size_t)n * 4)) || (rc = B.c_first.ensure(((size_t)n_prob + 1) * 8))) return rc;
	// runs longer than this get a wavefront each (k_chain_fwd_wave); MPA_CHAIN_SERIAL_RUN overrides (tests: 4 = almost every run)
	const int32_t serial_run = [] { const char *e = getenv("MPA_CHAIN_SERIAL_RUN"); return e ? std::max(1, atoi(e)) : 48; }();
	const size_t long_cap = (size_t)n / (size_t)(serial_run + 1) + 16;
	if ((rc = B.c_long.ensure(64 + long_cap * sizeof(LongRun)))) return rc;
	unsigned int *d_nlong = B.c_long.as<unsigned int>();
	LongRun *d_long = (LongRun*)(B.c_long.as<char>() + 64);
	HIP_TRY(hipMemsetAsync(d_nlong, 0, 64, s));
	HIP_TRY(hipMemcpyAsync(B.c_a.p, io.a, (size_t)n * 8, hipMemcpyHostToDevice, s));
	HIP_TRY(hipMemcpyAsync(B.c_first.p, first, ((size_t)n_prob + 1) * 8, hipMemcpyHostToDevice, s));
	const unsigned nblk = (unsigned)((n + 255) / 256);
	hipLaunchKernelGGL(k_seed_fill, dim3(nblk), dim3(256), 0, s, n, pp.kmer, B.c_f.as<int32_t>(), B.c_pred.as<int32_t>(), B.c_mark.as<int32_t>(), B.c_flag.as<uint32_t>());
	hipLaunchKernelGGL(k_chain_fwd, dim3(nblk), dim3(256), 0, s, B.c_a.as<uint64_t>(), n, B.c_first.as<int64_t>(), (const int64_t*)nullptr, n_prob, pp, B.c_f.as<int32_t>(), B.c_pred.as<int32_t>(),
	                   B.c_mark.as<int32_t>(), serial_run, d_long, d_nlong, (unsigned int)long_cap);
	hipLaunchKernelGGL(k_chain_fwd_wave, dim3((unsigned)std::min<size_t>(long_cap, 65536)), dim3(64), 0, s, B.c_a.as<uint64_t>(), (const LongRun*)d_long, (const unsigned int*)d_nlong,
	                   (unsigned int)long_cap, pp, B.c_f.as<int32_t>(), B.c_pred.as<int32_t>(), B.c_mark.as<int32_t>());
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipMemcpyAsync(io.f, B.c_f.p, (size_t)n * 4, hipMemcpyDeviceToHost, s));
	HIP_TRY(hipMemcpyAsync(io.pred, B.c_pred.p, (size_t)n * 4, hipMemcpyDeviceToHost, s));
	HIP_TRY(wait_stream(ctx, s));                       // (first[] may be pageable memory of the caller: it is consumed by now)
	return MPA_OK;
}
} // namespace mpa

namespace mpa {
// Refinement scan of a mini-batch's region windows on the device (k_refine_scan).  qw_first/qwords: the distinct k-mer words
// of every query.  out.first[w] .. out.first[w+1]: the hits (hash << 32 | window position) of window w, unsorted.
// MPA_ERR_UNSUPPORTED (the caller scans on the host): a query with more words than the LDS hash set takes, k too large.
int dev_refine_scan(mpa_ctx_t *ctx, mpa_idx_s *mi, int32_t kmer, int32_t min_aa_len, int32_t n_query, const int64_t *qw_first, const uint32_t *qwords,
                    int64_t n_win, const RefineWindow *wins, RefineHits &out)
{
	out.first.assign((size_t)n_win + 1, 0);
	out.hits.clear();
	if (n_win == 0) return MPA_OK;
	if (kmer < 1 || kmer > 7 || min_aa_len > (REFINE_HALO - 3 * kmer) / 3 + kmer) { set_error("refinement scan: parameters outside the device kernel's range"); return MPA_ERR_UNSUPPORTED; }
	int64_t max_words = 0;
	for (int32_t q = 0; q < n_query; ++q) max_words = std::max(max_words, qw_first[q + 1] - qw_first[q]);
	int hs_log2 = 10;
	while ((1LL << hs_log2) < 2 * max_words) ++hs_log2;
	if (hs_log2 > 13) { set_error("refinement scan: query too long for the LDS k-mer set"); return MPA_ERR_UNSUPPORTED; }
	HIP_TRY(hipSetDevice(ctx->device));
	if (dev_upload_index(ctx, mi) != MPA_OK) return MPA_ERR_HIP;
	SeedBufs &B = ctx->seed;
	ensure_seed_stream(ctx);
	hipStream_t s = ctx->seed_stream;
	// windows, chunks, the queries' k-mer words: laid out in ONE pinned block and uploaded with one copy (pageable copies are
	// staged by the runtime, synchronously and spinning)
	int64_t n_pos = 0, n_chunk = 0;
	for (int64_t k = 0; k < n_win; ++k) n_pos += wins[k].len, n_chunk += (wins[k].len + REFINE_CHUNK - 1) / REFINE_CHUNK;
	if (n_chunk == 0) return MPA_OK;
	const unsigned long long cap = (unsigned long long)(n_pos / 64 + (1 << 20));   // ~0.04 % of the positions hit on random sequence
	const int64_t n_words = qw_first[n_query];
	auto al64 = [](size_t x) { return (x + 63) & ~(size_t)63; };
	const size_t o_win = 0, o_chunk = al64((size_t)n_win * sizeof(RefineWindowDev)), o_qf = o_chunk + al64((size_t)n_chunk * sizeof(RefineChunk)),
	             o_words = o_qf + al64(((size_t)n_query + 1) * 8), up_bytes = o_words + al64((size_t)n_words * 4 + 16);
	int rc;
	if ((rc = B.h_meta.ensure(up_bytes + 64)) || (rc = B.r_win.ensure(up_bytes)) || (rc = B.r_hits.ensure((size_t)cap * 16)) || (rc = B.r_count.ensure(16)) ||
	    (rc = B.h_back.ensure(64))) return rc;
	char *hm = B.h_meta.as<char>();
	{
		RefineWindowDev *dw = (RefineWindowDev*)(hm + o_win);
		RefineChunk *ch = (RefineChunk*)(hm + o_chunk);
		int64_t c = 0;
		for (int64_t k = 0; k < n_win; ++k) {
			dw[k] = RefineWindowDev{ wins[k].as, wins[k].qid, wins[k].vid, wins[k].len, 0 };
			for (int32_t st = 0; st < wins[k].len; st += REFINE_CHUNK) ch[c++] = RefineChunk{ (int32_t)k, st };
		}
		memcpy(hm + o_qf, qw_first, ((size_t)n_query + 1) * 8);
		memcpy(hm + o_words, qwords, (size_t)n_words * 4);
	}
	HIP_TRY(hipMemcpyAsync(B.r_win.p, hm, up_bytes, hipMemcpyHostToDevice, s));
	HIP_TRY(hipMemsetAsync(B.r_count.p, 0, 16, s));
	RefineTab rt;
	for (int c = 0; c < 64; ++c) rt.t[c] = tab_codon()[c] >= 20 ? 0xff : tab_codon13()[c];
	DevGenome dg{ mi->dev[ctx->device]->seq, mi->dev[ctx->device]->ctg_off, mi->dev[ctx->device]->ctg_len, nullptr, mi->l_seq };
	const size_t lds = ((size_t)4 << hs_log2) + REFINE_CHUNK + 2 * REFINE_HALO;
	const char *dm = B.r_win.as<char>();
	hipLaunchKernelGGL(k_refine_scan, dim3((unsigned)n_chunk), dim3(256), lds, s, dg, (const RefineWindowDev*)(dm + o_win), (const RefineChunk*)(dm + o_chunk),
	                   (const int64_t*)(dm + o_qf), (const uint32_t*)(dm + o_words), rt, kmer, min_aa_len, hs_log2, B.r_hits.as<uint4>(), B.r_count.as<unsigned long long>(), cap);
	HIP_TRY(hipGetLastError());
	unsigned long long *h_n = B.h_back.as<unsigned long long>();
	HIP_TRY(hipMemcpyAsync(h_n, B.r_count.p, 8, hipMemcpyDeviceToHost, s));
	HIP_TRY(wait_stream(ctx, s));
	const unsigned long long n_hits = *h_n;
	if (n_hits > cap) { set_error("refinement scan: more hits than the buffer holds"); return MPA_ERR_UNSUPPORTED; }
	if (n_hits == 0) return MPA_OK;
	if ((rc = B.h_rhits.ensure((size_t)n_hits * 16)) != MPA_OK) return rc;
	HIP_TRY(hipMemcpyAsync(B.h_rhits.p, B.r_hits.p, (size_t)n_hits * 16, hipMemcpyDeviceToHost, s));
	HIP_TRY(wait_stream(ctx, s));
	// group by window (counting sort)
	const uint4 *h = B.h_rhits.as<uint4>();
	for (unsigned long long k = 0; k < n_hits; ++k) ++out.first[(size_t)h[k].x + 1];
	for (int64_t k = 0; k < n_win; ++k) out.first[(size_t)k + 1] += out.first[(size_t)k];
	out.hits.resize((size_t)n_hits);
	std::vector<int64_t> at(out.first.begin(), out.first.end() - 1);
	for (unsigned long long k = 0; k < n_hits; ++k) out.hits[(size_t)at[h[k].x]++] = (uint64_t)h[k].z << 32 | h[k].y;
	return MPA_OK;
}
} // namespace mpa

namespace mpa {
// mp_refine_reg (map.c:32-96) for all windows of a mini-batch on the device: see the kernels in seed_exec.hip ("Refinement
// pairing on the device") and k_chain_fwd / k_chain_fwd_wave / k_chain_extract.  MPA_ERR_UNSUPPORTED: outside the kernels' range
// (the caller refines on the host).
int dev_refine_chains(mpa_ctx_t *ctx, mpa_idx_s *mi, int32_t kmer, int32_t min_aa_len, int32_t max_ava, const ChainParams &cp, int32_t n_query, const RefineGroupsHost &G,
                      int64_t n_win, const RefineWindow *wins, RefineChains &out)
{
	out.u_first.assign((size_t)n_win + 1, 0), out.a_first.assign((size_t)n_win + 1, 0);
	out.U = out.A = nullptr;
	if (n_win == 0) return MPA_OK;
	if (kmer < 1 || kmer > 7 || min_aa_len > (REFINE_HALO - 3 * kmer) / 3 + kmer || cp.bbit != 0) { set_error("device refinement: parameters outside the kernels' range"); return MPA_ERR_UNSUPPORTED; }
	if (n_win >= (1 << 20)) { set_error("device refinement: more than 2^20 windows in a batch"); return MPA_ERR_UNSUPPORTED; }
	int64_t max_groups = 0;
	for (int32_t q = 0; q < n_query; ++q) max_groups = std::max(max_groups, G.qg_first[(size_t)q + 1] - G.qg_first[(size_t)q]);
	int hs_log2 = 10;
	while ((1LL << hs_log2) < 2 * max_groups) ++hs_log2;
	if (hs_log2 > 12) { set_error("device refinement: query too long for the LDS k-mer map"); return MPA_ERR_UNSUPPORTED; }
	for (uint32_t p : G.qpos) if (p >= (1u << 22)) { set_error("device refinement: a query longer than 2^22 residues"); return MPA_ERR_UNSUPPORTED; }
	HIP_TRY(hipSetDevice(ctx->device));
	if (dev_upload_index(ctx, mi) != MPA_OK) return MPA_ERR_HIP;
	SeedBufs &B = ctx->seed;
	ensure_seed_stream(ctx);
	hipStream_t s = ctx->seed_stream;
	const double t0 = now_ms();
	// ---- one pinned block up: windows | chunks | wg_first | qg_first | gword | gfirst | gcount | qpos
	static const int n_super = [] { const char *e = getenv("MPA_REFINE_SUPER"); const int v = e ? atoi(e) : REFINE_SUPER; return v < 1 ? 1 : v > 16 ? 16 : v; }();
	int64_t n_pos = 0, n_chunk = 0, wg_total = 0;
	for (int64_t k = 0; k < n_win; ++k) {
		if (wins[k].len >= (1 << 22)) { set_error("device refinement: a window longer than 2^22 bases"); return MPA_ERR_UNSUPPORTED; }
		n_pos += wins[k].len, n_chunk += (wins[k].len + n_super * REFINE_CHUNK - 1) / (n_super * REFINE_CHUNK);   // (a workgroup sweeps n_super chunks of its window)
	}
	if (n_chunk == 0) return MPA_OK;
	const unsigned long long cap = (unsigned long long)(n_pos / 64 + (1 << 20));
	const size_t n_group = G.gword.size(), n_qpos = G.qpos.size(), NW = (size_t)n_win, NQ = (size_t)n_query;
	auto al64 = [](size_t x) { return (x + 63) & ~(size_t)63; };
	const size_t o_win = 0, o_chunk = al64(NW * sizeof(RefineWindowDev)), o_wg = o_chunk + al64((size_t)n_chunk * sizeof(RefineChunk)), o_qg = o_wg + al64((NW + 1) * 8),
	             o_gw = o_qg + al64((NQ + 1) * 8), o_gf = o_gw + al64(n_group * 4 + 4), o_gc = o_gf + al64(n_group * 4 + 4), o_qp = o_gc + al64(n_group * 4 + 4),
	             up_bytes = o_qp + al64(n_qpos * 4 + 4);
	int rc;
	int64_t cls_end[3] = { 0, 0, 0 };                          // chunks of the windows whose query's map has 1 024 / 2 048 / 4 096 slots end here
	if ((rc = B.h_meta.ensure(up_bytes + 64)) || (rc = B.r_win.ensure(up_bytes)) || (rc = B.r_hits.ensure((size_t)cap * 16)) || (rc = B.r_count.ensure(16)) ||
	    (rc = B.h_back.ensure(256))) return rc;
	char *hm = B.h_meta.as<char>();
	{
		RefineWindowDev *dw = (RefineWindowDev*)(hm + o_win);
		RefineChunk *ch = (RefineChunk*)(hm + o_chunk);
		int64_t *wg = (int64_t*)(hm + o_wg);
		for (int64_t k = 0; k < n_win; ++k) {
			dw[k] = RefineWindowDev{ wins[k].as, wins[k].qid, wins[k].vid, wins[k].len, 0 };
			wg[k] = wg_total;
			wg_total += G.qg_first[(size_t)wins[k].qid + 1] - G.qg_first[(size_t)wins[k].qid];
		}
		wg[n_win] = wg_total;
		// the workgroups of a window, grouped by the size of its query's k-mer map (1 024 / 2 048 / 4 096 slots): one launch per size, so
		// that the windows of ordinary proteins take 13 KB of LDS per workgroup and not the 37 KB the batch's longest protein needs
		int64_t c = 0;
		for (int cls = 0; cls < 3; ++cls) {
			for (int64_t k = 0; k < n_win; ++k) {
				const int64_t ng = G.qg_first[(size_t)wins[k].qid + 1] - G.qg_first[(size_t)wins[k].qid];
				const int kc = 2 * ng <= 1024 ? 0 : 2 * ng <= 2048 ? 1 : 2;
				if (kc != cls) continue;
				for (int32_t st = 0; st < wins[k].len; st += n_super * REFINE_CHUNK) ch[c++] = RefineChunk{ (int32_t)k, st };
			}
			cls_end[cls] = c;
		}
		memcpy(hm + o_qg, G.qg_first.data(), (NQ + 1) * 8);
		if (n_group) memcpy(hm + o_gw, G.gword.data(), n_group * 4), memcpy(hm + o_gf, G.gfirst.data(), n_group * 4), memcpy(hm + o_gc, G.gcount.data(), n_group * 4);
		if (n_qpos) memcpy(hm + o_qp, G.qpos.data(), n_qpos * 4);
	}
	// device tables: per (window, group) hit counts and per-window pair counts, zeroed
	size_t at = 0;
	auto carve = [&](size_t bytes) { const size_t o = at; at += (bytes + 255) & ~(size_t)255; return o; };
	const size_t o_wcnt = carve(((size_t)wg_total + 1) * 4), o_wpairs = carve((NW + 2) * 4), o_first = carve((NW + 2) * 8);
	const size_t zero_bytes = at;
	if ((rc = B.rx_all.ensure(at))) return rc;
	HIP_TRY(hipMemcpyAsync(B.r_win.p, hm, up_bytes, hipMemcpyHostToDevice, s));
	HIP_TRY(hipMemsetAsync(B.r_count.p, 0, 16, s));
	HIP_TRY(hipMemsetAsync(B.rx_all.p, 0, zero_bytes, s));
	RefineTab rt;
	for (int c = 0; c < 64; ++c) rt.t[c] = tab_codon()[c] >= 20 ? 0xff : tab_codon13()[c];
	DevGenome dg{ mi->dev[ctx->device]->seq, mi->dev[ctx->device]->ctg_off, mi->dev[ctx->device]->ctg_len, nullptr, mi->l_seq };
	const char *dm = B.r_win.as<char>();
	RefineGroups gr{ (const int64_t*)(dm + o_qg), (const uint32_t*)(dm + o_gw), (const uint32_t*)(dm + o_gf), (const uint32_t*)(dm + o_gc), (const uint32_t*)(dm + o_qp) };
	const int64_t *d_wg = (const int64_t*)(dm + o_wg);
	char *R = B.rx_all.as<char>();
	uint32_t *d_wcnt = (uint32_t*)(R + o_wcnt), *d_wpairs = (uint32_t*)(R + o_wpairs);
	int64_t *d_first = (int64_t*)(R + o_first);
	HIP_TRY(ensure_dynamic_lds((const void*)k_refine_scan_map, ctx->device, 48 * 1024));
	for (int cls = 0; cls < 3; ++cls) {
		const int64_t c_first = cls ? cls_end[cls - 1] : 0, c_n = cls_end[cls] - c_first;
		if (c_n == 0) continue;
		const int hs = 10 + cls;
		const size_t lds = ((size_t)8 << hs) + 2 * (REFINE_CHUNK + 2 * REFINE_HALO);   // k-mer map, bases, codons
		hipLaunchKernelGGL(k_refine_scan_map, dim3((unsigned)c_n), dim3(256), lds, s, dg, (const RefineWindowDev*)(dm + o_win), (const RefineChunk*)(dm + o_chunk) + c_first, gr, d_wg, rt,
		                   kmer, min_aa_len, hs, B.r_hits.as<uint4>(), B.r_count.as<unsigned long long>(), cap, d_wcnt, (int32_t)n_super);
	}
	HIP_TRY(hipGetLastError());
	unsigned long long *h_n = B.h_back.as<unsigned long long>();
	HIP_TRY(hipMemcpyAsync(h_n, B.r_count.p, 8, hipMemcpyDeviceToHost, s));
	HIP_TRY(wait_stream(ctx, s));
	const int64_t n_hits = (int64_t)*h_n;
	if ((unsigned long long)n_hits > cap) { set_error("device refinement: more hits than the buffer holds"); return MPA_ERR_UNSUPPORTED; }
	timing_note("    refine: scan (wait)", now_ms() - t0);
	if (n_hits == 0) return MPA_OK;
	// ---- pairs: count, offsets, emit, sort, decode
	const double t1 = now_ms();
	if ((rc = B.r_chunk.ensure((size_t)n_hits * 4 + 16)) || (rc = B.r_words.ensure((size_t)n_hits * 8 + 16))) return rc;   // pairs per hit, and where they go
	uint32_t *d_pc = B.r_chunk.as<uint32_t>();
	uint64_t *d_po = B.r_words.as<uint64_t>();
	const unsigned nbh = (unsigned)((n_hits + 255) / 256);
	hipLaunchKernelGGL(k_refine_pair_count, dim3(nbh), dim3(256), 0, s, B.r_hits.as<uint4>(), n_hits, d_wg, d_wcnt, gr.gcount, max_ava, d_pc, d_wpairs);
	HIP_TRY(hipGetLastError());
	{
		size_t tb = 0, tb2 = 0;
		auto in = rocprim::make_transform_iterator((const uint32_t*)d_pc, U32ToU64());
		auto inw = rocprim::make_transform_iterator((const uint32_t*)d_wpairs, U32ToU64());
		HIP_TRY(rocprim::exclusive_scan(nullptr, tb, in, d_po, (uint64_t)0, (size_t)n_hits, rocprim::plus<uint64_t>(), s));
		HIP_TRY(rocprim::exclusive_scan(nullptr, tb2, inw, (uint64_t*)d_first, (uint64_t)0, NW + 1, rocprim::plus<uint64_t>(), s));
		if ((rc = B.tmp.ensure(std::max(tb, tb2) + 256))) return rc;
		HIP_TRY(rocprim::exclusive_scan(B.tmp.p, tb, in, d_po, (uint64_t)0, (size_t)n_hits, rocprim::plus<uint64_t>(), s));
		HIP_TRY(rocprim::exclusive_scan(B.tmp.p, tb2, inw, (uint64_t*)d_first, (uint64_t)0, NW + 1, rocprim::plus<uint64_t>(), s));
	}
	int64_t *h_np = (int64_t*)(h_n + 1);
	HIP_TRY(hipMemcpyAsync(h_np, d_first + n_win, 8, hipMemcpyDeviceToHost, s));
	HIP_TRY(wait_stream(ctx, s));
	const int64_t np = *h_np;
	if (np == 0) return MPA_OK;
	if ((rc = B.rx_keys.ensure((size_t)np * 24 + 64))) return rc;
	uint64_t *keys0 = B.rx_keys.as<uint64_t>(), *keys1 = keys0 + np, *d_a = keys1 + np;
	hipLaunchKernelGGL(k_refine_pair_emit, dim3(nbh), dim3(256), 0, s, B.r_hits.as<uint4>(), n_hits, (const uint32_t*)d_pc, (const uint64_t*)d_po, gr, keys0);
	HIP_TRY(hipGetLastError());
	{
		int wbits = 1;
		while ((1LL << wbits) < n_win) ++wbits;
		size_t tb = 0;
		HIP_TRY(rocprim::radix_sort_keys(nullptr, tb, keys0, keys1, (size_t)np, 0u, (unsigned)(44 + wbits), s));
		if ((rc = B.tmp.ensure(tb + 256))) return rc;
		HIP_TRY(rocprim::radix_sort_keys(B.tmp.p, tb, keys0, keys1, (size_t)np, 0u, (unsigned)(44 + wbits), s));
	}
	const unsigned nbp = (unsigned)((np + 255) / 256);
	hipLaunchKernelGGL(k_refine_pair_decode, dim3(nbp), dim3(256), 0, s, (const uint64_t*)keys1, np, d_a);
	HIP_TRY(hipGetLastError());
	// ---- the chains of every window: forward pass (base resolution), extraction, pack
	PreParams pm;
	pm.max_dist_x = std::max(cp.max_dist_x, cp.bw), pm.max_dist_y = cp.max_dist_y;
	if (pm.max_dist_y < cp.bw && !cp.is_spliced) pm.max_dist_y = cp.bw;
	pm.bw = cp.bw, pm.max_skip = cp.max_skip, pm.max_iter = cp.max_iter, pm.kmer = cp.kmer, pm.bbit = cp.bbit;
	pm.is_spliced = cp.is_spliced, pm.coef_log = cp.coef_log, pm.max_dblock = pm.max_dist_x >> cp.bbit;
	const size_t M = (size_t)np;
	size_t xat = 0;
	auto xcarve = [&](size_t bytes) { const size_t o = xat; xat += (bytes + 255) & ~(size_t)255; return o; };
	const size_t x_mark = xcarve(M * 4), x_order = xcarve(M * 4), x_ends = xcarve((M + 64 * NW + 64) * sizeof(Pair64)), x_tail8 = xcarve(M * sizeof(Pair64)),
	             x_items = xcarve(M * sizeof(SparseItem)), x_moved = xcarve(M * sizeof(SparseItem)), x_merged = xcarve(M * sizeof(SparseItem)), x_kept = xcarve(M), x_stack = xcarve((M / 64 + 6 * NW + 16) * sizeof(SortRange)), x_status = xcarve(NW * 4 + 16), x_f = xcarve(M * 4), x_pred = xcarve(M * 4),
	             x_fm = xcarve(M * 4), x_outa = xcarve(M * 8), x_outu = xcarve(M * 8), x_na = xcarve(NW * 8 + 8), x_nu = xcarve(NW * 8 + 8), x_offa = xcarve(NW * 8 + 16), x_offu = xcarve(NW * 8 + 16);
	const int32_t kSerialRun = 48;
	const size_t long_cap = M / (size_t)(kSerialRun + 1) + 16, x_long = xcarve(long_cap * sizeof(LongRun)), x_nlong = xcarve(64);
	if ((rc = B.x_all.ensure(xat))) return rc;
	char *X = B.x_all.as<char>();
	HIP_TRY(hipMemsetAsync(X + x_status, 0, NW * 4 + 16, s));
	HIP_TRY(hipMemsetAsync(X + x_nlong, 0, 64, s));
	hipLaunchKernelGGL(k_seed_fill, dim3(nbp), dim3(256), 0, s, np, pm.kmer, (int32_t*)(X + x_f), (int32_t*)(X + x_pred), (int32_t*)(X + x_fm), (uint32_t*)(X + x_mark));
	hipLaunchKernelGGL(k_chain_fwd, dim3(nbp), dim3(256), 0, s, (const uint64_t*)d_a, np, (const int64_t*)d_first, (const int64_t*)nullptr, (int32_t)n_win, pm, (int32_t*)(X + x_f), (int32_t*)(X + x_pred),
	                   (int32_t*)(X + x_fm), kSerialRun, (LongRun*)(X + x_long), (unsigned int*)(X + x_nlong), (unsigned int)long_cap);
	hipLaunchKernelGGL(k_chain_fwd_wave, dim3((unsigned)std::min<size_t>(long_cap, 65536)), dim3(64), 0, s, (const uint64_t*)d_a, (const LongRun*)(X + x_long), (const unsigned int*)(X + x_nlong),
	                   (unsigned int)long_cap, pm, (int32_t*)(X + x_f), (int32_t*)(X + x_pred), (int32_t*)(X + x_fm));
	HIP_TRY(hipGetLastError());
	ExtractArgs xa;
	xa.first = d_first, xa.cnt = nullptr, xa.ntot_first = nullptr;
	xa.v_pos = nullptr, xa.v_f = (const int32_t*)(X + x_f), xa.v_pred = (const int32_t*)(X + x_pred), xa.v_a = (const uint64_t*)d_a;
	xa.mark = (int32_t*)(X + x_mark), xa.order = (int32_t*)(X + x_order), xa.ends = (Pair64*)(X + x_ends), xa.tail8 = (Pair64*)(X + x_tail8);
	xa.items = (SparseItem*)(X + x_items), xa.moved = (SparseItem*)(X + x_moved), xa.merged = (SparseItem*)(X + x_merged);
	xa.kept = (uint8_t*)(X + x_kept), xa.stack = (SortRange*)(X + x_stack);
	xa.a_out = (uint64_t*)(X + x_outa), xa.u_out = (uint64_t*)(X + x_outu), xa.n_a = (int64_t*)(X + x_na), xa.n_u = (int64_t*)(X + x_nu);
	xa.status = (int32_t*)(X + x_status), xa.p = cp, xa.set_only = 0;
	hipLaunchKernelGGL(k_chain_extract, dim3((unsigned)n_win), dim3(64), EXTRACT_LDS_BYTES, s, xa, (int32_t)n_win);
	hipLaunchKernelGGL(k_offsets2, dim3(1), dim3(256), 0, s, (const int64_t*)(X + x_na), (const int64_t*)(X + x_nu), (int32_t)n_win, (int64_t*)(X + x_offa), (int64_t*)(X + x_offu));
	HIP_TRY(hipGetLastError());
	const size_t offb = (NW + 1) * 8;
	if ((rc = B.h_xoff.ensure(2 * offb + NW * 4 + 64))) return rc;
	int64_t *h_offa = B.h_xoff.as<int64_t>(), *h_offu = h_offa + (NW + 1);
	int32_t *h_status = (int32_t*)(h_offu + (NW + 1));
	HIP_TRY(hipMemcpyAsync(h_offa, X + x_offa, offb, hipMemcpyDeviceToHost, s));
	HIP_TRY(hipMemcpyAsync(h_offu, X + x_offu, offb, hipMemcpyDeviceToHost, s));
	HIP_TRY(hipMemcpyAsync(h_status, X + x_status, NW * 4, hipMemcpyDeviceToHost, s));
	HIP_TRY(wait_stream(ctx, s));
	for (size_t w = 0; w < NW; ++w) if (h_status[w]) { set_error("device refinement: a chain extraction needs the host"); return MPA_ERR_UNSUPPORTED; }   // (dense views never do)
	const int64_t tot_a = h_offa[n_win], tot_u = h_offu[n_win];
	SeedHold &H = B.own;
	if ((rc = H.h_A.ensure((size_t)tot_a * 8 + 64)) || (rc = H.h_U.ensure((size_t)tot_u * 8 + 64))) return rc;
	if (tot_a > 0 || tot_u > 0) {
		hipLaunchKernelGGL(k_chain_pack, dim3((unsigned)n_win), dim3(256), 0, s, (const int64_t*)d_first, (const int64_t*)(X + x_na), (const int64_t*)(X + x_nu), (const int64_t*)(X + x_offa),
		                   (const int64_t*)(X + x_offu), (const uint64_t*)(X + x_outa), (const uint64_t*)(X + x_outu), H.h_A.as<uint64_t>(), H.h_U.as<uint64_t>());
		HIP_TRY(hipGetLastError());
		HIP_TRY(wait_stream(ctx, s));
	}
	out.a_first.assign(h_offa, h_offa + n_win + 1), out.u_first.assign(h_offu, h_offu + n_win + 1);
	out.A = H.h_A.as<uint64_t>(), out.U = H.h_U.as<uint64_t>();
	timing_note("    refine: pairs + chains (wait)", now_ms() - t1);
	return MPA_OK;
}
} // namespace mpa

extern "C" {

} // extern "C" (reopened below)

namespace mpa {
// ---- DP worker pool: host side (the kernels and the protocol are in dp_kernels.hip, "The DP worker pool")
// MPA_DP_POOL=1 selects the pool; the default is one k_dp_round launch per round.  Measured (profiles/r05_experiments.txt): with
// identical sweep code the pool is level with the launches on the DP-bound config 5 (1.05-1.14 M against 1.09 M residues/s), 5-10 %
// behind on config 3 (19.4-19.8 M against 20.0-21.9 M) and a third behind on config 2 (17-ms batches: arming, the second host
// round trip and the L2 write-back weigh).  The hardware's workgroup dispatcher already IS a work-conserving queue across the
// launches in flight -- a slot that a finished unit frees goes to the oldest pending workgroup of ANY launch -- so what the pool
// adds (units taken wave by wave, a bounded DP population) buys no throughput here, and its round latency is longer because the
// rounds in flight share the workers instead of being served in order of arrival.
static bool dp_pool_enabled()
{
	static const bool on = [] { const char *e = getenv("MPA_DP_POOL"); return e && atoi(e) != 0; }();
	return on;
}
// resident worker workgroups the pool admits (MPA_DP_WORKERS).  The workers ask for a fifth of a CU's LDS plus a little, so four
// fit per CU (1 024 on the chip); the default keeps three per CU busy and leaves the fourth slot to the workgroups that arrive,
// find the pool full and leave -- and registers and LDS to the seeding kernels of the batches behind.
static int dp_pool_budget()
{
	static const int b = [] { const char *e = getenv("MPA_DP_WORKERS"); const int v = e ? atoi(e) : 768; return v < 1 ? 1 : v; }();
	return b;
}
// MPA_DP_TRACE=<file>: every unit of every round appends "slot generation unit kind priority start end" (100-MHz device ticks);
// tools/dp_trace.py turns that into resident units over time
static const char *dp_trace_path()
{
	static const char *p = [] { const char *e = getenv("MPA_DP_TRACE"); return e && *e ? e : (const char*)nullptr; }();
	return p;
}
// the calling context's slot, done word and worker stream (created on its first round); the device's pool on the first of all
static int pool_attach(mpa_ctx_t *ctx)
{
	mpa_ctx_s *root = ctx->root ? ctx->root : ctx;
	std::lock_guard<std::mutex> g(root->pool_mu);
	if (!root->dp_pool) {
		DpPool *p = nullptr;
		HIP_TRY(hipMalloc((void**)&p, sizeof(DpPool)));
		HIP_TRY(hipMemset(p, 0, sizeof(DpPool)));
		const int32_t budget = dp_pool_budget();
		HIP_TRY(hipMemcpy(&p->ctl.budget, &budget, 4, hipMemcpyHostToDevice));
		const int32_t acq = [] { const char *e = getenv("MPA_DP_ACQUIRE"); return e ? atoi(e) : 2; }();
		HIP_TRY(hipMemcpy(&p->ctl.acquire_mode, &acq, 4, hipMemcpyHostToDevice));
		HIP_TRY(hipEventCreate(&root->pool_base));
		HIP_TRY(hipEventRecord(root->pool_base, root->stream));
		root->dp_pool = p;
	}
	if (ctx->dp_slot < 0) {
		if (root->pool_slots >= MPA_DP_SLOTS) { set_error("more than " + std::to_string(MPA_DP_SLOTS) + " contexts of one device run DP rounds"); return MPA_ERR_UNSUPPORTED; }
		HIP_TRY(hipHostMalloc((void**)&ctx->dp_done, 64, hipHostMallocDefault));
		*(volatile int32_t*)ctx->dp_done = 0;
		HIP_TRY(hipStreamCreateWithFlags(&ctx->worker_stream, hipStreamNonBlocking));
		HIP_TRY(hipEventCreateWithFlags(&ctx->arm_ev, hipEventDisableTiming));
		ctx->dp_slot = root->pool_slots++;
	}
	return MPA_OK;
}
// durations of the context's worker launches that have ended (wait: of all of them -- only when no round is pending anywhere, the
// workers then leave within microseconds) into the context's totals and the device's interval list
void pool_harvest(mpa_ctx_t *ctx, bool wait)
{
	if (ctx->wl_busy.empty()) return;
	mpa_ctx_s *root = ctx->root ? ctx->root : ctx;
	(void)hipSetDevice(ctx->device);
	if (wait) (void)wait_stream(ctx, ctx->worker_stream);
	size_t keep = 0;
	for (size_t k = 0; k < ctx->wl_busy.size(); ++k) {
		mpa_ctx_s::WorkerLaunch w = ctx->wl_busy[k];
		float a = 0, b = 0;
		if (hipEventQuery(w.e1) == hipSuccess && hipEventElapsedTime(&a, root->pool_base, w.e0) == hipSuccess && hipEventElapsedTime(&b, root->pool_base, w.e1) == hipSuccess) {
			ctx->total.ms_round += (double)(b - a), ctx->total.launches_round++;
			{ std::lock_guard<std::mutex> g(root->pool_mu); root->pool_iv.emplace_back(a, b); }
			ctx->wl_free.push_back(w);
		} else ctx->wl_busy[keep++] = w;
	}
	(void)hipGetLastError();
	ctx->wl_busy.resize(keep);
}
// time during which at least one worker launch of the device was running
static double pool_union_ms(mpa_ctx_s *root, bool reset)
{
	std::lock_guard<std::mutex> g(root->pool_mu);
	std::vector<std::pair<float, float>> iv = root->pool_iv;
	if (reset) root->pool_iv.clear();
	std::sort(iv.begin(), iv.end());
	double sum = 0, lo = 0, hi = -1;
	for (auto &x : iv) {
		if (hi < 0) { lo = x.first, hi = x.second; continue; }
		if (x.first <= hi) { hi = std::max<double>(hi, x.second); continue; }
		sum += hi - lo, lo = x.first, hi = x.second;
	}
	if (hi >= 0) sum += hi - lo;
	return sum;
}
} // namespace mpa

extern "C" {

void mpa_dp_last_stats(const mpa_ctx_t *ctx, mpa_dp_stats_t *st) { *st = ctx->stats; }
void mpa_dp_total_stats(mpa_ctx_t *ctx, mpa_dp_stats_t *st, int reset)
{
	pool_harvest(ctx, true);
	mpa_ctx_s *root = ctx->root ? ctx->root : ctx;
	const double u = pool_union_ms(root, reset != 0);
	if (st) *st = ctx->total, st->ms_round_union = u;
	if (reset) ctx->total = mpa_dp_stats_t();
}

#define MPA_RETRY_NO_SPLIT (-100)   /* internal: repeat the round without split extension calls */
static int mpa_dp_run_impl(mpa_ctx_t *ctx, const mpa_idx_t *mi, const mpa_dpopt_t *opt, const mpa_qbatch_t *q,
               int64_t n, const mpa_dp_task_t *in, mpa_dp_rst_t *rst, uint32_t **cigar_pool, int64_t *n_pool)
{
	if (cigar_pool) *cigar_pool = nullptr;
	if (n_pool) *n_pool = 0;
	if (!ctx) { set_error("no device context"); return MPA_ERR_NO_DEVICE; }
	if (ctx->device >= mpa_idx_s::kMaxDevices || !mi->dev[ctx->device]) { set_error("index is not resident on this context's device (call mpa_idx_to_device)"); return MPA_ERR_ARG; }
	ctx->stats = mpa_dp_stats_t();
	if (n <= 0) return MPA_OK;
	HIP_TRY(hipSetDevice(ctx->device));
	hipStream_t s = ctx->stream;
	const double t_begin = now_ms();

	// ---- parameter guards: outside these the packed-int16 kernels would not be bit-exact
	int32_t max_mat = 0;
	for (int k = 0; k < 484; ++k) max_mat = std::max<int32_t>(max_mat, opt->mat[k]);
	if (opt->go < 0 || opt->go > 32000 || opt->ge < 0 || opt->ge > 16000 || opt->fs < 0 || opt->fs > 16000 || opt->xdrop < 0 || opt->xdrop > 32000 ||
	    opt->end_bonus < 0 || opt->end_bonus > 1000 || (opt->ge > 255 && opt->go + opt->ge > 32000)) {
		set_error("DP parameters outside the supported range (go <= 32000, ge, fs <= 16000, 0 <= xdrop <= 32000, 0 <= end_bonus <= 1000)"); return MPA_ERR_UNSUPPORTED;
	}
	// Gap-extension / frameshift penalties above 255 (-E / -F of the reference's command line, main.c:133,136) do not fit the byte
	// the row records give them.  Such a run keeps the records' layout -- the byte then flags a stop codon -- and sweeps every call
	// with the kernels that read it that way (glob_cands<K, true>): the stand-alone traceback kernels and, for extension calls,
	// the block-major one-wave sweep (k_ext_huge); the packed round kernel is not used.  Slow, exact, and nobody's default.
	const bool wide_ge = opt->ge > 255 || opt->fs > 255;

	// ---- classify, order and lay out the calls
	std::vector<DTask> T(n);
	std::vector<int32_t> ext_ids, glob_ids;
	int32_t max_nl_ext = 0;
	for (int64_t k = 0; k < n; ++k) {
		const mpa_dp_task_t &x = in[k];
		DTask &t = T[k];
		memset(&t, 0, sizeof(t));
		if (x.nl < 0 || x.al <= 0 || x.qid < 0 || x.qid >= q->n_seq || x.io < 0 || x.io > 32000) { set_error("malformed DP task"); return MPA_ERR_ARG; }
		// the kernels address the resident genome and the query buffer with these: a window or a protein slice that leaves its
		// contig / its query would read foreign memory (or fault the context), so it is refused here
		if (x.vid < 0 || x.vid >= 2 * (int32_t)mi->ctg.size() || x.nt_off < 0 || x.nt_off + (int64_t)x.nl > mi->ctg[x.vid >> 1].len ||
		    x.aa_off < 0 || (int64_t)x.aa_off + x.al > q->q_off[x.qid + 1] - q->q_off[x.qid]) {
			set_error("DP task " + std::to_string(k) + " reaches outside its contig or its query"); return MPA_ERR_ARG;
		}
		t.nt_off = x.nt_off, t.vid = x.vid, t.nl = x.nl, t.al = x.al, t.flag = x.flag, t.io = x.io;
		t.q_off = q->q_off[x.qid] + x.aa_off - q->q_off[0];     // relative to the slice uploaded below
		t.ncol = (x.al + 7) / 8 * 8;
		t.out_idx = (int32_t)k;
		// (the int32 sweeps keep the striped reference's lane segments apart by offsets of 2^20 in their scans: column * ge must stay below)
		// (checked below where it matters: the traceback sweeps and the block-major extension sweep; the packed extension kernels
		// have their own bound, may_saturate)
		const bool seg_overflow = (int64_t)t.ncol * opt->ge >= (1 << 19);
		const bool is_ext = (x.flag & (MPA_F_EXT_LEFT | MPA_F_EXT_RIGHT)) != 0;
		if (is_ext) {
			int cls = ext_class_of(t.ncol);                        // -1: wider than k_ext_wide<16> covers -> k_ext_huge (class 7)
			if (ctx->no_split && cls >= 5) cls = -1;               // repeated round: no inter-workgroup hand-off (see mpa_dp_run)
			if (wide_ge) cls = -1;
			if (cls < 0 && seg_overflow) { set_error("DP call " + std::to_string(k) + ": too wide for this gap-extension penalty (columns x ge must stay below 2^19 in the int32 sweeps)"); return MPA_ERR_UNSUPPORTED; }
			// The packed kernels run their gap scan on h + j*ge with saturating adds, which is only the reference's value while
			// nothing can reach the int16 limits; calls that could (more than ~2900 columns with BLOSUM62, hence always of the
			// "huge" class) are swept by k_ext_huge, whose int32 arithmetic clamps every operation like the reference does.
			const bool may_saturate = (int64_t)x.al * max_mat + (int64_t)t.ncol * opt->ge + std::max(0, opt->end_bonus) > 32000 || opt->go + (int64_t)t.ncol * opt->ge > 32000;
			if (may_saturate && cls >= 0) { set_error("extension call could saturate int16 in the packed kernels; not supported with these scores"); return MPA_ERR_UNSUPPORTED; }
			t.pw = cls >= 0 ? kExtClasses[cls].G * kExtClasses[cls].NB : t.ncol;
			t.pad_ = cls >= 0 ? cls : 7;
			// 65..128 columns: one wave per call, column c + 64 in the high half of lane c (ext_narrow<64, false, true>, class 8), instead
			// of a two-wave group per pair of calls (MPA_DP_EXT_DUAL=0: the two-wave groups of rounds 2-5)
			static const bool ext_dual = [] { const char *e = getenv("MPA_DP_EXT_DUAL"); return !e || atoi(e) != 0; }();
			if (cls == 3 && ext_dual && !ctx->antidiag) t.pad_ = 8;
			ext_ids.push_back((int32_t)k);
			max_nl_ext = std::max(max_nl_ext, x.nl);
		} else {
			if (!(x.flag & MPA_F_CIGAR)) { set_error("global DP without CIGAR is not part of miniprot's path"); return MPA_ERR_UNSUPPORTED; }
			if (seg_overflow) { set_error("DP call " + std::to_string(k) + ": too wide for this gap-extension penalty (columns x ge must stay below 2^19 in the int32 sweeps)"); return MPA_ERR_UNSUPPORTED; }
			t.pw = t.ncol;
			t.pad_ = t.ncol <= 16 ? 0 : t.ncol <= 32 ? 1 : t.ncol <= 64 ? 2 : t.ncol <= 128 ? 3 : t.ncol <= 256 ? 4 : t.ncol <= 512 ? 5 : t.ncol <= 1024 ? 6 : 7;
			// Checkpointed traceback (dp_device.h): a call of up to 64 columns and many rows -- the gap fills across introns and the spans
			// of accepted extensions, where nearly every row lies inside an intron -- is swept by the packed sweep (classes 8, 9, 10: 16,
			// 32, 64 lanes per call, two calls per lane) and walked by k_walk.  Short calls stay on the plain traceback sweep: the walk
			// would recompute all of their rows anyway.  MPA_DP_LITE_MIN (rows; 0: never).
			static const int lite_min = [] { const char *e = getenv("MPA_DP_LITE_MIN"); return e ? atoi(e) : 384; }();
			if (lite_min > 0 && !wide_ge && t.ncol <= 128 && x.nl >= lite_min && x.nl >= 3) t.pad_ += 8, t.pw = 16 << (t.pad_ - 8);   // (class 11: 65..128 columns, one call per wave)
			glob_ids.push_back((int32_t)k);
		}
	}
	auto by_class_then_len = [&](int32_t a, int32_t b) {
		if (T[a].pad_ != T[b].pad_) return T[a].pad_ < T[b].pad_;
		if (T[a].nl != T[b].nl) return T[a].nl > T[b].nl;
		return a < b;
	};
	std::sort(ext_ids.begin(), ext_ids.end(), by_class_then_len);
	std::sort(glob_ids.begin(), glob_ids.end(), by_class_then_len);
	int64_t rec_total = 0, prof_total = 0, cig_total = 0, bnd_total = 0;
	int32_t max_nl = 0;
	std::vector<PrepChunk> chunks;
	auto layout = [&](int32_t id) {
		DTask &t = T[id];
		t.rec_off = rec_total, rec_total += t.nl;
		t.prof_off = prof_total, prof_total += (int64_t)22 * t.pw;
		max_nl = std::max(max_nl, t.nl);
		for (int32_t r = 0; r < t.nl; r += MPA_PREP_CHUNK_ROWS) chunks.push_back(PrepChunk{ id, r });
	};
	int64_t hkey_total = 0;                                 // per-row keys (8 B) of the extension calls wider than 1024 columns
	std::vector<int32_t> huge_ids;
	for (int32_t id : ext_ids) {
		layout(id);
		DTask &t = T[id];
		if (t.pad_ == 7) t.bnd_off = bnd_total, bnd_total += t.nl, t.tb_off = hkey_total, hkey_total += t.nl, huge_ids.push_back(id);
	}
	for (int32_t id : glob_ids) {
		layout(id);
		DTask &t = T[id];
		t.cig_cap = t.nl + t.al + 4;
		t.cig_off = cig_total, cig_total += t.cig_cap;
		if (t.pad_ == 7) t.bnd_off = bnd_total, bnd_total += t.nl;
	}
	rec_total += max_nl + 96 + (ctx->antidiag ? 64 : 0);    // kernels prefetch records up to 48 rows past a call's end (the anti-diagonal prototype: 128)

	// ---- waves of the extension kernel
	std::vector<ExtWave> ewaves;
	std::vector<int> ewave_cnt(kNumExtClasses, 0), ewave_first(kNumExtClasses, 0);
	{
		size_t p = 0;
		for (int cls = 0; cls < kNumExtClasses; ++cls) {
			const int slots = 2 * (64 / kExtClasses[cls].G);
			ewave_first[cls] = (int)ewaves.size();
			while (p < ext_ids.size() && T[ext_ids[p]].pad_ == cls) {
				ExtWave w;
				memset(&w, 0, sizeof(w));
				for (int k = 0; k < 8; ++k) w.task[k] = -1;
				w.rec_base = T[ext_ids[p]].rec_off;
				for (int k = 0; k < slots && p < ext_ids.size() && T[ext_ids[p]].pad_ == cls; ++k, ++p) {
					w.task[k] = ext_ids[p];
					w.max_nl = std::max(w.max_nl, T[ext_ids[p]].nl);
				}
				ewaves.push_back(w);
			}
			ewave_cnt[cls] = (int)ewaves.size() - ewave_first[cls];
		}
	}
	// ---- one-call waves of the 65..128-column extension class (8: sorts behind the huge class 7)
	int dwave_first = (int)ewaves.size(), dwave_cnt = 0;
	for (size_t p = 0; p < ext_ids.size(); ++p) {
		if (T[ext_ids[p]].pad_ != 8) continue;
		ExtWave w;
		memset(&w, 0, sizeof(w));
		for (int k = 0; k < 8; ++k) w.task[k] = -1;
		w.task[0] = ext_ids[p], w.rec_base = T[ext_ids[p]].rec_off, w.max_nl = T[ext_ids[p]].nl;
		ewaves.push_back(w), ++dwave_cnt;
	}
	// ---- waves of the checkpointed traceback's packed sweep (classes 8, 9, 10 sort behind every other traceback call)
	size_t n_reg_glob = 0;
	while (n_reg_glob < glob_ids.size() && T[glob_ids[n_reg_glob]].pad_ < 8) ++n_reg_glob;
	int lwave_first[4] = { 0, 0, 0, 0 }, lwave_cnt[4] = { 0, 0, 0, 0 };
	int64_t lite_total = 0, ck_total = 0;
	{
		size_t p = n_reg_glob;
		for (int cls = 0; cls < 4; ++cls) {
			const int slots = cls == 3 ? 1 : 2 * (64 / (16 << cls));
			lwave_first[cls] = (int)ewaves.size();
			while (p < glob_ids.size() && T[glob_ids[p]].pad_ == 8 + cls) {
				ExtWave w;
				memset(&w, 0, sizeof(w));
				for (int k = 0; k < 8; ++k) w.task[k] = -1;
				w.rec_base = T[glob_ids[p]].rec_off;
				const size_t p0 = p;
				for (int k = 0; k < slots && p < glob_ids.size() && T[glob_ids[p]].pad_ == 8 + cls; ++k, ++p) {
					w.task[k] = glob_ids[p];
					w.max_nl = std::max(w.max_nl, T[glob_ids[p]].nl);
					T[glob_ids[p]].flag |= k << MPA_LITE_SLOT_SHIFT;
				}
				w.lite_off = lite_total, lite_total += ((int64_t)w.max_nl / 3 + 2) * 64;
				w.ck_off = ck_total, ck_total += (int64_t)std::max(0, (w.max_nl - 3) / MPA_TB_BLOCK) * 9 * 64;
				for (size_t q2 = p0; q2 < p; ++q2) T[glob_ids[q2]].tb_off = w.lite_off, T[glob_ids[q2]].bnd_off = w.ck_off;
				ewaves.push_back(w);
			}
			lwave_cnt[cls] = (int)ewaves.size() - lwave_first[cls];
		}
	}
	const size_t n_lite = glob_ids.size() - n_reg_glob;
	// per-row keys of the wide extension kernels: [group][2 halves][key_stride]
	int64_t key_stride = 0, n_wide_groups = 0;
	for (int cls = 3; cls < kNumExtClasses; ++cls)
		for (int k = 0; k < ewave_cnt[cls]; ++k) key_stride = std::max<int64_t>(key_stride, ewaves[ewave_first[cls] + k].max_nl), ++n_wide_groups;
	key_stride = (key_stride + 64) & ~(int64_t)63;
	PenTable pen;
	if (build_pen_table(opt->ie_coef, std::max(max_nl_ext, 2), &pen) < 0) { set_error("ie_coef produces too many penalty steps"); return MPA_ERR_UNSUPPORTED; }

	// ---- rounds of the traceback kernel, bounded by traceback memory
	struct GlobRound { size_t first, last; int64_t tb_words; };
	std::vector<GlobRound> rounds;
	{
		size_t p = 0;
		while (p < n_reg_glob) {
			GlobRound r{ p, p, 0 };
			while (r.last < n_reg_glob) {
				DTask &t = T[glob_ids[r.last]];
				int64_t wds = (int64_t)t.nl * t.ncol;
				if (r.last > r.first && (size_t)(r.tb_words + wds) * 2 > ctx->tb_budget) break;
				t.tb_off = r.tb_words, r.tb_words += wds, ++r.last;
			}
			rounds.push_back(r);
			p = r.last;
		}
	}
	int64_t tb_max = 0;
	for (auto &r : rounds) tb_max = std::max(tb_max, r.tb_words);

	timing_note("  dp: classify/sort/layout", now_ms() - t_begin);
	double t_mark = now_ms();
	auto mark = [&](const char *what) { const double t = now_ms(); timing_note(what, t - t_mark); t_mark = t; };   // (MPA_TIMING: wall clock between marks)
	// ---- device memory
	const int64_t q_bytes = q->q_off[q->n_seq] - q->q_off[0];
	int rc;
	if ((rc = ctx->tasks.ensure(sizeof(DTask) * n)) || (rc = ctx->chunks.ensure(sizeof(PrepChunk) * (chunks.size() + 1))) ||
	    (rc = ctx->qseq.ensure(q_bytes + 16)) || (rc = ctx->rec.ensure((size_t)rec_total * 4)) || (rc = ctx->prof.ensure((size_t)prof_total * 2 + 16)) ||
	    (rc = ctx->waves.ensure(sizeof(ExtWave) * (ewaves.size() + 1))) || (rc = ctx->extout.ensure(sizeof(ExtOut) * n)) ||
	    (rc = ctx->tb.ensure((size_t)tb_max * 2 + 16)) || (rc = ctx->cig.ensure((size_t)cig_total * 4 + 16)) || (rc = ctx->ncig.ensure(n * 4)) ||
	    (rc = ctx->lite.ensure((size_t)lite_total * 4 + 256)) || (rc = ctx->ckpt.ensure((size_t)ck_total * 4 + 256)) || (rc = ctx->wlist.ensure(n_lite * 4 + 128)) ||
	    (rc = ctx->score.ensure(n * 4)) || (rc = ctx->rowkey.ensure((size_t)(n_wide_groups * 2 * key_stride * 4 + 64))) || (rc = ctx->bnd.ensure((size_t)bnd_total * 16 + 16)) || (rc = ctx->hkey.ensure((size_t)hkey_total * 8 + 16 + (sizeof(GlobWave) + 4) * (huge_ids.size() + 1))) || (rc = ctx->list.ensure(n * 4 + 128 + sizeof(GlobWave) * (glob_ids.size() + 1))))
		return rc;
	// Everything the device needs from the host goes through ONE pinned staging buffer (sections below), so that no copy is
	// staged by the runtime and the host never waits for one: a DP round is enqueued in one go and waited for once.
	auto al256 = [](size_t x) { return (x + 255) & ~(size_t)255; };
	const size_t up_tasks = 0, up_chunks = up_tasks + al256(sizeof(DTask) * n), up_q = up_chunks + al256(sizeof(PrepChunk) * chunks.size()),
	             up_waves = up_q + al256((size_t)q_bytes), up_list = up_waves + al256(sizeof(ExtWave) * ewaves.size()),
	             up_gw = up_list + al256(4 * glob_ids.size()), up_units = up_gw + al256(sizeof(GlobWave) * (glob_ids.size() + 8)),
	             up_off = up_units + al256(sizeof(DpUnit) * (4 * ewaves.size() + glob_ids.size() + 64)),
	             up_ids = up_off + al256(8 * glob_ids.size()), up_args = up_ids + al256(4 * glob_ids.size()), up_wl = up_args + al256(sizeof(DpRoundArgs)),
	             up_end = up_wl + al256(4 * n_lite);
	if ((rc = ctx->h_up.ensure(up_end + 256))) return rc;
	char *hup = ctx->h_up.as<char>();
	memcpy(hup + up_tasks, T.data(), sizeof(DTask) * n);
	memcpy(hup + up_chunks, chunks.data(), sizeof(PrepChunk) * chunks.size());
	memcpy(hup + up_q, q->seqs + q->q_off[0], (size_t)q_bytes);
	memcpy(hup + up_waves, ewaves.data(), sizeof(ExtWave) * ewaves.size());
	mark("    dp: buffers");
	HIP_TRY(hipMemcpyAsync(ctx->tasks.p, hup + up_tasks, sizeof(DTask) * n, hipMemcpyHostToDevice, s));
	HIP_TRY(hipMemcpyAsync(ctx->chunks.p, hup + up_chunks, sizeof(PrepChunk) * chunks.size(), hipMemcpyHostToDevice, s));
	HIP_TRY(hipMemcpyAsync(ctx->qseq.p, hup + up_q, q_bytes, hipMemcpyHostToDevice, s));
	if (!ewaves.empty()) HIP_TRY(hipMemcpyAsync(ctx->waves.p, hup + up_waves, sizeof(ExtWave) * ewaves.size(), hipMemcpyHostToDevice, s));
	// (k_prep_rows writes every row of every call; only the padding the kernels prefetch behind the last call is cleared)
	{ const int64_t pad = max_nl + 96 + (ctx->antidiag ? 64 : 0); HIP_TRY(hipMemsetAsync((char*)ctx->rec.p + (size_t)(rec_total - pad) * 4, 0, (size_t)pad * 4, s)); }
	if (n_wide_groups) HIP_TRY(hipMemsetAsync(ctx->rowkey.p, 0, (size_t)(n_wide_groups * 2 * key_stride * 4), s));
	// split classes: boundary granules (16 B per row and boundary: 3 boundaries per 1024-column group, 1 per 512-column group),
	// then the per-group completion counters and the error flag; all zero before the launch (a granule's tag is row + 1)
	const int64_t n_split = ewave_cnt[5] + ewave_cnt[6], n_bound = 3 * (int64_t)ewave_cnt[6] + ewave_cnt[5];
	const size_t xg_bytes = (size_t)n_bound * key_stride * 16, xg_tail = (2 * (size_t)n_split + 1) * 4;   // + done[n_split], ticket[n_split], err
	if (n_split) {
		if ((rc = ctx->xg.ensure(xg_bytes + xg_tail + 64))) return rc;
		HIP_TRY(hipMemsetAsync(ctx->xg.p, 0, xg_bytes + xg_tail, s));
	}
	// extension calls wider than 1024 columns: keys (zeroed), then one GlobWave and one list entry per call
	GlobWave *d_hw = nullptr;
	int32_t *d_hlist = nullptr;
	if (!huge_ids.empty()) {
		HIP_TRY(hipMemsetAsync(ctx->hkey.p, 0, (size_t)hkey_total * 8, s));
		d_hw = (GlobWave*)((char*)ctx->hkey.p + (((size_t)hkey_total * 8 + 15) & ~(size_t)15));
		d_hlist = (int32_t*)(d_hw + huge_ids.size());
		std::vector<GlobWave> hw(huge_ids.size());
		for (size_t k = 0; k < huge_ids.size(); ++k) {
			memset(&hw[k], 0, sizeof(GlobWave));
			hw[k].task[0] = huge_ids[k], hw[k].task[1] = hw[k].task[2] = hw[k].task[3] = -1, hw[k].max_nl = T[huge_ids[k]].nl;
		}
		HIP_TRY(hipMemcpyAsync(d_hw, hw.data(), sizeof(GlobWave) * hw.size(), hipMemcpyHostToDevice, s));
		HIP_TRY(hipMemcpyAsync(d_hlist, huge_ids.data(), 4 * huge_ids.size(), hipMemcpyHostToDevice, s));
		HIP_TRY(wait_stream(ctx, s));                  // (hw is a local; the calls are rare)
	}

	DevTables tabs;
	memcpy(tabs.aa20, tab_aa20(), 256);
	memcpy(tabs.codon, tab_codon(), 64);
	memcpy(tabs.mat, opt->mat, 484);
	DpConst dc;
	dc.go = opt->go, dc.ge = opt->ge, dc.fs = opt->fs, dc.xdrop = opt->xdrop, dc.end_bonus = opt->end_bonus;
	for (int k = 0; k < 6; ++k) dc.sp[k] = opt->sp[k];
	dc.sp_null_bonus = opt->sp_null_bonus;
	dc.wide_ge = wide_ge ? 1 : 0;
	DevGenome dg{ mi->dev[ctx->device]->seq, mi->dev[ctx->device]->ctg_off, mi->dev[ctx->device]->ctg_len, mi->dev[ctx->device]->spsc, mi->l_seq };

	// ---- K3: per-row records and profiles
	// (measured, round 4: putting these two on a high-priority stream of their own gives every DP lane a second active hardware
	// queue, and with ten more queues in use the round kernels are time-sliced: 41 -> 72 ms per launch.  They stay in the lane's
	// own queue; MPA_SHORT_KERNEL raises their wave priority instead.)
	HIP_TRY(hipEventRecord(ctx->ev[0], s));
	if (!chunks.empty())
		hipLaunchKernelGGL(k_prep_rows, dim3((unsigned)chunks.size()), dim3(256), 0, s, dg, ctx->tasks.as<DTask>(), ctx->chunks.as<PrepChunk>(), ctx->rec.as<uint32_t>(), dc, tabs);
	hipLaunchKernelGGL(k_prep_prof, dim3((unsigned)n), dim3(256), 0, s, ctx->tasks.as<DTask>(), ctx->qseq.as<char>(), ctx->prof.as<int16_t>(), tabs);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipEventRecord(ctx->ev[1], s));
	mark("    dp: uploads + prep enqueued");
	// ---- fork: every kernel class (extension classes and the traceback classes of the first round) gets its
	// own stream, so the long single-wave tails of the classes overlap instead of adding up
	HIP_TRY(hipEventRecord(ctx->fork_ev, s));
	int n_side = 0;
	struct Launch { int side; bool is_ext; };
	std::vector<Launch> launches;
	const int n_streams = mpa_ctx_s::kSide;
	auto begin_side = [&](bool is_ext) -> hipStream_t {
		const int k = n_side++;
		// (side streams are created when first used: HIP deals hardware queues to streams in creation order, and sixteen idle side
		// streams per context pushed the main streams of later contexts onto queues that other contexts' long kernels were using)
		hipStream_t &slot = ctx->side[(k + ctx->side_off) % n_streams];
		if (!slot && hipStreamCreateWithFlags(&slot, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); slot = nullptr; }
		hipStream_t st = slot ? slot : s;
		(void)hipStreamWaitEvent(st, ctx->fork_ev, 0);
		(void)hipEventRecord(ctx->lev[2 * k], st);
		launches.push_back(Launch{ k, is_ext });
		return st;
	};
	auto end_side = [&]() { const int k = launches.back().side; hipStream_t st = ctx->side[(k + ctx->side_off) % n_streams]; (void)hipEventRecord(ctx->lev[2 * k + 1], st ? st : s); };

	// ---- K1: extension calls
	ExtArgs ea;
	ea.tasks = ctx->tasks.as<DTask>(), ea.rec = ctx->rec.as<uint32_t>(), ea.prof = ctx->prof.as<int16_t>(), ea.out = ctx->extout.as<ExtOut>();
	ea.c = dc, ea.pen = pen;
	ea.lite = ctx->lite.as<uint32_t>(), ea.ckpt = ctx->ckpt.as<uint32_t>(), ea.score = ctx->score.as<int32_t>();
	ExtWideArgs wa;
	wa.tasks = ea.tasks, wa.rec = ea.rec, wa.prof = ea.prof, wa.out = ea.out, wa.c = dc, wa.pen = pen, wa.key_stride = key_stride;
	wa.xg = n_split ? ctx->xg.as<unsigned long long>() : nullptr;
	wa.done = n_split ? (int32_t*)((char*)ctx->xg.p + xg_bytes) : nullptr;
	wa.ticket = n_split ? wa.done + n_split : nullptr;
	wa.err = n_split ? wa.ticket + n_split : nullptr;
	// rowkey slots follow the order of the wave descriptors (classes 3, 4, 5, 6)
	int64_t rk_first[kNumExtClasses] = { 0 };
	{ int64_t at = 0; for (int cls = 3; cls < kNumExtClasses; ++cls) rk_first[cls] = at, at += ewave_cnt[cls]; }
	// every DP unit of the round goes out in ONE k_dp_round launch on the context's main stream
	GlobArgs ga;
	ga.tasks = ctx->tasks.as<DTask>(), ga.rec = ctx->rec.as<uint32_t>(), ga.prof = ctx->prof.as<int16_t>();
	ga.tb = ctx->tb.as<uint16_t>(), ga.bnd = ctx->bnd.as<int4>(), ga.score = ctx->score.as<int32_t>(), ga.c = dc, ga.rowkey64 = nullptr, ga.waves = nullptr;
	hipEvent_t ev_round0 = ctx->lev[2 * (mpa_ctx_s::kSide - 1)], ev_round1 = ctx->lev[2 * (mpa_ctx_s::kSide - 1) + 1];
	bool round_launched = false;
	const bool use_pool = dp_pool_enabled();
	bool round_pending = false;                              // (worker pool) a round is armed and not yet known to be complete
	hipStream_t round_ws = s;                                // ... and the stream its workers were launched on
	unsigned int round_gen = 0;
	size_t round_units = 0;
	const DpUnit *round_unit_list = nullptr;
	// units of the round: the extension waves/groups of every class, plus (gw_first/gw_cnt != nullptr) the traceback waves of
	// the first traceback chunk; longest first
	auto launch_round = [&](const int *gw_first, const int *gw_cnt, const std::vector<GlobWave> *gwv, GlobWave *d_gw) -> int {
		struct Cost { int64_t cost; DpUnit u; };
		std::vector<Cost> cu;
		auto add = [&](int kind, int first, int count, int64_t cost, int blk = 0, int n_blk = 1, int sgroup = 0, int xg_first = 0) {
			cu.push_back(Cost{ cost, DpUnit{ kind, first, count, blk, n_blk, sgroup, xg_first, 0 } });
		};
		// (worker pool: the one-wave kinds are units of ONE wave descriptor each, taken by single waves; without the pool a
		// workgroup's four waves take four neighbours of the sorted list)
		const int per_narrow = use_pool ? 1 : 4;
		for (int cls = 0; cls < 3; ++cls)
			for (int k = 0; k < ewave_cnt[cls] && !(cls == 1 && ctx->antidiag); k += per_narrow)
				add(U_EXT16 + cls, ewave_first[cls] + k, std::min(per_narrow, ewave_cnt[cls] - k), (int64_t)ewaves[ewave_first[cls] + k].max_nl * 160);
		// (worker pool: a workgroup goes on to its next unit, so all four waves must leave a unit through the same barriers -- a
		// 65..128-column group then takes a whole workgroup on the four-wave body, its waves 2 and 3 on dead columns)
		if (use_pool) for (int k = 0; k < ewave_cnt[3]; ++k) add(U_EXT_W4, ewave_first[3] + k, 1, (int64_t)ewaves[ewave_first[3] + k].max_nl * 270);
		else for (int k = 0; k < ewave_cnt[3]; k += 2) add(U_EXT_W2, ewave_first[3] + k, std::min(2, ewave_cnt[3] - k), (int64_t)ewaves[ewave_first[3] + k].max_nl * 270);
		for (int k = 0; k < ewave_cnt[4]; ++k) add(U_EXT_W4, ewave_first[4] + k, 1, (int64_t)ewaves[ewave_first[4] + k].max_nl * 270);
		for (int k = 0; k < ewave_cnt[6]; ++k)
			for (int b = 0; b < 4; ++b) add(U_EXT_SPLIT, ewave_first[6] + k, 1, (int64_t)ewaves[ewave_first[6] + k].max_nl * 310, b, 4, k, 3 * k);
		for (int k = 0; k < ewave_cnt[5]; ++k)
			for (int b = 0; b < 2; ++b) add(U_EXT_SPLIT, ewave_first[5] + k, 1, (int64_t)ewaves[ewave_first[5] + k].max_nl * 310, b, 2, ewave_cnt[6] + k, 3 * ewave_cnt[6] + k);
		for (int k = 0; k < dwave_cnt; k += per_narrow)              // 65..128-column extension calls, one per wave
			add(U_EXT128, dwave_first + k, std::min(per_narrow, dwave_cnt - k), (int64_t)ewaves[dwave_first + k].max_nl * 200);
		for (int cls = 0; cls < 4; ++cls)                            // the packed sweeps of the checkpointed traceback
			for (int k = 0; k < lwave_cnt[cls]; k += per_narrow)
				add(U_LITE16 + cls, lwave_first[cls] + k, std::min(per_narrow, lwave_cnt[cls] - k), (int64_t)ewaves[lwave_first[cls] + k].max_nl * (cls == 3 ? 200 : 170));
		if (gw_first) {
			static const int kind_of[8] = { U_GLOB16, U_GLOB32, U_GLOB64, U_GLOB_W2, U_GLOB_W4, -1, -1, U_GLOB_MB };
			for (int cls = 0; cls < 8; ++cls) {
				if (kind_of[cls] < 0) continue;
				const int per = cls == 3 ? (use_pool ? 1 : 2) : cls == 4 ? 1 : per_narrow;
				for (int k = 0; k < gw_cnt[cls]; k += per) {
					const GlobWave &g = (*gwv)[gw_first[cls] + k];
					int64_t cost = (int64_t)g.max_nl * (cls >= 3 && cls <= 4 ? 510 : 430);
					if (cls == 7) cost *= (T[g.task[0]].ncol + 63) / 64;
					add(cls == 3 && use_pool ? (int)U_GLOB_W4 : kind_of[cls], gw_first[cls] + k, std::min(per, gw_cnt[cls] - k), cost);
				}
			}
		}
		if (cu.empty()) return MPA_OK;
		std::stable_sort(cu.begin(), cu.end(), [](const Cost &x, const Cost &y) { return x.cost > y.cost; });   // (stable: the workgroups of a split group stay adjacent, in column order)
		static const bool show_top = [] { const char *e = getenv("MPA_DP_TOP"); return e && atoi(e) != 0; }();
		if (show_top) {                                       // (measurement) what bounds the round: the costliest units by kind
			int64_t by_kind[16] = { 0 }, n_kind[16] = { 0 };
			for (const Cost &c : cu) by_kind[c.u.kind & 15] += c.cost, ++n_kind[c.u.kind & 15];
			fprintf(stderr, "[mpa-dp-top] units %zu; longest:", cu.size());
			for (size_t k = 0; k < cu.size() && k < 6; ++k) fprintf(stderr, " kind %d %.1f ms;", cu[k].u.kind, cu[k].cost * 1e-6);
			fprintf(stderr, " | wave-ms by kind:");
			for (int k = 0; k < 16; ++k) if (n_kind[k]) fprintf(stderr, " %d: %ld units %.0f ms;", k, (long)n_kind[k], by_kind[k] * 1e-6);
			fprintf(stderr, "\n");
		}
		// worker pool: the units that take a whole workgroup first (queue 0), then the one-wave units (queue 1), each longest first;
		// priorities stay relative to the round's longest unit of either kind
		const int64_t cost_max = cu[0].cost;
		size_t n_group_units = 0;
		if (use_pool) {
			auto is_group = [](const Cost &c) { return c.u.kind == U_EXT_W4 || c.u.kind == U_EXT_SPLIT || c.u.kind == U_GLOB_W4; };
			std::stable_partition(cu.begin(), cu.end(), is_group);
			for (const Cost &c : cu) n_group_units += is_group(c);
		}
		DpUnit *units = (DpUnit*)(hup + up_units);          // (pinned: the copy below needs no wait)
		const size_t n_units = cu.size();
		if (sizeof(DpUnit) * n_units > up_off - up_units) { set_error("internal: more DP units than the staging buffer holds"); return MPA_ERR_HIP; }
		// the units that bound the round's duration issue ahead of the short ones they share a SIMD with (s_setprio in k_dp_round)
		for (size_t k = 0; k < cu.size(); ++k) {
			units[k] = cu[k].u;
			units[k].pad_ = cu[k].cost * 2 >= cost_max ? 3 : cu[k].cost * 4 >= cost_max ? 2 : cu[k].cost * 10 >= cost_max ? 1 : 0;
			static const bool unit_prio = [] { const char *e = getenv("MPA_DP_PRIO"); return !e || atoi(e) != 0; }();   // (MPA_DP_PRIO=0: measurement)
			if (!unit_prio) units[k].pad_ = 0;
		}
		int rc2;
		if ((rc2 = ctx->units.ensure(n_units * sizeof(DpUnit)))) return rc2;
		HIP_TRY(hipMemcpyAsync(ctx->units.p, units, n_units * sizeof(DpUnit), hipMemcpyHostToDevice, s));
		ea.waves = ctx->waves.as<ExtWave>();
		wa.waves = ctx->waves.as<ExtWave>();                // absolute descriptor indices: the rowkey slot of group g is g - first wide group
		wa.rowkey = ctx->rowkey.as<uint32_t>() - (int64_t)ewave_first[3] * 2 * key_stride;
		ga.waves = d_gw;
		// Workgroups of the round kernel per CU, enforced through the LDS it asks for (MPA_DP_WG_PER_CU, default 3).  Round 6: the
		// kernel takes 124 VGPRs (the asm rows of ext_narrow keep the whole DP state of eight calls in registers and nothing is
		// spilled), so three workgroups hold 372 of each SIMD's 512 registers -- what four workgroups of the 95-register kernel of
		// rounds 4-5 held (384) -- and the seeding kernels of the next batches stay co-resident.  One more workgroup than wanted must
		// NOT fit; what is left of the LDS stays free for the seeding kernels.
		static const size_t round_lds = [] {
			const char *e = getenv("MPA_DP_WG_PER_CU");
			int want = e ? atoi(e) : 3;
			if (want < 1) want = 1;
			if (want > 4) want = 4;
			const size_t pad = (((size_t)160 * 1024 / (want + 1)) + 256) & ~(size_t)255;
			return pad > DP_ROUND_LDS ? pad : DP_ROUND_LDS;
		}();
		if (use_pool) {
			// the round's arguments and units into the lane's slot of the device's pool, the slot armed in stream order behind
			// everything the round reads, the lane's workers on the worker stream (a worker takes any lane's units: this stream is
			// never waited for by a round -- a round is complete when its last unit says so in pinned memory)
			if ((rc2 = pool_attach(ctx))) return rc2;
			mpa_ctx_s *root = ctx->root ? ctx->root : ctx;
			DpPool *pool = root->dp_pool;
			int n_slots;
			{ std::lock_guard<std::mutex> g(root->pool_mu); n_slots = root->pool_slots; }
			DpRoundArgs *ha = (DpRoundArgs*)(hup + up_args);
			ha->ea = ea, ha->wa = wa, ha->ga = ga, ha->units = ctx->units.as<DpUnit>(), ha->n_group = (int32_t)n_group_units, ha->pad_ = 0;
			HIP_TRY(hipMemcpyAsync(&pool->args[ctx->dp_slot], ha, sizeof(DpRoundArgs), hipMemcpyHostToDevice, s));
			long long *d_trace = nullptr;
			if (dp_trace_path()) {
				if ((rc2 = ctx->dp_trace.ensure(n_units * 16))) return rc2;
				HIP_TRY(hipMemsetAsync(ctx->dp_trace.p, 0, n_units * 16, s));
				d_trace = ctx->dp_trace.as<long long>();
			}
			round_gen = ++ctx->dp_gen;
			hipLaunchKernelGGL(k_dp_arm, dim3(1), dim3(1), 0, s, pool, ctx->dp_slot, (int)n_group_units, (int)(n_units - n_group_units), round_gen, ctx->dp_done, d_trace);
			HIP_TRY(hipGetLastError());
			// The workers go out on the lane's own stream (what follows the round on that stream then also waits for this launch's
			// workers to run out of units of ANY lane; measured level with a stream of their own, 19.6 against 19.7 M residues/s).
			// MPA_DP_WORKER_STREAM=1: a worker stream per lane -- one more stream per lane for HIP to deal hardware queues to, and
			// when that stream lands on a queue another context's long kernels use, every round waits for them (the evidence run of
			// round 5 measured 6.5 M residues/s that way: profiles/r05_experiments.txt).
			static const bool own_stream = [] { const char *e = getenv("MPA_DP_WORKER_STREAM"); return e && atoi(e) != 0; }();
			hipStream_t ws = own_stream ? ctx->worker_stream : s;
			if (own_stream) {
				HIP_TRY(hipEventRecord(ctx->arm_ev, s));
				HIP_TRY(hipStreamWaitEvent(ws, ctx->arm_ev, 0));
			}
			mpa_ctx_s::WorkerLaunch wl;
			if (!ctx->wl_free.empty()) wl = ctx->wl_free.back(), ctx->wl_free.pop_back();
			else { HIP_TRY(hipEventCreate(&wl.e0)); HIP_TRY(hipEventCreate(&wl.e1)); }
			HIP_TRY(ensure_dynamic_lds((const void*)k_dp_worker, ctx->device, round_lds));
			static const int launch_cap = [] { const char *e = getenv("MPA_DP_LAUNCH_WORKERS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : dp_pool_budget(); }();
			// a workgroup serves one workgroup unit at a time, or four one-wave units side by side
			const unsigned grid = (unsigned)std::min<size_t>(n_group_units + (n_units - n_group_units + 3) / 4, (size_t)launch_cap);
			HIP_TRY(hipEventRecord(wl.e0, ws));
			hipLaunchKernelGGL(k_dp_worker, dim3(grid), dim3(256), round_lds, ws, pool, ctx->dp_slot, n_slots);
			HIP_TRY(hipGetLastError());
			HIP_TRY(hipEventRecord(wl.e1, ws));
			ctx->wl_busy.push_back(wl);
			round_pending = true, round_units = n_units, round_unit_list = units;
		} else {
			if (round_lds > 48 * 1024) HIP_TRY(ensure_dynamic_lds((const void*)k_dp_round, ctx->device, round_lds));
			HIP_TRY(hipEventRecord(ev_round0, s));
			hipLaunchKernelGGL(k_dp_round, dim3((unsigned)n_units), dim3(256), round_lds, s, ea, wa, ga, ctx->units.as<DpUnit>());
			HIP_TRY(hipGetLastError());
			HIP_TRY(hipEventRecord(ev_round1, s));
		}
		round_launched = true;
		ctx->stats.launches_ext++;
		return MPA_OK;
	};
	// worker pool: the round is complete when the last of its units has stored the round's generation into the lane's pinned word
	auto wait_round = [&]() -> int {
		if (!round_pending) return MPA_OK;
		volatile int32_t *d = ctx->dp_done;
		const double t0 = now_ms();
		for (int polls = 0; (unsigned int)*d != round_gen; ++polls) {
			if (polls >= 8) { struct timespec ts = { 0, 100000L }; nanosleep(&ts, nullptr); }
			if ((polls & 1023) == 1023) {
				if (now_ms() - t0 > 120000.0) { set_error("DP worker pool: a round did not complete within two minutes"); return MPA_ERR_HIP; }
				const hipError_t e = hipStreamQuery(round_ws);                     // (a fault in a worker kernel shows up here, not in the word)
				if (e != hipSuccess && e != hipErrorNotReady) { set_error(std::string("DP worker launch: ") + hipGetErrorString(e)); return MPA_ERR_HIP; }
			}
		}
		round_pending = false;
		hipLaunchKernelGGL(k_l2_writeback, dim3(128), dim3(64), 0, s);       // the units' results out of the L2s, before anything enqueued behind reads them
		HIP_TRY(hipGetLastError());
		if (const char *path = dp_trace_path()) {                             // (debug) one line per unit: who ran when
			std::vector<long long> tr(2 * round_units);
			HIP_TRY(hipMemcpy(tr.data(), ctx->dp_trace.p, round_units * 16, hipMemcpyDeviceToHost));
			static std::mutex tmu;
			std::lock_guard<std::mutex> g(tmu);
			if (FILE *f = fopen(path, "a")) {
				for (size_t k = 0; k < round_units; ++k)
					fprintf(f, "%d\t%u\t%zu\t%d\t%d\t%lld\t%lld\n", ctx->dp_slot, round_gen, k, round_unit_list[k].kind, round_unit_list[k].pad_, tr[2 * k], tr[2 * k + 1]);
				fclose(f);
			}
		}
		return MPA_OK;
	};
	if (ctx->antidiag && ewave_cnt[1] > 0) {                           // (measurement) the 32-column class on the anti-diagonal prototype, one wave per block
		ea.waves = ctx->waves.as<ExtWave>();
		hipStream_t st = begin_side(true);
		hipLaunchKernelGGL(k_ext_antidiag, dim3((unsigned)ewave_cnt[1]), dim3(64), EXT_ANTIDIAG_LDS, st, ea, ewave_first[1]);
		HIP_TRY(hipGetLastError());
		end_side();
		ctx->stats.launches_ext++;
	}
	if (!huge_ids.empty()) {                                           // block-major sweep with the traceback kernel's arithmetic, then the replay
		GlobArgs ha;
		ha.tasks = ctx->tasks.as<DTask>(), ha.waves = d_hw, ha.rec = ctx->rec.as<uint32_t>(), ha.prof = ctx->prof.as<int16_t>();
		ha.tb = nullptr, ha.bnd = ctx->bnd.as<int4>(), ha.score = nullptr, ha.c = dc, ha.rowkey64 = ctx->hkey.as<unsigned long long>();
		hipStream_t st = begin_side(true);
		if (wide_ge) hipLaunchKernelGGL(k_ext_huge<true>, dim3((unsigned)huge_ids.size()), dim3(64), (size_t)22 * 64 * 2 + 4 * 32 * 4, st, ha);
		else hipLaunchKernelGGL(k_ext_huge<false>, dim3((unsigned)huge_ids.size()), dim3(64), (size_t)22 * 64 * 2 + 4 * 32 * 4, st, ha);
		HIP_TRY(hipGetLastError());
		hipLaunchKernelGGL(k_ext_replay, dim3((unsigned)huge_ids.size()), dim3(64), 0, st, ctx->tasks.as<DTask>(), d_hlist, (int32_t)huge_ids.size(),
		                   ctx->hkey.as<unsigned long long>(), ctx->extout.as<ExtOut>(), dc, pen);
		HIP_TRY(hipGetLastError());
		end_side();
		ctx->stats.launches_ext++;
	}

	// ---- K2 + traceback walk
	float ms_glob = 0, ms_bt = 0;
	bool glob_timed = false;
	std::vector<GlobWave> gwaves;
	std::vector<int32_t> glist;
	for (size_t ri = 0; ri < rounds.size(); ++ri) {
		auto &r = rounds[ri];
		if (r.last == r.first) continue;
		const bool concurrent = ri == 0;                                  // later rounds reuse the traceback buffer: serial
		gwaves.clear(), glist.clear();
		const int kNumGlobClasses = 8;
		int cnt[kNumGlobClasses] = { 0 }, first[kNumGlobClasses] = { 0 };
		size_t p = r.first;
		for (int cls = 0; cls < kNumGlobClasses; ++cls) {
			const int per = cls == 0 ? 4 : cls == 1 ? 2 : 1;
			first[cls] = (int)gwaves.size();
			while (p < r.last && T[glob_ids[p]].pad_ == cls) {
				GlobWave w;
				memset(&w, 0, sizeof(w));
				for (int k = 0; k < 4; ++k) w.task[k] = -1;
				for (int k = 0; k < per && p < r.last && T[glob_ids[p]].pad_ == cls; ++k, ++p) {
					w.task[k] = glob_ids[p];
					w.max_nl = std::max(w.max_nl, T[glob_ids[p]].nl);
					glist.push_back(glob_ids[p]);
				}
				gwaves.push_back(w);
			}
			cnt[cls] = (int)gwaves.size() - first[cls];
		}
		if (!concurrent) {                                                // join everything before the buffers (and their staging) are reused
			for (auto &l : launches) (void)hipStreamWaitEvent(s, ctx->lev[2 * l.side + 1], 0);
			HIP_TRY(wait_stream(ctx, s));
			float a = 0, b = 0;                                              // (the previous chunk's sweep and walk)
			(void)hipEventElapsedTime(&a, ctx->ev[3], ctx->ev[4]);
			(void)hipEventElapsedTime(&b, ctx->ev[4], ctx->ev[5]);
			ms_glob += a, ms_bt += b;
		}
		int32_t *d_list = ctx->list.as<int32_t>();
		GlobWave *d_gw = (GlobWave*)((char*)ctx->list.p + (((size_t)n * 4 + 63) & ~(size_t)63));
		memcpy(hup + up_list, glist.data(), glist.size() * 4);
		memcpy(hup + up_gw, gwaves.data(), gwaves.size() * sizeof(GlobWave));
		HIP_TRY(hipMemcpyAsync(d_list, hup + up_list, glist.size() * 4, hipMemcpyHostToDevice, s));
		HIP_TRY(hipMemcpyAsync(d_gw, hup + up_gw, gwaves.size() * sizeof(GlobWave), hipMemcpyHostToDevice, s));
		mark("    dp: traceback lists enqueued");
		ga.tb = ctx->tb.as<uint16_t>();
		// classes 0,1,2,7 share one launch ("narrow"); every wide class has its own
		auto launch_glob_class = [&](int cls, hipStream_t st) -> hipError_t {
			if (cls < 0) { ga.waves = d_gw; return launch_glob_narrow(ga, first, cnt, st, wide_ge); }
			ga.waves = d_gw + first[cls];
			switch (cls) {
			case 3: return launch_glob_wide<2>(ga, cnt[cls], st, wide_ge);
			case 4: return launch_glob_wide<4>(ga, cnt[cls], st, wide_ge);
			case 5: return launch_glob_wide<8>(ga, cnt[cls], st, wide_ge);
			default: return launch_glob_wide<16>(ga, cnt[cls], st, wide_ge);
			}
		};
		HIP_TRY(hipEventRecord(ctx->ev[3], s));
		// every launch on its own stream (next to the extension classes in the first round); the walk needs them all
		HIP_TRY(hipEventRecord(ctx->fork_ev, s));
		const size_t first_glob_launch = launches.size();
		const bool in_round = ri == 0 && !wide_ge;           // the first chunk's calls ride in the round's one launch
		const int order[5] = { 6, 5, 4, 3, -1 };
		for (int oi = 0; oi < 5; ++oi) {
			const int cls = order[oi];
			if (in_round && cls < 5) continue;                               // (only the 512/1024-thread traceback classes keep their own launch)
			if (cls >= 0 ? !cnt[cls] : !(cnt[0] + cnt[1] + cnt[2] + cnt[7])) continue;
			if (n_side >= mpa_ctx_s::kSide - 1) {                            // out of side streams (the last event pair times the round's launch): main stream
				HIP_TRY(launch_glob_class(cls, s));
			} else {
				hipStream_t st = begin_side(false);
				HIP_TRY(launch_glob_class(cls, st));
				end_side();
			}
			ctx->stats.launches_glob++;
		}
		if (in_round) {                                        // (behind the 512/1024-thread classes' own launches: with the worker pool the host waits here)
			if ((rc = launch_round(first, cnt, &gwaves, d_gw)) != MPA_OK) return rc;
			ctx->stats.launches_glob++;
			mark("    dp: units up, round launched");
			if ((rc = wait_round()) != MPA_OK) return rc;
			if (use_pool) mark("    dp: round (units done)");
		}
		for (size_t k = first_glob_launch; k < launches.size(); ++k) (void)hipStreamWaitEvent(s, ctx->lev[2 * launches[k].side + 1], 0);
		HIP_TRY(hipEventRecord(ctx->ev[4], s));
		hipLaunchKernelGGL(k_backtrack, dim3((unsigned)glist.size()), dim3(64), 0, s, ctx->tasks.as<DTask>(), d_list, (int32_t)glist.size(),
		                   ctx->tb.as<uint16_t>(), ctx->cig.as<uint32_t>(), ctx->ncig.as<int32_t>());
		HIP_TRY(hipGetLastError());
		HIP_TRY(hipEventRecord(ctx->ev[5], s));
		glob_timed = true;                                                // (ev[3..5] are read after the next wait)
	}
	if (!round_launched) {
		if ((rc = launch_round(nullptr, nullptr, nullptr, nullptr)) != MPA_OK) return rc;
		mark("    dp: (round without traceback launched)");
		if ((rc = wait_round()) != MPA_OK) return rc;
		if (use_pool) mark("    dp: round (units done)");
	}
	// ---- the walk of the checkpointed traceback: behind the round that swept its calls
	if (n_lite) {
		memcpy(hup + up_wl, glob_ids.data() + n_reg_glob, 4 * n_lite);
		HIP_TRY(hipMemcpyAsync(ctx->wlist.p, hup + up_wl, 4 * n_lite, hipMemcpyHostToDevice, s));
		WalkArgs wk;
		wk.ga = ga, wk.ga.waves = nullptr, wk.list = ctx->wlist.as<int32_t>(), wk.n_list = (int32_t)n_lite;
		wk.lite = ctx->lite.as<uint32_t>(), wk.ckpt = ctx->ckpt.as<uint32_t>(), wk.cig = ctx->cig.as<uint32_t>(), wk.n_cigar = ctx->ncig.as<int32_t>();
		wk.n_blocks = (unsigned long long*)((char*)ctx->wlist.p + ((n_lite * 4 + 63) & ~(size_t)63));
		HIP_TRY(hipMemsetAsync(wk.n_blocks, 0, 8, s));
		// (the list is sorted by class: one launch per class, with the LDS that class's block of direction words needs)
		size_t at = 0;
		for (int cls = 0; cls < 4; ++cls) {
			size_t n_c = 0;
			while (at + n_c < n_lite && T[glob_ids[n_reg_glob + at + n_c]].pad_ == 8 + cls) ++n_c;
			if (n_c == 0) continue;
			wk.list = ctx->wlist.as<int32_t>() + at, wk.n_list = (int32_t)n_c;
			hipLaunchKernelGGL(k_walk, dim3((unsigned)n_c), dim3(64), cls == 0 ? WALK_LDS(16) : cls == 1 ? WALK_LDS(32) : cls == 2 ? WALK_LDS(64) : WALK_LDS(128), s, wk);
			at += n_c;
		}
		HIP_TRY(hipGetLastError());
		ctx->stats.launches_glob++;
	}
	// ---- join
	for (auto &l : launches) (void)hipStreamWaitEvent(s, ctx->lev[2 * l.side + 1], 0);
	HIP_TRY(hipEventRecord(ctx->ev[2], s));
	// results into pinned memory behind the last kernel: extension outputs, traceback scores and CIGAR lengths, hand-off error flag
	const size_t dn_eo = 0, dn_sc = dn_eo + al256(sizeof(ExtOut) * n), dn_nc = dn_sc + al256(4 * n), dn_err = dn_nc + al256(4 * n), dn_wb = dn_err + 256, dn_end = dn_wb + 256;
	if ((rc = ctx->h_down.ensure(dn_end))) return rc;
	char *hdn = ctx->h_down.as<char>();
	*(int32_t*)(hdn + dn_err) = 0;
	if (!ext_ids.empty()) HIP_TRY(hipMemcpyAsync(hdn + dn_eo, ctx->extout.p, sizeof(ExtOut) * n, hipMemcpyDeviceToHost, s));
	if (!glob_ids.empty()) {
		HIP_TRY(hipMemcpyAsync(hdn + dn_sc, ctx->score.p, n * 4, hipMemcpyDeviceToHost, s));
		HIP_TRY(hipMemcpyAsync(hdn + dn_nc, ctx->ncig.p, n * 4, hipMemcpyDeviceToHost, s));
	}
	if (n_split) HIP_TRY(hipMemcpyAsync(hdn + dn_err, wa.err, 4, hipMemcpyDeviceToHost, s));
	*(unsigned long long*)(hdn + dn_wb) = 0;
	if (n_lite) HIP_TRY(hipMemcpyAsync(hdn + dn_wb, (char*)ctx->wlist.p + ((n_lite * 4 + 63) & ~(size_t)63), 8, hipMemcpyDeviceToHost, s));
	mark("    dp: round enqueued");
	HIP_TRY(wait_stream(ctx, s));
	mark("    dp: round (wait)");
	if (glob_timed) {
		float a = 0, b = 0;
		(void)hipEventElapsedTime(&a, ctx->ev[3], ctx->ev[4]);
		(void)hipEventElapsedTime(&b, ctx->ev[4], ctx->ev[5]);
		ms_glob += a, ms_bt += b;
	}
	float ms_ext_sum = 0;                                                 // sum of the per-launch durations of the extension kernels
	if (round_launched && !use_pool) {
		(void)hipEventElapsedTime(&ms_ext_sum, ev_round0, ev_round1);
		ctx->stats.ms_round = ms_ext_sum, ctx->stats.launches_round = 1;
	}
	if (use_pool) pool_harvest(ctx, false);                             // (worker launches that have ended: into the context's totals)
	for (auto &l : launches) {
		float ms = 0;
		(void)hipEventElapsedTime(&ms, ctx->lev[2 * l.side], ctx->lev[2 * l.side + 1]);
		if (l.is_ext) ms_ext_sum += ms;
	}

	// a boundary hand-off that never arrived (bounded spin in the kernel): the producer workgroup was running (it drew its
	// ticket first) but made no progress for seconds -- a stalled hardware queue under oversubscription.  Nothing of this round
	// has been handed to the caller yet: mpa_dp_run() repeats it with those calls on the one-wave path (k_ext_huge, same bits).
	static const bool test_fail = [] { const char *e = getenv("MPA_TEST_HANDOFF_FAIL"); return e && atoi(e) != 0; }();
	if (*(const int32_t*)(hdn + dn_err) || (test_fail && n_split && !ctx->no_split)) {
		set_error("k_ext_wide_split: a column-block hand-off between workgroups timed out"); return MPA_RETRY_NO_SPLIT;
	}
	timing_note("  dp: upload+kernels (wall)", now_ms() - t_begin);
	const double t_res = now_ms();
	// ---- results
	const ExtOut *eo = (const ExtOut*)(hdn + dn_eo);
	const int32_t *sc = (const int32_t*)(hdn + dn_sc), *nc = (const int32_t*)(hdn + dn_nc);
	int64_t pool_n = 0;
	int64_t *dense_off = (int64_t*)(hup + up_off);                        // (the staging buffer's earlier sections have been consumed)
	for (size_t g = 0; g < glob_ids.size(); ++g) dense_off[g] = pool_n, pool_n += nc[glob_ids[g]];
	uint32_t *pool = (uint32_t*)malloc((size_t)(pool_n > 0 ? pool_n : 1) * 4);
	if (pool_n > 0) {
		// the slots were sized for the worst case (nl+al+4 words each): gather the real CIGARs into a dense pool on the
		// device and copy only that over PCIe
		int rc2;
		if ((rc2 = ctx->cigd.ensure((size_t)pool_n * 4)) || (rc2 = ctx->cigoff.ensure(glob_ids.size() * 12 + 64)) || (rc2 = ctx->h_pool.ensure((size_t)pool_n * 4))) { free(pool); return rc2; }
		int64_t *d_off = ctx->cigoff.as<int64_t>();
		int32_t *d_ids = (int32_t*)(d_off + glob_ids.size());
		memcpy(hup + up_ids, glob_ids.data(), glob_ids.size() * 4);
		HIP_TRY(hipMemcpyAsync(d_off, dense_off, glob_ids.size() * 8, hipMemcpyHostToDevice, s));
		HIP_TRY(hipMemcpyAsync(d_ids, hup + up_ids, glob_ids.size() * 4, hipMemcpyHostToDevice, s));
		hipLaunchKernelGGL(k_cigar_gather, dim3((unsigned)glob_ids.size()), dim3(64), 0, s, ctx->tasks.as<DTask>(), d_ids, d_off, (int32_t)glob_ids.size(),
		                   ctx->ncig.as<int32_t>(), ctx->cig.as<uint32_t>(), ctx->cigd.as<uint32_t>());
		HIP_TRY(hipGetLastError());
		HIP_TRY(hipMemcpyAsync(ctx->h_pool.p, ctx->cigd.p, (size_t)pool_n * 4, hipMemcpyDeviceToHost, s));
		HIP_TRY(wait_stream(ctx, s));
		memcpy(pool, ctx->h_pool.p, (size_t)pool_n * 4);
	}
	{
		size_t g = 0;
		std::vector<int64_t> off_of(n, 0);
		for (size_t k = 0; k < glob_ids.size(); ++k) off_of[glob_ids[k]] = dense_off[k];
		(void)g;
		for (int64_t k = 0; k < n; ++k) {
			const DTask &t = T[k];
			mpa_dp_rst_t &o = rst[k];
			if (t.flag & (MPA_F_EXT_LEFT | MPA_F_EXT_RIGHT)) {
				o.nt_len = eo[k].nt_len, o.aa_len = eo[k].aa_len, o.score = eo[k].score, o.n_cigar = 0, o.cigar_off = 0;
			} else {
				o.nt_len = t.nl, o.aa_len = t.al, o.score = sc[k], o.n_cigar = nc[k], o.cigar_off = off_of[k];
			}
		}
	}
	timing_note("  dp: download+assemble", now_ms() - t_res);
	if (cigar_pool) *cigar_pool = pool; else free(pool);
	if (n_pool) *n_pool = pool_n;

	// ---- statistics (cells = (nl-2) * 8*ceil(al/8); algorithmic bytes per call)
	mpa_dp_stats_t &st = ctx->stats;
	for (int32_t id : ext_ids) {
		const DTask &t = T[id];
		if (round_launched && t.pad_ != 7) st.cells_ext_round += (int64_t)std::max(0, t.nl - 2) * t.ncol;
		st.n_ext++, st.cells_ext += (int64_t)std::max(0, t.nl - 2) * t.ncol;
		st.alg_bytes_ext += (t.nl + 1) / 2 + t.al + 12;
	}
	for (size_t gi = 0; gi < glob_ids.size(); ++gi) {
		const int32_t id = glob_ids[gi];
		const DTask &t = T[id];
		const int64_t cells = (int64_t)std::max(0, t.nl - 2) * t.ncol;
		// (the first traceback chunk rides in the round's launch, except the 512/1024-thread classes)
		if (round_launched && ((!rounds.empty() && gi < rounds[0].last && t.pad_ != 5 && t.pad_ != 6) || gi >= n_reg_glob)) st.cells_glob_round += cells;
		st.n_glob++, st.cells_glob += cells;
		if (gi >= n_reg_glob) st.n_ckpt++, st.cells_ckpt += cells;
		st.alg_bytes_glob += (t.nl + 1) / 2 + t.al + 12 + 2 * cells + 2 * ((int64_t)t.nl + t.al) + 4 * (int64_t)nc[id];
	}
	st.rows_prep = rec_total;
	st.walk_blocks = (int64_t)*(const unsigned long long*)(hdn + dn_wb);
	float ms = 0;
	(void)hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]); st.ms_prep = ms;
	st.ms_ext = ms_ext_sum;
	st.ms_glob = ms_glob, st.ms_backtrack = ms_bt;
	timing_note("    dp: GPU prep kernels", st.ms_prep);
	timing_note("    dp: GPU round kernel", ctx->stats.ms_round);
	timing_note("    dp: GPU walk", ms_bt);
	{ float w = 0; (void)hipEventElapsedTime(&w, ctx->ev[0], ctx->ev[2]); st.ms_total = w; }   // wall time of the whole batch on the device
	{
		mpa_dp_stats_t &t = ctx->total;
		t.n_ext += st.n_ext, t.n_glob += st.n_glob, t.cells_ext += st.cells_ext, t.cells_glob += st.cells_glob, t.rows_prep += st.rows_prep;
		t.alg_bytes_ext += st.alg_bytes_ext, t.alg_bytes_glob += st.alg_bytes_glob;
		t.n_ckpt += st.n_ckpt, t.cells_ckpt += st.cells_ckpt, t.walk_blocks += st.walk_blocks;
		t.ms_prep += st.ms_prep, t.ms_ext += st.ms_ext, t.ms_glob += st.ms_glob, t.ms_backtrack += st.ms_backtrack, t.ms_total += st.ms_total;
		t.launches_ext += st.launches_ext, t.launches_glob += st.launches_glob;
		t.cells_ext_round += st.cells_ext_round, t.cells_glob_round += st.cells_glob_round, t.ms_round += st.ms_round, t.launches_round += st.launches_round;
	}
	return MPA_OK;
}

int mpa_dp_run(mpa_ctx_t *ctx, const mpa_idx_t *mi, const mpa_dpopt_t *opt, const mpa_qbatch_t *q,
               int64_t n, const mpa_dp_task_t *in, mpa_dp_rst_t *rst, uint32_t **cigar_pool, int64_t *n_pool)
{
	return mpa::guarded<int>(MPA_ERR_HIP, [&] {
		int rc = mpa_dp_run_impl(ctx, mi, opt, q, n, in, rst, cigar_pool, n_pool);
		if (rc == MPA_RETRY_NO_SPLIT) {
			ctx->no_split = true, ++ctx->handoff_retries;
			rc = mpa_dp_run_impl(ctx, mi, opt, q, n, in, rst, cigar_pool, n_pool);
			ctx->no_split = false;
			if (rc == MPA_RETRY_NO_SPLIT) rc = MPA_ERR_HIP;
		}
		return rc;
	});
}

int64_t mpa_dp_handoff_retries(const mpa_ctx_t *ctx) { return ctx ? ctx->handoff_retries : 0; }
void mpa_dbg_antidiag(mpa_ctx_t *ctx, int on) { if (ctx) ctx->antidiag = on != 0; }
int64_t mpa_device_bytes(void) { return (int64_t)g_dev_bytes.load(); }
int64_t mpa_pool_growths(void) { return (int64_t)g_pool_growths.load(); }

} // extern "C"

#include "gs32_exec.hip"
