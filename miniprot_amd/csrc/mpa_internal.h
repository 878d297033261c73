// mpa_internal.h -- internal declarations shared by the host pipeline and the HIP executor.
// Not part of the C ABI (that is include/mpamd.h).
#pragma once
#include <cstdint>
#include <cstddef>
#include <string>
#include <new>
#include <exception>
#include <vector>
#include <cstring>
#include "../../include/mpamd.h"
#include "chain_core.h"

namespace mpa {

// ---- tables (tables.cpp) -------------------------------------------------------------------------
// Alphabets follow the reference so that scores, hashes and output text are identical:
// nt4: A0 C1 G2 T3 N4 (nasw-tab.c:6); aa20: ARNDCQEGHILKMFPSTWYV*X (nasw-tab.c:9); aa13: nasw-tab.c:12.
extern const uint8_t *tab_nt4();     // [256]
extern const uint8_t *tab_aa20();    // [256]
extern const uint8_t *tab_aa13();    // [256]
extern const uint8_t *tab_codon();   // [64]  codon -> aa20 for the active translation table
extern const uint8_t *tab_codon13(); // [64]  codon -> reduced alphabet
extern const int8_t *blosum62();     // [484]
int set_trans_code(int code);        // ns_make_tables (nasw-tab.c:85); returns <0 if undefined
extern const char *const kAA;        // "ARNDCQEGHILKMFPSTWYV*X"

void set_error(const std::string &msg);
// no C++ exception may cross the C ABI: the extern "C" entry points run their bodies through this
template<typename R, typename F> static inline R guarded(R on_error, F f)
{
	try { return f(); }
	catch (const std::bad_alloc&) { set_error("out of host memory"); }
	catch (const std::exception &e) { set_error(std::string("internal error: ") + e.what()); }
	return on_error;
}

// MPA_TIMING=1 prints wall-clock stage timings to stderr
double now_ms();
bool timing_on();
void timing_note(const char *what, double ms);
double thread_cpu_ms();                                   // CPU time of the calling thread
void cpu_add(const char *label, double ms);               // accumulate under a label (coarse: per stage call, per parallel region and worker)
void cpu_report_and_reset(const char *header, double wall_ms);
struct CpuSpan {                                          // CPU time of the calling thread over a scope, under a label
	const char *label; double t0; bool on;
	explicit CpuSpan(const char *l) : label(l), t0(0), on(timing_on()) { if (on) t0 = thread_cpu_ms(); }
	~CpuSpan() { if (on) cpu_add(label, thread_cpu_ms() - t0); }
};

// ---- index (index.cpp) ---------------------------------------------------------------------------
struct Contig { int64_t off, len; std::string name; };

struct DeviceIndex;                  // opaque, owned by dp executor (mpa_dp.hip)

// The big arrays of an index (packed genome, bucket offsets, occurrence lists): either owned (an index that was built or read
// into memory) or a VIEW into a read-only mapping of the .mpi file (mpa_idx_restore), so that the processes of a multi-GPU job --
// one per GPU, each restoring the same file -- share ONE copy in the page cache instead of holding 7.6 GB each at 3 Gbp.
// The interface is the subset of std::vector the library uses.  A view may be misaligned for T (the .mpi packs its sections back
// to back): at() reads through memcpy; data() is for byte copies (upload, dump) and for callers of the reference's mp_idx_t.
template<class T> struct IdxArray {
	std::vector<T> own;
	T *p = nullptr;
	size_t n = 0;
	T *data() { return p; }
	const T *data() const { return p; }
	size_t size() const { return n; }
	bool empty() const { return n == 0; }
	T &operator[](size_t i) { return p[i]; }                 // (writes: owned storage only, which is aligned)
	T operator[](size_t i) const { return at(i); }           // (reads go through memcpy: a view may be misaligned for T)
	T at(size_t i) const { T v; memcpy(&v, (const char*)p + i * sizeof(T), sizeof(T)); return v; }
	void resize(size_t k) { adopt(); own.resize(k); sync(); }
	void resize(size_t k, T v) { adopt(); own.resize(k, v); sync(); }
	void assign(size_t k, T v) { own.assign(k, v); sync(); }
	void clear() { own.clear(); sync(); }
	void swap(std::vector<T> &o) { own.swap(o); sync(); }
	void view(T *q, size_t k) { std::vector<T>().swap(own); p = q, n = k; }
private:
	void sync() { p = own.data(), n = own.size(); }
	void adopt() { if (n && p != own.data()) { std::vector<T> c(n); memcpy(c.data(), p, n * sizeof(T)); own.swap(c); sync(); } }
};

} // namespace mpa

struct mpa_idx_s {
	mpa_idxopt_t opt;
	int64_t n_kb = 0;
	int64_t l_seq = 0;
	std::vector<mpa::Contig> ctg;
	mpa::IdxArray<uint8_t> seq;      // 4-bit packed, low nibble = even offset (ntseq.c:64-67)
	std::vector<char> names;         // NUL separated, exactly as stored in the .mpi
	mpa::IdxArray<int64_t> ki;       // bucket offsets, 1<<(4*kmer-mod_bit) entries
	mpa::IdxArray<uint32_t> kb;      // global block ids
	std::vector<uint32_t> bo;        // per (contig,strand) block offset, 2*n_ctg+1 entries (index.c:11-26)
	uint32_t n_block = 0;
	std::vector<uint8_t> spsc;       // splice-score track (--spsc), empty if none: [strand][l_seq], indexed by contig offset + strand-local position
	// genome (and, once seeding has run there, the occurrence lists) resident in the HBM of device d (set by mpa_idx_to_device): one copy
	// per device, so that the pipelines of several GPUs of one process map against the same index (mpa_map_batches_multi)
	static const int kMaxDevices = 16;
	mpa::DeviceIndex *dev[kMaxDevices] = {};
	void *map_base = nullptr;        // the .mpi file mapped by mpa_idx_restore (seq / kb are views into it), or null
	size_t map_len = 0;
};

namespace mpa {

inline uint8_t nt_at(const mpa_idx_s *mi, int64_t p) { return mi->seq[p >> 1] >> ((p & 1) * 4) & 0xf; }
// strand-local window fetch; same contract as mp_ntseq_get_by_v (ntseq.c:108-114)
int64_t fetch_nt(const mpa_idx_s *mi, int32_t vid, int64_t st, int64_t en, uint8_t *out);
int32_t block2vid(const mpa_idx_s *mi, uint32_t blk);        // mp_idx_block2pos (index.c:28-44)
int64_t idx_read_spsc(mpa_idx_s *mi, const char *fn, int32_t max_sc);   // mp_ntseq_read_spsc (ntseq.c:234-296)

// ---- device executor (mpa_dp.hip) ----------------------------------------------------------------
int dev_upload_index(mpa_ctx_t *ctx, mpa_idx_s *mi);
int dev_index_build(mpa_ctx_t *ctx, mpa_idx_s *mi);           // k-mer table of a genome-only index on the device (index.c:52-136)
void dev_free_index(mpa_idx_s *mi);
mpa_ctx_t *ctx_sibling(mpa_ctx_t *ctx, int k);   // extra context on the same device (k >= 1), owned by ctx
void ctx_absorb_sibling_stats(mpa_ctx_t *ctx);
void ctx_set_side_offset(mpa_ctx_t *ctx, int off);   // which of its side streams a DP round starts with
void ctx_set_role(mpa_ctx_t *root, mpa_ctx_t *ctx, int role);   // ctx's pools share their high-water marks with the root's other contexts of that role (0 DP lane, 1 seeder, 2 planner)

// ---- GPU seeding (seed_exec.hip + the driver in dp_exec.hip) --------------------------------------
struct SeedJob { int64_t kb_off, dst; int32_t cnt, qpos, qid; };   // one kept seed: its occurrence list and where its anchors go
struct PrechainSparse {              // result for a mini-batch: the chained anchors (with a predecessor, or being one), query by query
	std::vector<int64_t> cfirst;     // [n_query + 1] offsets into the arrays below
	int64_t m = 0;
	const int32_t *pos = nullptr, *f = nullptr, *pred = nullptr;   // pinned buffers owned by the context, valid until its next call; pred = index into the query's part of the view, -1 for none
	const uint64_t *a = nullptr;
	std::vector<uint8_t> on_host;    // [n_query] 1 = the device declined this query (k_seed_sift: too many anchors in one block; k_chain_extract: a degenerate problem): seed and chain it on the host; empty = none
	// has_chains: both chaining rounds ran on the device (pre-chain extraction, main-chain forward pass and extraction): what comes
	// back are the main chains of every query -- u (score << 32 | anchors per chain) and their anchors, chain by chain -- and the
	// arrays above are not filled
	bool has_chains = false;
	std::vector<int64_t> u_first, a_first;   // [n_query + 1]
	const uint64_t *U = nullptr, *A = nullptr;   // pinned buffers of the context, valid until its next call
};
struct RefineWindow { int64_t as; int32_t qid, vid, len; };    // strand-local window [as, as + len) on vid, refined for query qid
struct RefineHits { std::vector<int64_t> first; std::vector<uint64_t> hits; };
int dev_refine_scan(mpa_ctx_t *ctx, mpa_idx_s *mi, int32_t kmer, int32_t min_aa_len, int32_t n_query, const int64_t *qw_first, const uint32_t *qwords,
                    int64_t n_win, const RefineWindow *wins, RefineHits &out);
// The whole refinement of a mini-batch on the device (mp_refine_reg, map.c:32-96): window scan, pairing of equal k-mers, the sort
// of every window's pairs, the forward pass and the extraction of its chain.  The query side comes as GROUPS (the distinct k-mers
// of every query: packed word, and the query positions that carry it).  out: the chains of every window as mp_chain() returns them.
struct RefineGroupsHost {
	std::vector<int64_t> qg_first;           // [n_query + 1]
	std::vector<uint32_t> gword, gfirst, gcount, qpos;
};
struct RefineChains {
	std::vector<int64_t> u_first, a_first;   // [n_win + 1]
	const uint64_t *U = nullptr, *A = nullptr;   // pinned buffers of the context, valid until its next call
};
int dev_refine_chains(mpa_ctx_t *ctx, mpa_idx_s *mi, int32_t kmer, int32_t min_aa_len, int32_t max_ava, const ChainParams &cp, int32_t n_query, const RefineGroupsHost &groups,
                      int64_t n_win, const RefineWindow *wins, RefineChains &out);
// forward pass of mp_chain (chain.c:181-209) for a batch of problems on the device; see dp_exec.hip
struct ChainIO { uint64_t *a = nullptr; int32_t *f = nullptr, *pred = nullptr; };   // pinned buffers of the context, valid until its next chain call
int dev_chain_buffers(mpa_ctx_t *ctx, int64_t n, ChainIO &io);
int dev_chain_forward(mpa_ctx_t *ctx, const ChainParams &p, int32_t n_prob, const int64_t *first, const ChainIO &io);
// main != nullptr: carry on with the main chain on the device (has_chains), if the seeding mode supports it
// hold != nullptr: the pinned result buffers live there (valid until the holder's next use) instead of in the context
struct SeedHold;
SeedHold *ctx_seed_hold(mpa_ctx_t *ctx, int k);   // k-th result holder of a context (created on first use, owned by it)
int dev_prechain_forward(mpa_ctx_t *ctx, mpa_idx_s *mi, const ChainParams &pre, int32_t n_query, const int64_t *qfirst,
                         const SeedJob *jobs, int64_t n_jobs, PrechainSparse &out, const ChainParams *main = nullptr, SeedHold *hold = nullptr);

} // namespace mpa
