// compat.cpp -- the reference's OWN entry points, exported by libmpamd.so with the reference's signatures and record
// layouts, so that a program written against lh3/miniprot's miniprot.h / nasw.h (its example.c, its main.c, third-party
// callers) relinks against this library unchanged.
//
//   boundary #1  miniprot.h:158-286   mp_start, mp_idxopt_init, mp_mapopt_init/_set_fs/_set_max_intron/_check,
//                                     mp_idx_load/restore/dump/destroy/print_stat, mp_set_spsc, mp_ntseq_read_spsc,
//                                     mp_tbuf_init/destroy, mp_map (a batch of one), mp_map_file (a stream of
//                                     mini-batches through mpa_map_batches), mp_verbose, mp_dbg_flag, the sys.c timers
//   boundary #2  nasw.h:94-139        ns_make_tables, ns_opt_init, ns_opt_set_sp, ns_set_stop_sc, ns_global_gs16b
//                                     (one mpa_dp_run call on an uploaded copy of the caller's nucleotide buffer)
//
// Ownership follows the reference: mp_map() returns a calloc'd mp_reg1_t[] whose ->p and ->feat are malloc'd and freed
// by the caller (map.c:314-318, example.c:54-56); ->a is never valid after the call.  The index returned by
// mp_idx_load() is a view: the public mp_idx_t / mp_ntdb_t / mp_ctg_t records point into the library's own index.
//
// There is no CPU fallback behind these either: without a usable HIP device the mapping entry points print the reason
// and abort() -- the reference's functions have no error channel (SURVEY.md 8b "Errors").
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstddef>
#include <mutex>
#include <string>
#include <vector>
#include <sys/time.h>
#include <sys/resource.h>
#include <fcntl.h>
#include <unistd.h>
#include "../../include/miniprot.h"
#include "../../include/nasw.h"
#include "mpa_internal.h"
#include "fasta.h"

using namespace mpa;

// the option records are the same bytes under both names
static_assert(sizeof(mp_mapopt_t) == sizeof(mpa_mapopt_t) && offsetof(mp_mapopt_t, mat) == offsetof(mpa_mapopt_t, mat) &&
              offsetof(mp_mapopt_t, gff_prefix) == offsetof(mpa_mapopt_t, gff_prefix), "mp_mapopt_t layout");
static_assert(sizeof(mp_idxopt_t) == sizeof(mpa_idxopt_t) && sizeof(mp_idxopt_t) == 20, "mp_idxopt_t layout");
static_assert(sizeof(mp_feat_t) == sizeof(mpa_feat_t) && sizeof(mp_feat_t) == 56 && offsetof(mp_feat_t, donor) == offsetof(mpa_feat_t, donor), "mp_feat_t layout");
static_assert(sizeof(mp_reg1_t) == 104 && sizeof(mp_extra_t) == 48 && sizeof(mp_mapopt_t) == 656, "record sizes of the reference ABI");
static_assert(sizeof(ns_opt_t) == 96 && sizeof(ns_rst_t) == 32, "operator records of the reference ABI");

extern "C" {
int32_t mp_verbose = 3, mp_dbg_flag = 0;                 // misc.c:10
void *kmalloc(void *km, size_t size) __attribute__((weak));   // the caller's kalloc, if the program has one (nasw.h:77)
}

namespace {

[[noreturn]] void die(const char *where)
{
	fprintf(stderr, "[libmpamd] %s: %s\n", where, mpa_last_error());
	abort();
}

// ---- the index view ---------------------------------------------------------------------------------
struct CompatIndex {
	mp_idx_t pub;                 // first member: mp_idx_t* <-> CompatIndex*
	mp_ntdb_t nt;
	std::vector<mp_ctg_t> ctg;
	mpa_idx_t *impl = nullptr;
};

CompatIndex *wrap_index(mpa_idx_t *impl)
{
	if (!impl) return nullptr;
	CompatIndex *c = new CompatIndex();
	c->impl = impl;
	const int32_t n = (int32_t)impl->ctg.size();
	c->ctg.resize((size_t)n);
	const char *p = impl->names.data();
	for (int32_t i = 0; i < n; ++i) {
		c->ctg[i].off = impl->ctg[i].off, c->ctg[i].len = impl->ctg[i].len, c->ctg[i].name = const_cast<char*>(p);
		p += strlen(p) + 1;
	}
	memset(&c->nt, 0, sizeof(c->nt));
	c->nt.n_ctg = c->nt.m_ctg = n, c->nt.l_name = (int32_t)impl->names.size();
	c->nt.l_seq = impl->l_seq, c->nt.m_seq = (impl->l_seq + 1) >> 1 << 1;
	c->nt.seq = impl->seq.data(), c->nt.ctg = c->ctg.data(), c->nt.name = impl->names.data();
	memset(&c->pub, 0, sizeof(c->pub));
	memcpy(&c->pub.opt, &impl->opt, sizeof(mp_idxopt_t));
	c->pub.n_block = impl->n_block, c->pub.nt = &c->nt;
	// (kb of a RESTORED index is a view into the mapped .mpi and may be misaligned for uint32_t -- the section follows the contig
	// names; a caller that reads mp_idx_t::kb itself should memcpy the entries out.  ki, bo and every built index are aligned.)
	c->pub.n_kb = impl->n_kb, c->pub.ki = impl->ki.data(), c->pub.bo = impl->bo.data(), c->pub.kb = impl->kb.data();
	return c;
}
inline CompatIndex *owner(const mp_idx_t *mi) { return reinterpret_cast<CompatIndex*>(const_cast<mp_idx_t*>(mi)); }

// ---- device contexts ----------------------------------------------------------------------------------
int default_device()
{
	// MPA_DEVICE names the device; without it a process started by a one-process-per-GPU launcher (torchrun, mpirun, srun) takes
	// the GPU of its local rank, so that N ranks do not all sit on device 0 -- or, worse, each on every GPU of the node
	for (const char *name : { "MPA_DEVICE", "LOCAL_RANK", "OMPI_COMM_WORLD_LOCAL_RANK", "SLURM_LOCALID" }) {
		const char *e = getenv(name);
		if (e && *e) {
			const int d = atoi(e), n = mpa_device_count();
			return n > 0 ? (d % n + n) % n : d;
		}
	}
	return 0;
}
std::mutex g_mu;                     // guards the shared context and index uploads
mpa_ctx_t *g_ctx = nullptr;
std::vector<mpa_ctx_t*> g_file_ctx;
// The contexts of the reference-signature entry points live as long as the process (the reference's API has no shutdown call).
// What their teardown costs at exit is measured here when MPA_TIMING is set (a 100-GB set of device pools, pinned staging,
// a dozen contexts' streams); MPA_FAST_EXIT=1 leaves all of it to the kernel: stdio is flushed and the process ends with
// _exit() -- for command-line use where nothing else is registered to run at exit.
void at_exit(int status, void*)
{
	const char *fast = getenv("MPA_FAST_EXIT");
	if (fast && atoi(fast) != 0) { fflush(nullptr); _exit(status); }
	if (!timing_on()) return;
	const double t0 = now_ms();
	for (mpa_ctx_t *c : g_file_ctx) if (c != g_ctx) mpa_ctx_destroy(c);
	if (g_ctx) mpa_ctx_destroy(g_ctx);
	g_file_ctx.clear(), g_ctx = nullptr;
	fprintf(stderr, "[mpa-timing] at exit: device contexts destroyed in %.1f ms (then the HIP runtime's own teardown)\n", now_ms() - t0);
}
void hook_exit();
mpa_ctx_t *shared_ctx()
{
	if (!g_ctx) {
		g_ctx = mpa_ctx_create(default_device());
		if (!g_ctx) die("no usable HIP device (there is no CPU fallback)");
		hook_exit();
	}
	return g_ctx;
}

void hook_exit()
{
	static bool hooked = false;
	if (!hooked) hooked = true, on_exit(at_exit, nullptr);
}

// mp_map_file() maps on ONE device unless told otherwise (the reference's kt_for over threads, map.c:291, becomes one pipeline per
// device NAMED): MPA_DEVICES = "all" (every visible GPU), or a list ("0,2,5"; a single id is a list of one; a device may be named
// twice: two pipelines on it -- how the single-GPU tests exercise this path).  Every id is checked against the visible devices; a
// malformed value is an error, not a silent default.  Without MPA_DEVICES: the default device (MPA_DEVICE, or the launcher's local
// rank) -- a process on a shared node, or one rank of a multi-rank launch, must not take every GPU just because it can see them.
// (Round 5 mapped on all of them by default, before the multi-device path had run on more than one device.)
const std::vector<mpa_ctx_t*> &file_ctxs()             // caller holds g_mu
{
	if (!g_file_ctx.empty()) return g_file_ctx;
	std::vector<int> devs;
	const int n_dev = mpa_device_count();
	if (n_dev <= 0) die("mp_map_file: no usable HIP device (there is no CPU fallback)");
	const char *e = getenv("MPA_DEVICES");
	if (!e || !*e) devs.push_back(default_device());
	else if (strcmp(e, "all") == 0) { for (int d = 0; d < n_dev; ++d) devs.push_back(d); }
	else {
		for (const char *p = e; *p; ) {
			char *end = nullptr;
			const long d = strtol(p, &end, 10);
			if (end == p || (*end && *end != ',') || d < 0 || d >= n_dev) {
				fprintf(stderr, "[mpamd] MPA_DEVICES=\"%s\": expected \"all\" or a comma-separated list of device ids below %d\n", e, n_dev);
				exit(1);
			}
			devs.push_back((int)d);
			p = *end ? end + 1 : end;
		}
	}
	if (devs.size() > 8) devs.resize(8);
	bool shared_used = false;
	for (int d : devs) {
		mpa_ctx_t *c = nullptr;
		if (d == default_device() && !shared_used) c = shared_ctx(), shared_used = true;
		else c = mpa_ctx_create(d);
		if (!c) die("mp_map_file: no usable HIP device (there is no CPU fallback)");
		g_file_ctx.push_back(c);
	}
	if (g_file_ctx.empty()) g_file_ctx.push_back(shared_ctx());
	hook_exit();                                         // (also when none of the devices is the default one: MPA_FAST_EXIT / MPA_TIMING apply)
	return g_file_ctx;
}

double g_t0 = 0;
double wall_now()
{
	struct timeval tv;
	gettimeofday(&tv, nullptr);
	return tv.tv_sec + tv.tv_usec * 1e-6;
}

} // namespace

struct mp_tbuf_s { mpa_ctx_t *ctx; };

extern "C" {

// ---- sys.c / misc.c ------------------------------------------------------------------------------------
double mp_realtime(void) { const double t = wall_now(); if (g_t0 == 0) g_t0 = t; return t - g_t0; }
double mp_cputime(void)
{
	struct rusage r;
	getrusage(RUSAGE_SELF, &r);
	return r.ru_utime.tv_sec + r.ru_stime.tv_sec + 1e-6 * (r.ru_utime.tv_usec + r.ru_stime.tv_usec);
}
long mp_peakrss(void)
{
	struct rusage r;
	getrusage(RUSAGE_SELF, &r);
	return r.ru_maxrss * 1024;
}
double mp_percent_cpu(void) { const double t = mp_realtime(); return t > 0 ? (mp_cputime() + 1e-9) / (t + 1e-9) : 0; }

void mp_start(void) { ns_make_tables(MP_CODON_STD); mp_realtime(); }

// ---- options (options.c) -----------------------------------------------------------------------------
void mp_idxopt_init(mp_idxopt_t *io) { mpa_idxopt_init(reinterpret_cast<mpa_idxopt_t*>(io)); }
void mp_mapopt_init(mp_mapopt_t *mo) { mpa_mapopt_init(reinterpret_cast<mpa_mapopt_t*>(mo)); }
void mp_mapopt_set_fs(mp_mapopt_t *mo, int32_t fs) { mpa_mapopt_set_fs(reinterpret_cast<mpa_mapopt_t*>(mo), fs); }
void mp_mapopt_set_max_intron(mp_mapopt_t *mo, int64_t gsize)
{
	mpa_mapopt_set_max_intron(reinterpret_cast<mpa_mapopt_t*>(mo), gsize);
	if (mp_verbose >= 3) fprintf(stderr, "[M::%s] set max intron size to %d\n", __func__, mo->max_intron);
}
int32_t mp_mapopt_check(const mp_mapopt_t *mo)
{
	if (mo->sp_model < 0 || mo->sp_model > 2) {
		fprintf(stderr, "[ERROR]\033[1;31m option -j should be between 0 and 2\033[0m\n");
		return -1;
	}
	return 0;
}

// ---- nasw-tab.c ------------------------------------------------------------------------------------------
int ns_make_tables(int codon_type)
{
	if (codon_type < 0 || codon_type > 33) return -1;
	return set_trans_code(codon_type) < 0 ? -2 : 0;
}
void ns_opt_set_sp(ns_opt_t *opt, int32_t model)
{
	static const int32_t tab[3][6] = { {0,0,0,0,0,0}, {8,15,21,30,0,0}, {8,15,21,30,4,4} };
	const int m = model == NS_S_MAMMAL ? 2 : model == NS_S_GENERIC ? 1 : 0;
	memcpy(opt->sp, tab[m], sizeof(opt->sp));
}
void ns_opt_init(ns_opt_t *opt)
{
	memset(opt, 0, sizeof(*opt));
	opt->go = 11, opt->ge = 1, opt->io = 29, opt->fs = 17, opt->xdrop = 100, opt->end_bonus = 5;
	ns_opt_set_sp(opt, NS_S_MAMMAL);
	opt->sp_null_bonus = -7, opt->asize = 22, opt->ie_coef = .5f;
	opt->sc = ns_mat_blosum62, opt->nt4 = ns_tab_nt4, opt->aa20 = ns_tab_aa20, opt->codon = ns_tab_codon;
}
void ns_set_stop_sc(int32_t asize, int8_t *mat, int8_t pen)
{
	const int32_t stop = ns_tab_aa20[(uint8_t)'*'];
	const int8_t keep = mat[stop * asize + stop];
	for (int32_t i = 0; i < asize; ++i) mat[stop * asize + i] = mat[i * asize + stop] = (int8_t)-pen;
	mat[stop * asize + stop] = keep;
}

// ---- index ---------------------------------------------------------------------------------------------
mp_idx_t *mp_idx_restore(const char *fn)
{
	CompatIndex *c = wrap_index(mpa_idx_restore(fn));
	if (c && mp_verbose >= 3)
		fprintf(stderr, "[M::%s@%.3f*%.2f] loaded the index\n", __func__, mp_realtime(), mp_percent_cpu());
	return c ? &c->pub : nullptr;
}

mp_idx_t *mp_idx_load(const char *fn, const mp_idxopt_t *io, int32_t n_threads)
{
	// an index file starts with "MPI" and a version byte <= 3 (index.c:166-187); "-" is a FASTA stream
	bool is_idx = false;
	if (strcmp(fn, "-") != 0) {
		const int fd = open(fn, O_RDONLY);
		if (fd < 0) return nullptr;
		char magic[4];
		if (read(fd, magic, 4) == 4 && memcmp(magic, MP_IDX_MAGIC, 3) == 0 && magic[3] <= MP_IDX_MAGIC[3]) is_idx = true;
		close(fd);
	}
	if (is_idx) return mp_idx_restore(fn);
	// mp_idx_build (index.c:97-136): the genome from the FASTA, then the k-mer table -- on the GPU when there is one (scan kernel
	// + one radix sort, seconds for a human-sized genome), on n_threads host threads otherwise; the same table either way
	mpa_idx_t *impl = mpa_idx_read_fasta(fn, reinterpret_cast<const mpa_idxopt_t*>(io));
	if (impl) {
		bool built = false;
		if (mpa_device_count() > 0) {
			std::lock_guard<std::mutex> g(g_mu);
			if (!g_ctx) g_ctx = mpa_ctx_create(default_device());
			built = g_ctx && mpa_idx_build_kmers_device(g_ctx, impl) == MPA_OK;
		}
		if (!built) mpa_idx_build_kmers(impl, n_threads > 1 ? n_threads : 1);
	}
	CompatIndex *c = wrap_index(impl);
	if (c && mp_verbose >= 3)
		fprintf(stderr, "[M::%s@%.3f*%.2f] %ld kmer-block pairs\n", "mp_idx_build", mp_realtime(), mp_percent_cpu(), (long)c->pub.n_kb);
	return c ? &c->pub : nullptr;
}

int mp_idx_dump(const char *fn, const mp_idx_t *mi) { return mpa_idx_dump(fn, owner(mi)->impl) == MPA_OK ? 0 : -1; }

void mp_idx_destroy(mp_idx_t *mi)
{
	if (!mi) return;
	CompatIndex *c = owner(mi);
	mpa_idx_destroy(c->impl);
	delete c;
}

void mp_idx_print_stat(const mp_idx_t *mi, int32_t max_occ)
{
	const uint32_t n = 1U << (mi->opt.kmer * MP_BITS_PER_AA - mi->opt.mod_bit);
	int64_t tot = 0, sum_large = 0;
	uint32_t n_occupied = 0, n_large = 0;
	for (uint32_t i = 0; i + 1 < n; ++i) {                // (the reference leaves the last bucket out as well, index.c:143)
		const int64_t c = mi->ki[i + 1] - mi->ki[i];
		if (c > 0) ++n_occupied;
		if (c > max_occ) ++n_large, sum_large += c;
		else tot += c;
	}
	fprintf(stderr, "[M::%s] %d distinct k-mers; mean occ of infrequent k-mers: %.2f; %d frequent k-mers accounting for %ld occurrences\n", __func__,
	        n_occupied, (double)tot / (n_occupied - n_large), n_large, (long)sum_large);
}

int32_t mp_ntseq_read_spsc(mp_ntdb_t *nt, const char *fn, int32_t max_sc)
{
	CompatIndex *c = reinterpret_cast<CompatIndex*>(reinterpret_cast<char*>(nt) - offsetof(CompatIndex, nt));
	return idx_read_spsc(c->impl, fn, max_sc) < 0 ? -1 : 0;
}

void mp_set_spsc(const char *fn, mp_idx_t *mi, mp_mapopt_t *mo, int32_t keep_io)
{
	if (!fn) return;
	if (mpa_idx_set_spsc(owner(mi)->impl, fn, reinterpret_cast<mpa_mapopt_t*>(mo), keep_io) < 0)
		fprintf(stderr, "[libmpamd] mp_set_spsc: %s\n", mpa_last_error());
}

// ---- mp_map: a batch of one ---------------------------------------------------------------------------
mp_tbuf_t *mp_tbuf_init(void)
{
	mp_tbuf_t *b = (mp_tbuf_t*)calloc(1, sizeof(mp_tbuf_t));
	return b;                                            // its device context is created by the first mp_map() that uses it
}

// Device contexts of thread buffers are pooled: the reference's own calling pattern (map.c:273-345, worker_pipeline) creates
// and destroys n_threads buffers per mini-batch, and a context is a set of HIP streams, events and grow-only device pools that
// is expensive to build.  A destroyed buffer parks its context; the next mp_map() on a fresh buffer picks it up.
static std::vector<mpa_ctx_t*> g_ctx_pool;              // (guarded by g_mu; contexts of default_device())
static mpa_ctx_t *pooled_ctx()                          // caller holds g_mu
{
	if (!g_ctx_pool.empty()) { mpa_ctx_t *c = g_ctx_pool.back(); g_ctx_pool.pop_back(); return c; }
	return mpa_ctx_create(default_device());
}

void mp_tbuf_destroy(mp_tbuf_t *b)
{
	if (!b) return;
	if (b->ctx) {
		std::lock_guard<std::mutex> g(g_mu);
		if (g_ctx_pool.size() < 64) g_ctx_pool.push_back(b->ctx);
		else mpa_ctx_destroy(b->ctx);
	}
	free(b);
}

// hits of one query of a result, as the reference's records
static mp_reg1_t *regs_from_result(const mpa_result_t *res, int32_t qid, int *n_reg)
{
	const int64_t *off = mpa_result_hit_off(res);
	const mpa_hit_t *hits = mpa_result_hits(res) + off[qid];
	const int32_t n = (int32_t)(off[qid + 1] - off[qid]);
	*n_reg = n;
	if (n == 0) return nullptr;
	mp_reg1_t *regs = (mp_reg1_t*)calloc((size_t)n, sizeof(mp_reg1_t));
	for (int32_t i = 0; i < n; ++i) {
		const mpa_hit_t &h = hits[i];
		mp_reg1_t &r = regs[i];
		r.cnt = h.cnt, r.id = h.id, r.parent = h.parent, r.n_sub = h.n_sub, r.subsc = h.subsc;
		r.n_exon = h.n_exon, r.chn_sc = h.chn_sc, r.chn_sc_ungap = h.chn_sc_ungap;
		r.vid = h.vid, r.qs = h.qs, r.qe = h.qe, r.vs = h.vs, r.ve = h.ve;
		if (h.has_aln) {
			mp_extra_t *e = (mp_extra_t*)calloc(1, sizeof(mp_extra_t) + sizeof(uint32_t) * (size_t)h.n_cigar);
			e->dp_score = h.dp_score, e->dp_max = h.dp_max, e->dp_max2 = h.dp_max2, e->n_cigar = e->m_cigar = h.n_cigar;
			e->blen = h.blen, e->n_fs = h.n_fs, e->n_stop = h.n_stop, e->dist_stop = h.dist_stop, e->dist_start = h.dist_start;
			e->n_iden = h.n_iden, e->n_plus = h.n_plus;
			memcpy(e->cigar, mpa_result_cigars(res) + h.cigar_off, sizeof(uint32_t) * (size_t)h.n_cigar);
			r.p = e;
			r.n_feat = r.m_feat = h.n_feat;
			if (h.n_feat > 0) {
				r.feat = (mp_feat_t*)calloc((size_t)h.n_feat, sizeof(mp_feat_t));
				memcpy(r.feat, mpa_result_feats(res) + h.feat_off, sizeof(mp_feat_t) * (size_t)h.n_feat);
			}
		}
	}
	return regs;
}

mp_reg1_t *mp_map(const mp_idx_t *mi, int qlen, const char *seq, int *n_reg, mp_tbuf_t *b, const mp_mapopt_t *opt, const char *qname)
{
	(void)qname;
	*n_reg = 0;
	CompatIndex *c = owner(mi);
	const int64_t q_off[2] = { 0, qlen };
	const mpa_qbatch_t q{ 1, seq, q_off };
	mpa_result_t *res = nullptr;
	mpa_ctx_t *ctx;
	std::unique_lock<std::mutex> g(g_mu);
	if (b) {
		if (!b->ctx && !(b->ctx = pooled_ctx())) die("mp_map: no usable HIP device (there is no CPU fallback)");
		ctx = b->ctx;
	} else if (!(ctx = pooled_ctx())) die("mp_map: no usable HIP device (there is no CPU fallback)");   // no buffer: a pooled context for this one call
	if (mpa_idx_to_device(ctx, c->impl) != MPA_OK) die("mp_map: index upload");
	g.unlock();                                          // every call has a context of its own by now: calls run concurrently
	if (mpa_map_batch(ctx, c->impl, reinterpret_cast<const mpa_mapopt_t*>(opt), &q, 1, &res) != MPA_OK) die("mp_map");
	if (!b) { g.lock(); g_ctx_pool.push_back(ctx); g.unlock(); }
	mp_reg1_t *regs = regs_from_result(res, 0, n_reg);
	mpa_result_destroy(res);
	return regs;
}

// ---- mp_map_file: mini-batches of mini_batch_size residues (bseq.c:56-84) through the pipelined mapper ------------
int32_t mp_map_file(const mp_idx_t *idx, const char *fn, const mp_mapopt_t *opt, int n_threads)
{
	FastaReader rd(fn);
	if (!rd.ok()) return -1;
	CompatIndex *c = owner(idx);
	const mpa_mapopt_t *mo = reinterpret_cast<const mpa_mapopt_t*>(opt);
	std::unique_lock<std::mutex> g(g_mu);
	const std::vector<mpa_ctx_t*> &ctxs = file_ctxs();
	for (mpa_ctx_t *ctx : ctxs) if (mpa_idx_to_device(ctx, c->impl) != MPA_OK) die("mp_map_file: index upload");
	if (mp_verbose >= 3 && ctxs.size() > 1) fprintf(stderr, "[M::%s] mapping on %d device pipelines\n", __func__, (int)ctxs.size());
	if (opt->flag & MP_F_GFF) puts("##gff-version 3");
	struct Batch { std::string seqs; std::vector<int64_t> off; std::vector<std::string> names; std::vector<const char*> name_ptr; };
	int64_t id = 0;
	bool more = true;
	std::string name, comment, seq;
	while (more) {
		// a group of mini-batches at a time: enough for the pipeline to overlap its stages, bounded memory for any file size
		const int kGroup = 16 * (int)ctxs.size();
		std::vector<Batch> grp;
		while (more && (int)grp.size() < kGroup) {
			Batch bt;
			bt.off.push_back(0);
			int64_t size = 0;
			while (size < opt->mini_batch_size) {
				if (!rd.next(name, comment, seq)) { more = false; break; }
				if (name.empty()) fprintf(stderr, "[WARNING]\033[1;31m empty sequence name in the input.\033[0m\n");
				bt.seqs += seq, bt.off.push_back((int64_t)bt.seqs.size()), bt.names.push_back(name);
				size += (int64_t)seq.size();
			}
			if (bt.names.empty()) break;
			grp.push_back(std::move(bt));
		}
		if (grp.empty()) break;
		const int32_t nb = (int32_t)grp.size();
		std::vector<mpa_qbatch_t> qb((size_t)nb);
		std::vector<const char *const *> names((size_t)nb);
		for (int32_t k = 0; k < nb; ++k) {
			Batch &bt = grp[k];
			for (const std::string &s : bt.names) bt.name_ptr.push_back(s.c_str());
			qb[k] = mpa_qbatch_t{ (int32_t)bt.names.size(), bt.seqs.data(), bt.off.data() };
			names[k] = bt.name_ptr.data();
		}
		std::vector<mpa_result_t*> res((size_t)nb, nullptr);
		std::vector<char*> text((size_t)nb, nullptr);
		std::vector<int64_t> tlen((size_t)nb, 0);
		if (mpa_map_batches_multi((int)ctxs.size(), ctxs.data(), c->impl, mo, nb, qb.data(), names.data(), n_threads > 1 ? n_threads : 1, res.data(), &id, text.data(), tlen.data()) != MPA_OK)
			die("mp_map_file");
		for (int32_t k = 0; k < nb; ++k) {
			fwrite(text[k], 1, (size_t)tlen[k], stdout);
			mpa_free(text[k]);
			mpa_result_destroy(res[k]);
			if (mp_verbose >= 3)
				fprintf(stderr, "[M::%s::%.3f*%.2f] mapped %d sequences\n", "worker_pipeline", mp_realtime(), mp_cputime() / mp_realtime(), qb[k].n_seq);
		}
	}
	return 0;
}

// ---- ns_global_gs16b / ns_global_gs32b: one call of the batched operator (mpa_dp_run / mpa_dp_run32) ---------------
} // extern "C"
static void global_one(const bool wide32, const char *who, void *km, const char *ns, int32_t nl, const char *as, int32_t al, const ns_opt_t *opt, const uint8_t *ss, ns_rst_t *r)
{
	r->n_cigar = 0, r->nt_len = nl, r->aa_len = al, r->score = INT32_MIN;   // (NS_GEN_PREPARE sets the lengths before its early return, nasw-sse.c:236)
	if (wide32 && nl == 0 && al > 0) {
		// the 32-bit function has no early return: with no rows it reports its initial H (-2^30) and ns_backtrack's tail (nasw-sse.c:77)
		r->score = -0x40000000;
		if ((opt->flag & NS_F_CIGAR) && !(opt->flag & (NS_F_EXT_LEFT | NS_F_EXT_RIGHT)) && al > 1) {
			if (r->m_cigar < 1 || !r->cigar) r->m_cigar = 1, r->cigar = (uint32_t*)(km && kmalloc ? kmalloc(km, sizeof(uint32_t)) : realloc(r->cigar, sizeof(uint32_t)));
			r->cigar[0] = (uint32_t)(al - 1) << 4 | 1, r->n_cigar = 1;
		}
		return;
	}
	if (nl <= 0 || al <= 0) return;
	if (opt->asize != 22 || (opt->nt4 && opt->nt4 != ns_tab_nt4) || (opt->aa20 && opt->aa20 != ns_tab_aa20) || (opt->codon && opt->codon != ns_tab_codon)) {
		set_error("ns_opt_t must use the library's own tables (ns_tab_nt4 / ns_tab_aa20 / ns_tab_codon) and asize 22");
		die(who);
	}
	if (km && !kmalloc) { set_error("a kalloc arena was passed but the program provides no kmalloc(); pass km = NULL"); die(who); }
	// the caller's nucleotides as a one-contig genome (ASCII or 0..4 codes: the table maps both, nasw-tab.c:93-95)
	std::vector<uint8_t> nt4((size_t)nl);
	for (int32_t i = 0; i < nl; ++i) nt4[i] = ns_tab_nt4[(uint8_t)ns[i]];
	const char *cname = "ns";
	const int64_t clen = nl;
	mpa_idx_t *mi = mpa_idx_from_nt4(1, &cname, &clen, nt4.data());
	if (ss) {                                             // the score bytes of the forward strand, indexed like the nucleotides
		mi->spsc.assign((size_t)nl * 2, 0xff);
		memcpy(mi->spsc.data(), ss, (size_t)nl);
	}
	mpa_dpopt_t dp;
	memset(&dp, 0, sizeof(dp));
	dp.go = opt->go, dp.ge = opt->ge, dp.fs = opt->fs, dp.xdrop = opt->xdrop, dp.end_bonus = opt->end_bonus, dp.ie_coef = opt->ie_coef;
	for (int k = 0; k < 6; ++k) dp.sp[k] = opt->sp[k];
	dp.sp_null_bonus = opt->sp_null_bonus;
	memcpy(dp.mat, opt->sc, 484);
	const bool is_ext = (opt->flag & (NS_F_EXT_LEFT | NS_F_EXT_RIGHT)) != 0;
	mpa_dp_task_t t;
	memset(&t, 0, sizeof(t));
	t.nt_off = 0, t.vid = 0, t.nl = nl, t.qid = 0, t.aa_off = 0, t.al = al, t.io = opt->io;
	t.flag = is_ext ? (opt->flag & (NS_F_EXT_LEFT | NS_F_EXT_RIGHT)) : MPA_F_CIGAR;   // the global sweep always keeps its traceback
	if (wide32 && !is_ext && !(opt->flag & NS_F_CIGAR)) t.flag = 0;                    // (the 32-bit sweep has a score-only mode: no traceback matrix)
	const int64_t q_off[2] = { 0, al };
	const mpa_qbatch_t q{ 1, as, q_off };
	mpa_dp_rst_t out;
	uint32_t *pool = nullptr;
	int64_t n_pool = 0;
	{
		std::lock_guard<std::mutex> g(g_mu);
		mpa_ctx_t *ctx = shared_ctx();
		if (mpa_idx_to_device(ctx, mi) != MPA_OK || (wide32 ? mpa_dp_run32(ctx, mi, &dp, &q, 1, &t, &out, &pool, &n_pool) : mpa_dp_run(ctx, mi, &dp, &q, 1, &t, &out, &pool, &n_pool)) != MPA_OK) {
			mpa_idx_destroy(mi);
			die(who);
		}
	}
	mpa_idx_destroy(mi);
	r->nt_len = out.nt_len, r->aa_len = out.aa_len, r->score = out.score;
	if (!is_ext && (opt->flag & NS_F_CIGAR) && out.n_cigar > 0) {
		if (out.n_cigar > r->m_cigar || !r->cigar) {             // (an arena block that is outgrown stays with its arena)
			r->m_cigar = out.n_cigar;
			r->cigar = (uint32_t*)(km ? kmalloc(km, sizeof(uint32_t) * (size_t)out.n_cigar) : realloc(r->cigar, sizeof(uint32_t) * (size_t)out.n_cigar));
		}
		memcpy(r->cigar, pool + out.cigar_off, sizeof(uint32_t) * (size_t)out.n_cigar);
		r->n_cigar = out.n_cigar;
	}
	mpa_free(pool);
}

extern "C" {
void ns_global_gs16b(void *km, const char *ns, int32_t nl, const char *as, int32_t al, const ns_opt_t *opt, const uint8_t *ss, ns_rst_t *r)
{
	global_one(false, "ns_global_gs16b", km, ns, nl, as, al, opt, ss, r);
}

// nasw.h:129,132 (nasw-sse.c:558-750): the 32-bit operator -- no extension mode, no end bonus (nasw.h:112-114); what exactly
// is reproduced (the reference's default x86 build of it is broken) is stated in gs32_core.h
void ns_global_gs32b(void *km, const char *ns, int32_t nl, const char *as, int32_t al, const ns_opt_t *opt, const uint8_t *ss, ns_rst_t *r)
{
	global_one(true, "ns_global_gs32b", km, ns, nl, as, al, opt, ss, r);
}

void ns_global_gs32(void *km, const char *ns, int32_t nl, const char *as, int32_t al, const ns_opt_t *opt, ns_rst_t *r)
{
	global_one(true, "ns_global_gs32b", km, ns, nl, as, al, opt, nullptr, r);
}

void ns_global_gs16(void *km, const char *ns, int32_t nl, const char *as, int32_t al, const ns_opt_t *opt, ns_rst_t *r)
{
	ns_global_gs16b(km, ns, nl, as, al, opt, nullptr, r);
}

} // extern "C"
