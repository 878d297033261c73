/* mpamd.h -- C ABI of the MI355X-native miniprot hot path (libmpamd.so).
 *
 * Plain C, plain pointers and sizes; no torch types.  Every entry point cites the interface of the
 * reference (lh3/miniprot v0.18-r281) that it replaces.  INTEGRATION.md shows the few lines a
 * miniprot maintainer adds to map.c / align.c to call these instead of the pthread + SSE path.
 *
 * Two boundaries, as in the reference:
 *   #2 operator level  -- mpa_dp_run():     a BATCH of ns_global_gs16b() calls  (nasw.h:135)
 *   #1 library level   -- mpa_map_batch():  a BATCH of mp_map() calls           (miniprot.h:268, map.c:264-271)
 *                         mpa_idx_restore(): mp_idx_restore()                    (miniprot.h:232, index.c:204)
 *                         mpa_format_paf():  mp_write_output() PAF branch        (format.c:333-358,453)
 *
 * The DP, the per-row splice/codon preparation and the traceback run on the GPU.  There is NO CPU
 * fallback: if no HIP device is usable every compute entry point returns MPA_ERR_NO_DEVICE.
 *
 * For CPU-only testing of the HOST logic the library-level path is also exposed as an explicit stage
 * machine (mpa_batch_*): the caller pulls DP tasks, executes them with any executor and pushes the
 * results back.  mpa_map_batch() is exactly that loop with mpa_dp_run() as the executor.
 */
#ifndef MPAMD_H
#define MPAMD_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPA_OK               0
#define MPA_ERR_NO_DEVICE   (-1)   /* no usable HIP device / kernels missing */
#define MPA_ERR_HIP         (-2)   /* a HIP runtime call failed (message via mpa_last_error) */
#define MPA_ERR_UNSUPPORTED (-3)   /* parameters outside what the kernels implement exactly */
#define MPA_ERR_IO          (-4)
#define MPA_ERR_ARG         (-5)

/* DP flags = NS_F_* (nasw.h:48-50) */
#define MPA_F_CIGAR      0x1
#define MPA_F_EXT_LEFT   0x2
#define MPA_F_EXT_RIGHT  0x4
#define MPA_F_SS_SKIP0   0x8   /* splice-score track only: nt_off is the first position of the window the reference cut its ss[] for
                                  (align.c:270-274); mp_ntseq_spsc_get() never sets the score of that position (ntseq.c:144-151) */

/* mapping flags = MP_F_* (miniprot.h:8-17) */
#define MPA_MF_NO_SPLICE    0x1
#define MPA_MF_NO_ALIGN     0x2
#define MPA_MF_SHOW_UNMAP   0x4
#define MPA_MF_GFF          0x8
#define MPA_MF_NO_PAF       0x10
#define MPA_MF_GTF          0x20
#define MPA_MF_NO_PRE_CHAIN 0x40
#define MPA_MF_SHOW_RESIDUE 0x80
#define MPA_MF_SHOW_TRANS   0x100
#define MPA_MF_NO_CS        0x200

/* ---- options: field-for-field mirrors of mp_idxopt_t / mp_mapopt_t (miniprot.h:36-77) ---------- */
typedef struct {
	int32_t bbit, min_aa_len, kmer, mod_bit;
	uint32_t trans_code;
} mpa_idxopt_t;                               /* also the .mpi header record, index.c:196 */

typedef struct {
	uint32_t flag;
	int64_t mini_batch_size;
	int32_t max_occ, max_gap, max_intron, min_max_intron, max_max_intron, bw, max_ext, max_ava;
	int32_t min_chn_cnt, max_chn_max_skip, max_chn_iter, min_chn_sc;
	float chn_coef_log, mask_level;
	int32_t mask_len;
	float pri_ratio, out_sim, out_cov;
	int32_t best_n, out_n, kmer2;
	int32_t go, ge, io, fs, io_end;
	float ie_coef;
	int32_t sp_model, sp_null_bonus, sp_max_bonus;
	float sp_scale;
	int32_t xdrop, end_bonus, asize, gff_delim, max_intron_flank;
	const char *gff_prefix;
	int8_t mat[484];
} mpa_mapopt_t;

void mpa_idxopt_init(mpa_idxopt_t *io);                         /* mp_idxopt_init, options.c:10 */
void mpa_mapopt_init(mpa_mapopt_t *mo);                         /* mp_mapopt_init, options.c:42 */
void mpa_mapopt_set_fs(mpa_mapopt_t *mo, int32_t fs);           /* mp_mapopt_set_fs, options.c:24 */
void mpa_mapopt_set_max_intron(mpa_mapopt_t *mo, int64_t gsize);/* mp_mapopt_set_max_intron, options.c:31 */

/* ---- device context -------------------------------------------------------------------------------- */
typedef struct mpa_ctx_s mpa_ctx_t;
/* device = HIP ordinal.  Returns NULL (and sets mpa_last_error) when there is no usable device. */
mpa_ctx_t *mpa_ctx_create(int device);
void mpa_ctx_destroy(mpa_ctx_t *ctx);
const char *mpa_last_error(void);
int mpa_device_count(void);

/* ---- index: mp_idx_t (miniprot.h:100-106) loaded from a .mpi file (index.c:189-229) ----------------- */
typedef struct mpa_idx_s mpa_idx_t;
mpa_idx_t *mpa_idx_restore(const char *fn);                     /* mp_idx_restore, index.c:204 */
/* genome-only index from nt4 codes (A0 C1 G2 T3 N4) held in memory, contigs concatenated; no k-mer table */
mpa_idx_t *mpa_idx_from_nt4(int32_t n_ctg, const char *const *names, const int64_t *lens, const uint8_t *nt4);
/* mp_idx_build (index.c:97-136): index of a FASTA genome (plain or gzip) with the given parameters */
mpa_idx_t *mpa_idx_build(const char *fasta_fn, const mpa_idxopt_t *io, int n_threads);
/* k-mer table of a genome-only index: build_worker + build_bidx (index.c:52-95) */
int mpa_idx_build_kmers(mpa_idx_t *mi, int n_threads);
/* the same table built on the GPU (scan kernel + one radix sort of all k-mer keys): same ki/kb, and kb stays resident for the
 * seeding kernels.  MPA_ERR_UNSUPPORTED when the parameters or the genome do not fit the device path (build on the host then). */
int mpa_idx_build_kmers_device(mpa_ctx_t *ctx, mpa_idx_t *mi);
/* genome-only index of a FASTA file with the given parameters (mp_ntseq_read, ntseq.c:29-75); build the table with either of the above */
mpa_idx_t *mpa_idx_read_fasta(const char *fasta_fn, const mpa_idxopt_t *io);
int mpa_idx_dump(const char *fn, const mpa_idx_t *mi);          /* mp_idx_dump, index.c:189: byte-identical file */
void mpa_idx_destroy(mpa_idx_t *mi);
int32_t mpa_idx_n_ctg(const mpa_idx_t *mi);
int64_t mpa_idx_ctg_len(const mpa_idx_t *mi, int32_t cid);
const char *mpa_idx_ctg_name(const mpa_idx_t *mi, int32_t cid);
int64_t mpa_idx_genome_len(const mpa_idx_t *mi);                /* nt->l_seq */
const mpa_idxopt_t *mpa_idx_opt(const mpa_idx_t *mi);
/* mp_ntseq_get_by_v (ntseq.c:108-114): nt4 codes of [st,en) on strand-oriented contig vid=cid<<1|rev */
int64_t mpa_idx_get_nt(const mpa_idx_t *mi, int32_t vid, int64_t st, int64_t en, uint8_t *out);
/* --spsc (mp_ntseq_read_spsc ntseq.c:234-296 + mp_set_spsc index.c:239-248): read splice scores "ctg offset +|- D|A score"
 * (plain or gzip) into a per-strand track, clamp them to the bound mp_set_spsc derives from *mo, and unless keep_io raise
 * mo->io / io_end by 10.  Call before mpa_idx_to_device().  Returns the number of scores kept or a negative error code. */
int64_t mpa_idx_set_spsc(mpa_idx_t *mi, const char *fn, mpa_mapopt_t *mo, int keep_io);
/* mp_ntseq_spsc_get_by_v (ntseq.c:153-159): the ss[] bytes of [st,en) on vid; 0xff = no score, else (score+64)<<1 | is_acceptor */
int64_t mpa_idx_get_spsc(const mpa_idx_t *mi, int32_t vid, int64_t st, int64_t en, uint8_t *out);
/* upload the packed genome (and later the k-mer index) to the context's device; idempotent */
int mpa_idx_to_device(mpa_ctx_t *ctx, mpa_idx_t *mi);

/* ---- boundary #2: batched spliced DP ------------------------------------------------------------------- */
/* Scalar parameters of ns_opt_t (nasw.h:62-73).  io is per task (io vs io_end, align.c:292,326). */
typedef struct {
	int32_t go, ge, fs, xdrop, end_bonus;
	int32_t sp[6];                  /* already scaled (align.c:59) */
	float ie_coef;
	int8_t mat[484];
	int32_t sp_null_bonus;          /* with a splice-score track: what a position without a score costs (nasw-sse.c:143-146) */
} mpa_dpopt_t;
void mpa_dpopt_from_mapopt(const mpa_mapopt_t *mo, mpa_dpopt_t *dp);  /* mp_map2ns_opt, align.c:50-60 */

/* One ns_global_gs16b() call (nasw.h:135).  The nucleotide side is a window of the device-resident
 * genome, the protein side a slice of a query of the batch. */
typedef struct {
	int64_t nt_off;                 /* window start, strand-local coordinate on vid */
	int32_t vid;                    /* contig<<1 | strand */
	int32_t nl;                     /* rows (window length) */
	int32_t qid;                    /* which query of the batch */
	int32_t aa_off, al;             /* protein slice */
	int32_t flag;                   /* MPA_F_* */
	int32_t io;                     /* intron-open penalty of this call */
	int32_t tag;                    /* opaque to the executor */
} mpa_dp_task_t;

/* ns_rst_t (nasw.h:75-80); the CIGAR lives in a shared pool at [cigar_off, cigar_off+n_cigar) */
typedef struct {
	int32_t nt_len, aa_len, score, n_cigar;
	int64_t cigar_off;
} mpa_dp_rst_t;

/* Queries of a batch as one concatenated ASCII buffer: query q = seqs[q_off[q] .. q_off[q+1]). */
typedef struct {
	int32_t n_seq;
	const char *seqs;
	const int64_t *q_off;           /* n_seq+1 entries */
} mpa_qbatch_t;

/* Run n DP calls on the GPU.  rst[n] is filled; *cigar_pool receives a malloc'd pool (caller frees with
 * mpa_free) holding every CIGAR as len<<4|op words (nasw.h:33-44).  The index must be on the device. */
int mpa_dp_run(mpa_ctx_t *ctx, const mpa_idx_t *mi, const mpa_dpopt_t *opt, const mpa_qbatch_t *q,
               int64_t n, const mpa_dp_task_t *tasks, mpa_dp_rst_t *rst, uint32_t **cigar_pool, int64_t *n_pool);
void mpa_free(void *p);

/* Statistics of the last mpa_dp_run() on this context (for bench.py's roofline record). */
typedef struct {
	int64_t n_ext, n_glob;          /* calls by mode */
	int64_t cells_ext, cells_glob;  /* padded DP cells (nl-2) * 8*ceil(al/8), SURVEY.md 8(d) */
	int64_t rows_prep;              /* rows of per-row records produced by the prep kernel */
	int64_t alg_bytes_ext, alg_bytes_glob; /* algorithmic HBM bytes, SURVEY.md 8(d) */
	double ms_prep, ms_ext, ms_glob, ms_backtrack, ms_total; /* HIP-event timings on the context's stream */
	int32_t launches_ext, launches_glob;
	/* the DP round kernel: k_dp_round (the default: one launch per DP round), or with MPA_DP_POOL=1 the k_dp_worker launches of the
	 * device's worker pool (one launch of workers per round; the workers of all lanes serve the units of all rounds in flight).  A round
	 * = every extension call + the traceback calls of the first traceback chunk. */
	int64_t cells_ext_round, cells_glob_round;   /* padded DP cells those launches swept */
	double ms_round;                             /* sum of their HIP-event durations */
	int32_t launches_round, pad_;
	double ms_round_union;                       /* worker pool: time during which at least one worker launch of the DEVICE was running
	                                              * (the union of the launches' intervals; the stream's rounds overlap) -- what one
	                                              * persistent kernel's duration would be.  0 unless MPA_DP_POOL=1. */
	/* checkpointed traceback (traceback calls of <= 64 columns and >= MPA_DP_LITE_MIN rows): calls and padded cells swept by the
	 * packed sweep, and the blocks of 96 rows whose traceback words the walk recomputed */
	int64_t n_ckpt, cells_ckpt, walk_blocks;
} mpa_dp_stats_t;
void mpa_dp_last_stats(const mpa_ctx_t *ctx, mpa_dp_stats_t *st);
/* sums over every mpa_dp_run() of this context since the last reset */
void mpa_dp_total_stats(mpa_ctx_t *ctx, mpa_dp_stats_t *st, int reset);
/* The same batch interface for the 32-bit operator, ns_global_gs32b (nasw.h:131-132, nasw-sse.c:558-750): global score of
 * every call and, for calls with MPA_F_CIGAR and no extension flag, its CIGAR; extension flags only select the reversed
 * preparation (ns_prep_seq_left), lengths come back as given -- the 32-bit operator has no extension mode (nasw.h:112-114).
 * Which behaviour of the reference this reproduces (its default x86 build of that function is broken): gs32_core.h. */
int mpa_dp_run32(mpa_ctx_t *ctx, const mpa_idx_t *mi, const mpa_dpopt_t *opt, const mpa_qbatch_t *q,
                 int64_t n, const mpa_dp_task_t *tasks, mpa_dp_rst_t *rst, uint32_t **cigar_pool, int64_t *n_pool);

/* Extension calls of 257..1024 columns are swept by two or four workgroups that hand column-block boundaries to each other
 * through HBM.  Should a hand-off not arrive within seconds (a stalled hardware queue), mpa_dp_run() repeats the round with
 * those calls on the one-wave path -- same results, no error.  How often that happened on this context: */
int64_t mpa_dp_handoff_retries(const mpa_ctx_t *ctx);

/* Bytes of device memory (HBM) this process holds through the library -- resident index (genome, occurrence lists, splice-score
 * track) plus every grow-only pool of every context -- and how often a pool had to be re-allocated (each growth is a hipFree,
 * which waits for the whole device).  What bench.py reports as hbm_resident_gb / pool_growth_events. */
int64_t mpa_device_bytes(void);
int64_t mpa_pool_growths(void);
/* Wall-clock milliseconds and calls of the stream pipeline's stages (mpa_map_batches) since the last reset, summed over the
 * stage's threads: [0] seeding (sketch, lookup, device seeding + both chaining rounds), [1] planning (regions, refinement, plans),
 * [2] the DP rounds of a batch, host round trips included, [3] output (statistics, flatten, text), [4] sketch (the host half of
 * seeding: protein sketch, bucket lookup, occurrence cut-off; [0] is then the device half).  ms / calls may be NULL. */
void mpa_stage_clocks(double ms[5], int64_t calls[5], int reset);

/* ---- boundary #1: batched mp_map() ------------------------------------------------------------------------ */
/* mp_feat_t / mp_extra_t / mp_reg1_t (miniprot.h:108-145) flattened for a C ABI */
typedef struct {
	int64_t vs, ve;
	int32_t qs, qe;
	int16_t type, phase;
	int32_t n_fs, n_stop, score, n_iden, blen;
	char donor[2], acceptor[2];
} mpa_feat_t;

typedef struct {
	int32_t qid;                    /* query index within the batch */
	int32_t id, parent, n_sub, subsc, cnt;
	int32_t n_exon, chn_sc, chn_sc_ungap;
	uint32_t vid;
	int32_t qs, qe;
	int64_t vs, ve;
	int32_t has_aln;                /* r->p != NULL */
	int32_t dp_score, dp_max, dp_max2, blen, n_fs, n_stop, dist_stop, dist_start, n_iden, n_plus;
	int32_t n_cigar, n_feat;
	int64_t cigar_off, feat_off;    /* into the result's cigar / feat pools */
} mpa_hit_t;

typedef struct mpa_result_s mpa_result_t;
int64_t mpa_result_n_hit(const mpa_result_t *r);
const mpa_hit_t *mpa_result_hits(const mpa_result_t *r);       /* grouped by qid ascending, in mp_map()'s order */
const int64_t *mpa_result_hit_off(const mpa_result_t *r);      /* n_seq+1 offsets into hits[] */
const uint32_t *mpa_result_cigars(const mpa_result_t *r);
const mpa_feat_t *mpa_result_feats(const mpa_result_t *r);
void mpa_result_destroy(mpa_result_t *r);

/* The batched replacement of kt_for(worker_for) (map.c:264-271,291): maps every query of the batch.
 * n_threads host threads run the non-DP stages; the DP runs on ctx's GPU. */
int mpa_map_batch(mpa_ctx_t *ctx, const mpa_idx_t *mi, const mpa_mapopt_t *opt, const mpa_qbatch_t *q,
                  int n_threads, mpa_result_t **out);

/* The same computation as an explicit stage machine (used by mpa_map_batch itself). */
typedef struct mpa_batch_s mpa_batch_t;
mpa_batch_t *mpa_batch_begin(const mpa_idx_t *mi, const mpa_mapopt_t *opt, const mpa_qbatch_t *q, int n_threads);
/* Returns the DP tasks of the next round (0 when mapping is complete); the array stays owned by the batch.  A batch has two
 * rounds: extensions (with the terminal-exon repeats of align.c:290-296,324-331 issued speculatively) + the gaps between anchors,
 * then the spans the extensions accepted. */
int64_t mpa_batch_dp_tasks(mpa_batch_t *b, const mpa_dp_task_t **tasks, mpa_dpopt_t *opt);
int mpa_batch_dp_results(mpa_batch_t *b, const mpa_dp_rst_t *rst, const uint32_t *cigar_pool);
mpa_result_t *mpa_batch_finish(mpa_batch_t *b);                 /* destroys the batch */

/* mp_write_output() for PAF (format.c:333-358, 453-473) + the output filters of map.c:298-311.
 * Appends the text of all queries of the batch to a malloc'd buffer (caller frees with mpa_free).
 * names[q] is the query name.  Returns the byte count. */
int64_t mpa_format_paf(const mpa_idx_t *mi, const mpa_mapopt_t *opt, const mpa_qbatch_t *q, const char *const *names,
                       const mpa_result_t *r, char **out);

/* mp_write_output() in full (format.c:453-473): PAF and/or GFF3 (MPA_MF_GFF, format.c:360-412) or GTF (MPA_MF_GTF,
 * format.c:414-451), plus the residue rows of --aln / --trans (MPA_MF_SHOW_RESIDUE / _TRANS, format.c:189-331), according to opt->flag.  *id_io is the running hit counter of the output file (map.c:306): pass the
 * same variable for every batch of a run, starting at 0.  The caller prints "##gff-version 3" once per file (map.c:338). */
int64_t mpa_format_output(const mpa_idx_t *mi, const mpa_mapopt_t *opt, const mpa_qbatch_t *q, const char *const *names,
                          const mpa_result_t *r, int64_t *id_io, char **out);

/* The number of hit records mpa_format_output() prints for this batch, i.e. by how much it advances *id_io.  A job whose batches
 * are mapped by different processes (one per GPU) exchanges these counts, and every process formats its batches with the
 * exclusive prefix sum as its starting id: the merged GFF3/GTF then carries the ids of a single-process run (map.c:306). */
int64_t mpa_result_n_output(const mpa_mapopt_t *opt, const mpa_qbatch_t *q, const mpa_result_t *r);

/* A stream of mini-batches (the loop of mp_map_file()'s kt_pipeline, map.c:273-345, which overlaps only file I/O with
 * mapping) as a software pipeline: sketch (host) | seeding + both chaining rounds (device) | regions, refinement (device), DP
 * plans | the two DP rounds of a batch (device; several batches at once on sibling contexts) | statistics + text (host), each
 * stage its own host thread(s).  results[k] is what mpa_map_batch() returns for
 * batches[k].  With names != NULL (names[k][q] = name of query q of batch k) the output text of every batch is produced
 * as well, as by mpa_format_output() with *id_io threaded through the batches in order: text[k] (mpa_free) / text_len[k]. */
int mpa_map_batches(mpa_ctx_t *ctx, const mpa_idx_t *mi, const mpa_mapopt_t *opt, int32_t n_batches, const mpa_qbatch_t *batches,
                    const char *const *const *names, int n_threads, mpa_result_t **results, int64_t *id_io, char **text, int64_t *text_len);

/* The same stream when the job's batches are shared between several callers (one process per GPU; the kt_for work stealing of
 * map.c:264-271 between processes): batches[] is the WHOLE job, claim(user) returns the index of the next batch this call
 * should map, or -1 once the job has none left; a caller takes a new batch whenever its pipeline has room for one, so a
 * faster GPU maps more of the job.  results / text / text_len hold what this call mapped, in the order it claimed them;
 * order[j] = index into batches[] of slot j, *n_mapped = how many.  The hit ids of *id_io run over this call's batches only
 * (GFF3/GTF jobs format after exchanging mpa_result_n_output() counts instead). */
typedef int32_t (*mpa_claim_fn)(void *user);
int mpa_map_batches_claim(mpa_ctx_t *ctx, const mpa_idx_t *mi, const mpa_mapopt_t *opt, int32_t n_batches, const mpa_qbatch_t *batches,
                          const char *const *const *names, int n_threads, mpa_result_t **results, int64_t *id_io, char **text, int64_t *text_len,
                          mpa_claim_fn claim, void *user, int32_t *n_mapped, int32_t *order);

/* The same job over SEVERAL GPUs of one process -- what mp_map_file()'s kt_for (map.c:264-271, 291) is across the threads of the
 * reference: one pipeline per device context (ctxs[0 .. n_ctx), each created on a different device with mpa_ctx_create), every
 * pipeline claiming the job's next mini-batch from a shared counter whenever it has room; the index is uploaded to each device
 * (mpa_idx_to_device is called for you).  results / text / text_len come back in INPUT order with the hit ids of a single
 * pipeline's run (GFF3 / GTF text is formatted after the mapping, from the prefix sum of mpa_result_n_output()).  n_threads is
 * shared out over the pipelines.  This is what the mp_map_file() of this library runs on every visible device (MPA_DEVICES). */
int mpa_map_batches_multi(int n_ctx, mpa_ctx_t *const *ctxs, mpa_idx_t *mi, const mpa_mapopt_t *opt, int32_t n_batches, const mpa_qbatch_t *batches,
                          const char *const *const *names, int n_threads, mpa_result_t **results, int64_t *id_io, char **text, int64_t *text_len);

const char *mpa_version(void);

/* ---- diagnostics (operator-level parity tests of the seeding kernels; not needed by a caller) ---------------------------- */
/* the k-mer hits (hash << 32 | window position) of the refinement scan (map.c:97-111) for n_win windows [as, as+len) on vid of one
 * query, from the device kernel (ctx != NULL) or the host stage (ctx == NULL): *out (mpa_free) holds them window by window,
 * first[n_win + 1] the boundaries.  Returns the number of hits or a negative error code. */
int64_t mpa_dbg_refine_hits(mpa_ctx_t *ctx, const mpa_idx_t *mi, int32_t kmer, const char *aa, int32_t l_aa, int32_t n_win, const int32_t *vid, const int64_t *as,
                            const int32_t *len, int64_t *first, uint64_t **out);
/* forward pass of mp_chain (chain.c:181-209) of n_prob problems (sorted anchors back to back in a[], first[n_prob + 1]): f and
 * pred (index inside the problem, -1 = none) of every anchor, from the device kernel (ctx != NULL) or the host pass (ctx == NULL) */
int mpa_dbg_chain_forward(mpa_ctx_t *ctx, int32_t max_dist_x, int32_t max_dist_y, int32_t bw, int32_t max_skip, int32_t max_iter, float coef_log,
                          int32_t is_spliced, int32_t kmer, int32_t bbit, int32_t n_prob, const int64_t *first, const uint64_t *a, int32_t *f, int32_t *pred);
/* the anchors that survive the pre-chain (map.c:186-192), query by query, from the host stage (ctx == NULL) or the device's */
int64_t mpa_dbg_anchors(const mpa_idx_t *mi, const mpa_mapopt_t *opt, const mpa_qbatch_t *q, int n_threads, int64_t *off, uint64_t **out);
int64_t mpa_dbg_prechain_survivors(mpa_ctx_t *ctx, const mpa_idx_t *mi, const mpa_mapopt_t *opt, const mpa_qbatch_t *q, int n_threads, int64_t *off, uint64_t **out);
/* the main chains of every query (what mp_chain() returns at map.c:195): u = score << 32 | anchors of each chain, a = the chains'
 * anchors; from the host stages (ctx == NULL) or from the device, which then runs both chaining rounds (k_chain_extract,
 * k_chain_fwd).  off_u / off_a [n_seq + 1], *out_u / *out_a (mpa_free).  Returns how many queries the device handed back to the
 * host, or a negative error code. */
/* Test model of the 32-bit sweep: the host instance of the source k_gs32 is compiled from (gs32_core.h), one call, inputs as
 * the device kernel gets them (per-row records, int16 profile [22][pw]); fills tb[nl][ncol] when not NULL, returns the score. */
int32_t mpa_dbg_gs32_model(int32_t nl, int32_t al, const uint32_t *rec, const int16_t *prof, int32_t pw, int32_t go, int32_t ge, int32_t io, int32_t fs,
                           uint16_t *tb, int32_t ncol);
/* (measurement) the 32-column class of score-only extension calls on the anti-diagonal prototype (dp_antidiag.hip:
 * lane j on row t - j, the mapping BASELINE.json's north_star names) instead of the row sweep: same results; tools/time_wide.py */
void mpa_dbg_antidiag(mpa_ctx_t *ctx, int on);
int64_t mpa_dbg_main_chains(mpa_ctx_t *ctx, const mpa_idx_t *mi, const mpa_mapopt_t *opt, const mpa_qbatch_t *q, int n_threads, int64_t *off_u, uint64_t **out_u,
                            int64_t *off_a, uint64_t **out_a);

#ifdef __cplusplus
}
#endif
#endif
