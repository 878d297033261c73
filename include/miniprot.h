/* miniprot.h -- source- and binary-compatible declarations of lh3/miniprot's library interface (v0.18-r281),
 * implemented by libmpamd.so on top of the batched mpa_* C ABI (include/mpamd.h).
 *
 * A program written against the reference's miniprot.h (its example.c, its main.c, third-party callers) compiles
 * against THIS header and links with -lmpamd instead of libminiprot.a.  Everything a caller can see of the reference's
 * ABI is reproduced: the option records, the index view (contig table, k-mer table), the hit records returned by
 * mp_map() and who frees them.  The structs are plain data whose layout IS the interface (sizes on LP64: mp_idxopt_t
 * 20 -- also the .mpi header record --, mp_mapopt_t 656, mp_feat_t 56, mp_extra_t 48 + CIGAR, mp_reg1_t 104), so the
 * field order below follows miniprot.h:36-145 of the reference; tests/test_compat_abi.py checks the sizes and offsets.
 *
 * What differs behind the interface: mp_map() is a batch of one through the GPU mapper, mp_map_file() streams
 * mini-batches through mpa_map_batches().  No usable HIP device: a message on stderr and abort() -- there is no
 * CPU fallback.
 */
/*
  The interface declared here -- record layouts, constants and prototypes -- is that of lh3/miniprot (v0.18-r281), reproduced
  so that programs written against the reference's header compile and link against this library unchanged.  miniprot is
  distributed under the following licence, whose notice is retained as it requires:

  The MIT License

  Copyright (c) 2022-     Dana-Farber Cancer Institute

  Permission is hereby granted, free of charge, to any person obtaining
  a copy of this software and associated documentation files (the
  "Software"), to deal in the Software without restriction, including
  without limitation the rights to use, copy, modify, merge, publish,
  distribute, sublicense, and/or sell copies of the Software, and to
  permit persons to whom the Software is furnished to do so, subject to
  the following conditions:

  The above copyright notice and this permission notice shall be
  included in all copies or substantial portions of the Software.

  THE SOFTWARE IS PROVIDED "AS IS", WITHOUT WARRANTY OF ANY KIND,
  EXPRESS OR IMPLIED, INCLUDING BUT NOT LIMITED TO THE WARRANTIES OF
  MERCHANTABILITY, FITNESS FOR A PARTICULAR PURPOSE AND
  NONINFRINGEMENT. IN NO EVENT SHALL THE AUTHORS OR COPYRIGHT HOLDERS
  BE LIABLE FOR ANY CLAIM, DAMAGES OR OTHER LIABILITY, WHETHER IN AN
  ACTION OF CONTRACT, TORT OR OTHERWISE, ARISING FROM, OUT OF OR IN
  CONNECTION WITH THE SOFTWARE OR THE USE OR OTHER DEALINGS IN THE
  SOFTWARE.
*/
#ifndef MINIPROT_H
#define MINIPROT_H

#include <stdint.h>

#define MP_VERSION "0.18-r281"     /* the reference version whose behaviour is reproduced */

/* mp_mapopt_t::flag */
#define MP_F_NO_SPLICE    0x1
#define MP_F_NO_ALIGN     0x2
#define MP_F_SHOW_UNMAP   0x4
#define MP_F_GFF          0x8
#define MP_F_NO_PAF       0x10
#define MP_F_GTF          0x20
#define MP_F_NO_PRE_CHAIN 0x40
#define MP_F_SHOW_RESIDUE 0x80
#define MP_F_SHOW_TRANS   0x100
#define MP_F_NO_CS        0x200

#define MP_FEAT_CDS   0
#define MP_FEAT_STOP  1

#define MP_BITS_PER_AA 4
#define MP_BLOCK_BONUS 2
#define MP_CODON_STD   1
#define MP_IDX_MAGIC   "MPI\3"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- plain-data records (layout = ABI) -------------------------------------------------------- */

typedef struct { uint64_t x, y; } mp128_t;
typedef struct { int32_t n, m; mp128_t *a; } mp128_v;
typedef struct { int32_t n, m; uint64_t *a; } mp64_v;

typedef struct {                      /* indexing parameters; stored verbatim in the .mpi header */
	int32_t bbit, min_aa_len, kmer, mod_bit;
	uint32_t trans_code;
} mp_idxopt_t;

typedef struct {                      /* mapping parameters; same record as mpa_mapopt_t */
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
} mp_mapopt_t;

typedef struct { uint32_t n, m; uint64_t *a; } mp_spsc_t;     /* per-strand splice scores; unused by this library (dense track) */
typedef struct { int64_t off, len; char *name; } mp_ctg_t;

typedef struct {                      /* the packed genome: 4 bits per base, even offsets in the low nibble */
	int32_t n_ctg, m_ctg, l_name;
	int64_t l_seq, m_seq;
	uint8_t *seq;
	mp_ctg_t *ctg;
	char *name;
	void *h;
	mp_spsc_t *spsc;
} mp_ntdb_t;

typedef struct {                      /* read-only view of the index; owned by the library */
	mp_idxopt_t opt;
	uint32_t n_block;
	mp_ntdb_t *nt;
	int64_t n_kb, *ki;
	uint32_t *bo, *kb;
} mp_idx_t;

typedef struct {                      /* alignment details of a hit; malloc'd, freed by the caller */
	int32_t dp_score, dp_max, dp_max2;
	int32_t n_cigar, m_cigar;
	int32_t blen, n_fs, n_stop, dist_stop, dist_start, n_iden, n_plus;
	uint32_t cigar[];
} mp_extra_t;

typedef struct {                      /* one exon (or the stop codon) of a hit */
	int64_t vs, ve;
	int32_t qs, qe;
	int16_t type, phase;
	int32_t n_fs, n_stop, score, n_iden, blen;
	char donor[2], acceptor[2];
} mp_feat_t;

typedef struct {                      /* one hit */
	int32_t off, cnt, id, parent, n_sub, subsc;
	int32_t n_feat, m_feat, n_exon, chn_sc, chn_sc_ungap;
	uint32_t hash, vid;               /* vid = contig << 1 | reverse strand */
	int32_t qs, qe;
	int64_t vs, ve;                   /* strand-local coordinates */
	uint64_t *a;                      /* never valid after mp_map() returns (NULL here) */
	mp_feat_t *feat;                  /* malloc'd, freed by the caller */
	mp_extra_t *p;                    /* malloc'd, freed by the caller */
} mp_reg1_t;

typedef struct mp_tbuf_s mp_tbuf_t;

/* ---- globals ------------------------------------------------------------------------------------ */
extern int32_t mp_verbose, mp_dbg_flag;
extern char *ns_tab_nt_i2c, *ns_tab_aa_i2c;
extern uint8_t ns_tab_a2r[22], ns_tab_nt4[256], ns_tab_aa20[256], ns_tab_aa13[256];
extern uint8_t ns_tab_codon[64], ns_tab_codon13[64];

/* ---- functions (reference: miniprot.h:158-286) -------------------------------------------------- */
void mp_start(void);                                                       /* tables for genetic code 1 + the timer */
void mp_idxopt_init(mp_idxopt_t *io);
void mp_mapopt_init(mp_mapopt_t *mo);
void mp_mapopt_set_fs(mp_mapopt_t *mo, int32_t fs);
void mp_mapopt_set_max_intron(mp_mapopt_t *mo, int64_t gsize);
int32_t mp_mapopt_check(const mp_mapopt_t *mo);

mp_idx_t *mp_idx_load(const char *fn, const mp_idxopt_t *io, int32_t n_threads);   /* .mpi file, or FASTA (built) */
mp_idx_t *mp_idx_restore(const char *fn);
int mp_idx_dump(const char *fn, const mp_idx_t *mi);
void mp_idx_destroy(mp_idx_t *mi);
void mp_idx_print_stat(const mp_idx_t *mi, int32_t max_occ);

int32_t mp_ntseq_read_spsc(mp_ntdb_t *nt, const char *fn, int32_t max_sc);
void mp_set_spsc(const char *fn, mp_idx_t *mi, mp_mapopt_t *mo, int32_t keep_io);

/* a thread buffer owns a device context (HIP streams + device pools), taken from a pool by its first mp_map() and parked
 * there again by mp_tbuf_destroy(): creating and destroying buffers per mini-batch, as map.c does, costs nothing extra */
mp_tbuf_t *mp_tbuf_init(void);
void mp_tbuf_destroy(mp_tbuf_t *b);
/* all hits of one protein.  The array and every ->p / ->feat are malloc'd; the caller frees them. */
mp_reg1_t *mp_map(const mp_idx_t *mi, int qlen, const char *seq, int *n_reg, mp_tbuf_t *b, const mp_mapopt_t *opt, const char *qname);
/* map a FASTA/FASTQ file of proteins and print PAF/GFF3/GTF to stdout */
int32_t mp_map_file(const mp_idx_t *idx, const char *fn, const mp_mapopt_t *opt, int n_threads);

/* timers used by the reference's main.c (sys.c) */
double mp_realtime(void);
double mp_cputime(void);
long mp_peakrss(void);
double mp_percent_cpu(void);

#ifdef __cplusplus
}
#endif
#endif
