/* nasw.h -- the reference's DP operator interface (lh3/miniprot v0.18-r281, nasw.h:33-152) as exported by libmpamd.so.
 *
 * ns_global_gs16b() here is ONE call of the batched GPU operator mpa_dp_run() (include/mpamd.h): the nucleotide buffer
 * is uploaded as a one-contig genome, the three kernels (row records, extension or traceback sweep, backtrack) run on
 * the default device and the result comes back in the reference's ns_rst_t.  Results are bit-identical to the
 * reference's SSE2 kernel, including the traceback's striping artefacts.  It exists for callers of the operator ABI;
 * throughput comes from batching (mpa_dp_run / mpa_map_batches), not from this entry point: every call builds and uploads a
 * one-contig index of its own and holds the library's global lock while it runs (a compatibility shim, not a fast path).
 *
 * Memory: ns_rst_t::cigar is allocated with malloc() when km == NULL.  A non-NULL km is a kalloc arena of the caller;
 * it is honoured when the program itself provides kmalloc() (weak reference), otherwise the call aborts with a message.
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
#ifndef NASW_H
#define NASW_H

#include <stdint.h>
#include <string.h>
#include <stdlib.h>

/* CIGAR operators: word = length << 4 | operator */
#define NS_CIGAR_M 0
#define NS_CIGAR_I 1
#define NS_CIGAR_D 2
#define NS_CIGAR_N 3     /* intron between codons */
#define NS_CIGAR_F 10    /* frameshift deletion */
#define NS_CIGAR_G 11    /* frameshift match */
#define NS_CIGAR_U 12    /* intron after the first base of a codon */
#define NS_CIGAR_V 13    /* intron after the second base of a codon */
#define NS_CIGAR_E 14
#define NS_CIGAR_STR "MIDNSHP=XBFGUVE"

/* ns_opt_t::flag */
#define NS_F_CIGAR     0x1
#define NS_F_EXT_LEFT  0x2
#define NS_F_EXT_RIGHT 0x4

/* splice models for ns_opt_set_sp() */
#define NS_S_NONE    0
#define NS_S_GENERIC 1
#define NS_S_MAMMAL  2

#define NS_SPSC_OFFSET 64

#ifdef __cplusplus
extern "C" {
#endif

extern char *ns_tab_nt_i2c, *ns_tab_aa_i2c;
extern uint8_t ns_tab_a2r[22], ns_tab_nt4[256], ns_tab_aa20[256], ns_tab_aa13[256];
extern uint8_t ns_tab_codon[64], ns_tab_codon13[64];
extern int8_t ns_mat_blosum62[484];

typedef struct {                      /* 96 bytes on LP64; field order = ABI */
	int32_t flag;
	int32_t go, ge, io, fs;
	int32_t xdrop, end_bonus;
	int32_t asize;                    /* 22 */
	int32_t sp[6];                    /* splice-signal penalties: pos3, GC-AG, AT-AC, other, pos0, poly-Y */
	int32_t sp_null_bonus;
	float ie_coef;
	const int8_t *sc;                 /* asize x asize scores */
	uint8_t *nt4, *aa20, *codon;      /* must be the library's own tables (ns_tab_*) */
} ns_opt_t;

typedef struct {                      /* 32 bytes */
	int32_t n_cigar, m_cigar;
	int32_t nt_len, aa_len;
	int32_t score;
	uint32_t *cigar;
} ns_rst_t;

int ns_make_tables(int codon_type);                     /* NCBI genetic code; <0 if undefined.  Not thread safe. */
void ns_opt_init(ns_opt_t *opt);
void ns_opt_set_sp(ns_opt_t *opt, int32_t model);
void ns_set_stop_sc(int32_t asize, int8_t *mat, int8_t score);

/* ns: ASCII or 0..4 codes; as: ASCII.  ss: optional splice-score bytes, one per nucleotide (0xff = none). */
void ns_global_gs16b(void *km, const char *ns, int32_t nl, const char *as, int32_t al, const ns_opt_t *opt, const uint8_t *ss, ns_rst_t *r);
void ns_global_gs16(void *km, const char *ns, int32_t nl, const char *as, int32_t al, const ns_opt_t *opt, ns_rst_t *r);
/* the 32-bit operator (nasw.h:129,132): global score and CIGAR without int16 saturation; no extension mode, no end bonus
 * (nasw.h:112-114).  One wavefront per call on the GPU (miniprot_amd/csrc/gs32_core.h says which build of the reference's
 * function it reproduces bit for bit). */
void ns_global_gs32(void *km, const char *ns, int32_t nl, const char *as, int32_t al, const ns_opt_t *opt, ns_rst_t *r);
void ns_global_gs32b(void *km, const char *ns, int32_t nl, const char *as, int32_t al, const ns_opt_t *opt, const uint8_t *ss, ns_rst_t *r);

/* append one operation to a malloc'd CIGAR (frameshift operators are never merged) */
static inline uint32_t *ns_push_cigar(void *km, int32_t *n_cigar, int32_t *m_cigar, uint32_t *cigar, uint32_t op, int32_t len)
{
	(void)km;
	if (*n_cigar > 0 && op == (cigar[*n_cigar - 1] & 0xf) && op != NS_CIGAR_F && op != NS_CIGAR_G) {
		cigar[*n_cigar - 1] += (uint32_t)len << 4;
		return cigar;
	}
	if (*n_cigar == *m_cigar) {
		*m_cigar += (*m_cigar >> 1) + 8;
		cigar = (uint32_t*)realloc(cigar, sizeof(uint32_t) * (size_t)*m_cigar);
	}
	cigar[(*n_cigar)++] = (uint32_t)len << 4 | op;
	return cigar;
}

static inline void ns_rst_init(ns_rst_t *r) { memset(r, 0, sizeof(*r)); }

#ifdef __cplusplus
}
#endif
#endif
