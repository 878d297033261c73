/* mpa_map.c -- a plain-C client of the C ABI (include/mpamd.h): what miniprot's own driver would do with the library.
 *
 *   mpa_map [-u] [-I] [--gff|--gtf|--aln|--trans] [--spsc FILE] [-K residues-per-batch] [-t threads] ref.mpi|ref.fa proteins.fa[.gz]
 *
 * Reads the proteins in mini-batches (mp_map_file reads mini_batch_size residues per batch, map.c:273-289), maps the whole
 * stream with one mpa_map_batches() call per 64 batches and prints what miniprot prints.  FASTA in, text out; no Python. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>
#include "mpamd.h"

typedef struct { char *seq; int64_t *off; char **names; int32_t n, m; int64_t l_seq, m_seq; } batch_t;

static void batch_push(batch_t *b, const char *name, const char *s, int64_t l)
{
	if (b->n == b->m) {
		b->m = b->m ? b->m * 2 : 256;
		b->off = (int64_t*)realloc(b->off, (b->m + 1) * sizeof(int64_t));
		b->names = (char**)realloc(b->names, b->m * sizeof(char*));
	}
	if (b->l_seq + l + 1 > b->m_seq) { b->m_seq = (b->l_seq + l + 1) * 2; b->seq = (char*)realloc(b->seq, b->m_seq); }
	if (b->n == 0) b->off[0] = 0;
	memcpy(b->seq + b->l_seq, s, l);
	b->l_seq += l;
	b->names[b->n] = strdup(name);
	b->off[++b->n] = b->l_seq;
}

/* one FASTA record at a time; returns 0 at end of file */
typedef struct { gzFile fp; char *line; size_t m_line; int have_header; char name[1024]; char *seq; int64_t l, m; } reader_t;
static int next_line(reader_t *r)
{
	size_t l = 0;
	for (;;) {
		if (r->m_line < l + 4096) r->line = (char*)realloc(r->line, r->m_line = (l + 4096) * 2);
		if (!gzgets(r->fp, r->line + l, (int)(r->m_line - l))) return l > 0;
		l += strlen(r->line + l);
		if (l && r->line[l - 1] == '\n') { r->line[--l] = 0; if (l && r->line[l - 1] == '\r') r->line[--l] = 0; return 1; }
	}
}
static int next_record(reader_t *r)
{
	if (!r->have_header) {
		while (next_line(r)) if (r->line[0] == '>') { r->have_header = 1; break; }
		if (!r->have_header) return 0;
	}
	{
		size_t k = 1, j = 0;
		while (r->line[k] && r->line[k] != ' ' && r->line[k] != '\t' && j + 1 < sizeof(r->name)) r->name[j++] = r->line[k++];
		r->name[j] = 0;
	}
	r->l = 0, r->have_header = 0;
	while (next_line(r)) {
		if (r->line[0] == '>') { r->have_header = 1; break; }
		size_t l = strlen(r->line);
		if (r->l + (int64_t)l + 1 > r->m) r->seq = (char*)realloc(r->seq, r->m = (r->l + l + 1) * 2);
		memcpy(r->seq + r->l, r->line, l);
		r->l += l;
	}
	return 1;
}

int main(int argc, char *argv[])
{
	mpa_idxopt_t io;
	mpa_mapopt_t mo;
	int i, n_threads = 4, set_I = 0, device = 0;
	int64_t batch_res = 2000000, id = 0;
	const char *fn_spsc = 0;
	mpa_idxopt_init(&io);
	mpa_mapopt_init(&mo);
	for (i = 1; i < argc && argv[i][0] == '-'; ++i) {
		if (!strcmp(argv[i], "-u")) mo.flag |= MPA_MF_SHOW_UNMAP;
		else if (!strcmp(argv[i], "-I")) set_I = 1;
		else if (!strcmp(argv[i], "--gff")) mo.flag |= MPA_MF_GFF;
		else if (!strcmp(argv[i], "--gtf")) mo.flag |= MPA_MF_GTF;
		else if (!strcmp(argv[i], "--aln")) mo.flag |= MPA_MF_SHOW_RESIDUE;
		else if (!strcmp(argv[i], "--trans")) mo.flag |= MPA_MF_SHOW_TRANS;
		else if (!strcmp(argv[i], "--spsc") && i + 1 < argc) fn_spsc = argv[++i];
		else if (!strcmp(argv[i], "-K") && i + 1 < argc) batch_res = atol(argv[++i]);
		else if (!strcmp(argv[i], "-t") && i + 1 < argc) n_threads = atoi(argv[++i]);
		else if (!strcmp(argv[i], "-d") && i + 1 < argc) device = atoi(argv[++i]);
		else { fprintf(stderr, "unknown option %s\n", argv[i]); return 1; }
	}
	if (argc - i < 2) { fprintf(stderr, "Usage: mpa_map [options] ref.mpi proteins.fa\n"); return 1; }
	mpa_ctx_t *ctx = mpa_ctx_create(device);
	if (!ctx) { fprintf(stderr, "[ERROR] %s\n", mpa_last_error()); return 1; }     /* no GPU, no mapping: there is no CPU fallback */
	mpa_idx_t *mi = mpa_idx_restore(argv[i]);
	if (!mi) { fprintf(stderr, "[ERROR] %s\n", mpa_last_error()); return 1; }
	if (set_I) mpa_mapopt_set_max_intron(&mo, mpa_idx_genome_len(mi));
	if (fn_spsc && mpa_idx_set_spsc(mi, fn_spsc, &mo, 0) < 0) { fprintf(stderr, "[ERROR] %s\n", mpa_last_error()); return 1; }
	if (mpa_idx_to_device(ctx, mi) != MPA_OK) { fprintf(stderr, "[ERROR] %s\n", mpa_last_error()); return 1; }
	if (mo.flag & MPA_MF_GFF) puts("##gff-version 3");

	reader_t rd;
	memset(&rd, 0, sizeof(rd));
	rd.fp = gzopen(argv[i + 1], "rb");
	if (!rd.fp) { fprintf(stderr, "[ERROR] cannot open %s\n", argv[i + 1]); return 1; }
	enum { MAX_B = 64 };
	int more = 1;
	while (more) {
		batch_t b[MAX_B];
		int n_b = 0, k;
		memset(b, 0, sizeof(b));
		while (n_b < MAX_B && more) {                        /* fill up to MAX_B mini-batches of ~batch_res residues */
			batch_t *cur = &b[n_b];
			while ((more = next_record(&rd)) != 0) {
				batch_push(cur, rd.name, rd.seq, rd.l);
				if (cur->l_seq >= batch_res) break;
			}
			if (cur->n > 0) ++n_b;
		}
		if (n_b == 0) break;
		mpa_qbatch_t qb[MAX_B];
		const char *const *names[MAX_B];
		mpa_result_t *res[MAX_B];
		char *text[MAX_B];
		int64_t len[MAX_B];
		for (k = 0; k < n_b; ++k) qb[k].n_seq = b[k].n, qb[k].seqs = b[k].seq, qb[k].q_off = b[k].off, names[k] = (const char *const*)b[k].names;
		if (mpa_map_batches(ctx, mi, &mo, n_b, qb, names, n_threads, res, &id, text, len) != MPA_OK) { fprintf(stderr, "[ERROR] %s\n", mpa_last_error()); return 1; }
		for (k = 0; k < n_b; ++k) {
			int32_t j;
			fwrite(text[k], 1, len[k], stdout);
			mpa_free(text[k]);
			mpa_result_destroy(res[k]);
			for (j = 0; j < b[k].n; ++j) free(b[k].names[j]);
			free(b[k].names), free(b[k].off), free(b[k].seq);
		}
	}
	gzclose(rd.fp);
	free(rd.line), free(rd.seq);
	mpa_idx_destroy(mi);
	mpa_ctx_destroy(ctx);
	return 0;
}
