/* nasw_oracle.c -- scalar, column-sequential restatement of miniprot's spliced protein-to-genome DP.
 *
 * TEST INFRASTRUCTURE ONLY (see mporacle.h).  Parity PINNED against oracle/_ref (the compiled
 * reference) by tests/test_oracle.py (and tests/golden/dp_vectors.npz, made by tools/make_golden.py).
 *
 * What is restated: ns_global_gs16b (nasw-sse.c:340-551) in its three modes, the input preparation
 * ns_prep_seq / ns_prep_seq_left / ns_prep_nas / ns_gen_prof (nasw-sse.c:91-224), and the traceback
 * walk ns_backtrack + ns_fix_tiny_UV (nasw-sse.c:30-89).
 *
 * The reference is a Farrar-striped SSE2 kernel (8 x int16 lanes, segment length slen=ceil(al/8)).
 * This file has no SIMD and no striping: it walks columns c = 0 .. 8*slen-1 in order and reproduces
 * the two artefacts of striping that are observable in the output:
 *   (1) padded columns al <= c < 8*slen are live cells (they enter the row maximum);
 *   (2) the traceback word records the FIRST striped pass, in which the horizontal gap chain I
 *       restarts from -32768 at every lane start (c % slen == 0), plus one "raised by lazy-F" bit.
 * See SURVEY.md A.2/A.3.
 */
#include <stdlib.h>
#include <string.h>
#include <assert.h>
#include "mporacle.h"

#define NEG (-32768)

static inline int32_t sat(int32_t x) { return x < -32768 ? -32768 : x > 32767 ? 32767 : x; }
static inline int32_t adds(int32_t a, int32_t b) { return sat(a + b); }   /* _mm_adds_epi16 */
static inline int32_t subs(int32_t a, int32_t b) { return sat(a - b); }   /* _mm_subs_epi16 */
static inline int32_t mx(int32_t a, int32_t b) { return a > b ? a : b; }  /* _mm_max_epi16  */

/* ---- tables (nasw-tab.c:8-15,85-107) ---------------------------------------------------------- */

static const char aa_alphabet[] = "ARNDCQEGHILKMFPSTWYV*X";
static const uint8_t aa_reduce[22] = { 0, 2, 4, 4, 6, 5, 5, 8, 3, 10, 11, 2, 11, 12, 7, 1, 1, 13, 12, 10, 14, 15 };

uint8_t mpo_nt4(uint8_t c)
{
	switch (c) {
	case 0: case 'A': case 'a': return 0;
	case 1: case 'C': case 'c': return 1;
	case 2: case 'G': case 'g': return 2;
	case 3: case 'T': case 't': return 3;
	default: return 4;
	}
}

uint8_t mpo_aa20(uint8_t c)
{
	int i;
	if (c < 22) return c;                       /* already a code (nasw-tab.c:98, first assignment) */
	if (c >= 'a' && c <= 'z') c -= 32;
	for (i = 0; i < 22; ++i) if ((uint8_t)aa_alphabet[i] == c) return i;
	return 21;
}

uint8_t mpo_aa13(uint8_t c)
{
	int i;
	if (c < 22) return aa_reduce[c];
	if (c >= 'a' && c <= 'z') c -= 32;
	for (i = 0; i < 22; ++i) if ((uint8_t)aa_alphabet[i] == c) return aa_reduce[i];
	return 15;
}

const uint8_t *mpo_codon_std(void)
{
	/* The standard genetic code in the order AAA, AAC, AAG, AAT, ACA, ... (nasw-tab.c:22). */
	static uint8_t tab[64];
	static int ready = 0;
	if (!ready) {
		static const char first[]  = "TTTTTTTTTTTTTTTTCCCCCCCCCCCCCCCCAAAAAAAAAAAAAAAAGGGGGGGGGGGGGGGG";
		static const char second[] = "TTTTCCCCAAAAGGGGTTTTCCCCAAAAGGGGTTTTCCCCAAAAGGGGTTTTCCCCAAAAGGGG";
		static const char third[]  = "TCAGTCAGTCAGTCAGTCAGTCAGTCAGTCAGTCAGTCAGTCAGTCAGTCAGTCAGTCAGTCAG";
		static const char amino[]  = "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG"; /* NCBI transl_table=1 */
		int k;
		for (k = 0; k < 64; ++k)
			tab[mpo_nt4(first[k])<<4 | mpo_nt4(second[k])<<2 | mpo_nt4(third[k])] = mpo_aa20(amino[k]);
		ready = 1;
	}
	return tab;
}

void mpo_nasw_opt_default(mpo_nasw_opt_t *o, const int8_t *mat)
{
	memset(o, 0, sizeof(*o));
	o->go = 11, o->ge = 1, o->io = 29, o->fs = 17, o->xdrop = 100, o->end_bonus = 5;
	o->sp[0] = 8, o->sp[1] = 15, o->sp[2] = 21, o->sp[3] = 30, o->sp[4] = 4, o->sp[5] = 4; /* NS_S_MAMMAL */
	o->sp_null_bonus = -7;
	o->ie_coef = .5f;
	o->mat = mat;
	o->codon = 0;
}

/* ---- per-call input preparation ---------------------------------------------------------------- */

typedef struct {
	int32_t nl, al, slen, ncol;
	uint8_t *nas;        /* [nl]   aa code of the codon ENDING at row i (21='X' when undefined) */
	uint8_t *aas;        /* [al]   aa20 codes of the protein (reversed for left extension) */
	int8_t  *donor;      /* [nl+1] */
	int8_t  *acceptor;   /* [nl+1] */
} prep_t;

/* ns_prep_nas, nasw-sse.c:91-104 */
static void prep_nas(const uint8_t *nt, int32_t nl, const uint8_t *codon_tab, uint8_t *nas)
{
	int32_t i, run = 0;
	uint8_t codon = 0;
	for (i = 0; i < nl; ++i) {
		uint8_t c = mpo_nt4(nt[i]);
		nas[i] = 21;
		if (c < 4) {
			codon = (codon << 2 | c) & 0x3f;
			if (++run >= 3) nas[i] = codon_tab[codon];
		} else codon = 0, run = 0;
	}
}

/* ns_prep_seq, nasw-sse.c:106-155 (forward: global alignment and right extension) */
static void prep_forward(prep_t *p, const uint8_t *nt, const uint8_t *aa, const mpo_nasw_opt_t *o, const uint8_t *ss, const uint8_t *codon_tab)
{
	int32_t i, j, nl = p->nl, al = p->al;
	uint8_t *b = (uint8_t*)malloc(nl > 0 ? nl : 1);
	for (j = 0; j < al; ++j) p->aas[j] = mpo_aa20(aa[j]);
	for (i = 0; i < nl; ++i) b[i] = mpo_nt4(nt[i]);
	for (i = 0; i <= nl; ++i) p->donor[i] = p->acceptor[i] = (int8_t)o->sp[3];
	for (i = 0; i < nl - 3; ++i) {  /* donor[i]: penalty of an intron starting at i+1 */
		int32_t t = 3;
		if (b[i+1] == 2 && b[i+2] == 3)                       /* GT */
			t = (i + 3 < nl && (b[i+3] == 0 || b[i+3] == 2)) ? (b[i] == 2 ? -1 : 4) : 0;
		else if (b[i+1] == 2 && b[i+2] == 1 && b[i] == 2) t = 1; /* G|GC */
		else if (b[i+1] == 0 && b[i+2] == 3) t = 2;           /* AT */
		p->donor[i] = (int8_t)(t < 0 ? 0 : o->sp[t]);
	}
	for (i = 1; i < nl; ++i) {      /* acceptor[i]: penalty of an intron ending at i */
		int32_t t = 3, pen_y = 0;
		if (b[i-1] == 0 && b[i] == 2) {                        /* AG */
			t = (i >= 2 && (b[i-2] == 1 || b[i-2] == 3)) ? -1 : 0;
			for (j = i - 4; j >= 0 && j > i - 7; --j)
				if (b[j] != 1 && b[j] != 3) pen_y += o->sp[5];
		} else if (b[i-1] == 0 && b[i] == 1) t = 2;            /* AC */
		p->acceptor[i] = (int8_t)(t < 0 ? 0 : o->sp[t]);
		if (t == -1 || t == 0) p->acceptor[i] += pen_y;
	}
	if (ss) {                                                  /* nasw-sse.c:138-152 */
		int32_t max_spsc = (o->io + 1) / 2 - 1;
		for (i = 1; i < nl; ++i) {
			int32_t spsc = (int8_t)(ss[i]>>1) - (int8_t)64;
			if (spsc > max_spsc) spsc = max_spsc;
			if (ss[i] == 0xff) p->donor[i-1] -= o->sp_null_bonus, p->acceptor[i-1] -= o->sp_null_bonus;
			else if (ss[i] & 1) p->acceptor[i-1] -= spsc;
			else p->donor[i-1] -= spsc;
		}
	}
	prep_nas(nt, nl, codon_tab, p->nas);
	free(b);
}

/* ns_prep_seq_left, nasw-sse.c:157-210 (left extension: everything mirrored) */
static void prep_reverse(prep_t *p, const uint8_t *nt, const uint8_t *aa, const mpo_nasw_opt_t *o, const uint8_t *ss, const uint8_t *codon_tab)
{
	int32_t i, j, nl = p->nl, al = p->al;
	uint8_t *b = (uint8_t*)malloc(nl > 0 ? nl : 1), *fw = (uint8_t*)malloc(nl > 0 ? nl : 1);
	for (j = 0; j < al; ++j) p->aas[al - 1 - j] = mpo_aa20(aa[j]);
	for (i = 0; i < nl; ++i) b[nl - 1 - i] = mpo_nt4(nt[i]);
	for (i = 0; i <= nl; ++i) p->donor[i] = p->acceptor[i] = (int8_t)o->sp[3];
	for (i = 0; i < nl - 3; ++i) {  /* "donor" of the reversed string = acceptor signal read backwards */
		int32_t t = 3, pen_y = 0;
		if (b[i+1] == 2 && b[i+2] == 0) {                      /* GA = reverse of AG */
			t = (i + 3 < nl && (b[i+3] == 1 || b[i+3] == 3)) ? -1 : 0;
			for (j = i + 5; j < nl && j < i + 8; ++j)
				if (b[j] != 1 && b[j] != 3) pen_y += o->sp[5];
		} else if (b[i+1] == 1 && b[i+2] == 0) t = 2;          /* CA = reverse of AC */
		p->donor[i] = (int8_t)(t < 0 ? 0 : o->sp[t]);
		if (t == -1 || t == 0) p->donor[i] += pen_y;
	}
	for (i = 1; i < nl; ++i) {      /* "acceptor" of the reversed string = donor signal read backwards */
		int32_t t = 3;
		if (b[i-1] == 3 && b[i] == 2)                          /* TG = reverse of GT */
			t = (i >= 2 && (b[i-2] == 0 || b[i-2] == 2)) ? ((i + 1 < nl && b[i+1] == 2) ? -1 : 4) : 0;
		else if (b[i-1] == 1 && b[i] == 2 && i + 1 < nl && b[i+1] == 1) t = 1; /* CG|G */
		else if (b[i-1] == 3 && b[i] == 0) t = 2;              /* TA */
		p->acceptor[i] = (int8_t)(t < 0 ? 0 : o->sp[t]);
	}
	if (ss) {                                                  /* nasw-sse.c:189-203 */
		int32_t max_spsc = (o->io + 1) / 2 - 1;
		for (i = 0; i < nl; ++i) {
			int32_t spsc = (int8_t)(ss[i]>>1) - (int8_t)64;
			if (spsc > max_spsc) spsc = max_spsc;
			if (ss[i] == 0xff) p->donor[nl-i-1] -= o->sp_null_bonus, p->acceptor[nl-i-1] -= o->sp_null_bonus;
			else if (ss[i] & 1) p->donor[nl-i-1] -= spsc;
			else p->acceptor[nl-i-1] -= spsc;
		}
	}
	/* nas: forward codon ends, reversed, then delayed by two rows (nasw-sse.c:204-208) */
	prep_nas(nt, nl, codon_tab, fw);
	for (i = 0; i < nl; ++i) p->nas[i] = 21;
	for (i = 2; i < nl; ++i) p->nas[i] = fw[nl - 1 - (i - 2)];
	free(b); free(fw);
}

/* ns_log2, nasw-sse.c:330-338.  Compiled with -ffp-contract=off: the reference object has no FMA. */
static float log2_approx(float x)
{
	union { float f; uint32_t i; } z = { x };
	float r = (float)((int32_t)((z.i >> 23) & 255) - 128);
	z.i &= ~(255u << 23);
	z.i += 127u << 23;
	r += (-0.34484843f * z.f + 2.02466578f) * z.f - 0.67487759f;
	return r;
}

/* ---- CIGAR (nasw.h:141-152, nasw-sse.c:30-89) -------------------------------------------------- */

static void push_op(mpo_nasw_rst_t *r, uint32_t op, int32_t len)
{
	if (r->n_cigar > 0 && (r->cigar[r->n_cigar-1] & 0xf) == op && op != 10 && op != 11) {
		r->cigar[r->n_cigar-1] += (uint32_t)len << 4;
		return;
	}
	if (r->n_cigar == r->m_cigar) {
		r->m_cigar = r->m_cigar ? r->m_cigar * 2 : 16;
		r->cigar = (uint32_t*)realloc(r->cigar, sizeof(uint32_t) * r->m_cigar);
	}
	r->cigar[r->n_cigar++] = (uint32_t)len << 4 | op;
}

static void walk_back(const uint16_t *tb, int32_t nl, int32_t al, int32_t ncol, mpo_nasw_rst_t *r)
{
	enum { opM = 0, opI = 1, opD = 2, opN = 3, opF = 10, opG = 11, opU = 12, opV = 13 };
	int32_t i = nl - 1, j = al - 1, carry = 0, k;
	while (i >= 2 && j >= 0) {
		int32_t w = tb[(int64_t)i * ncol + j], st, ext;
		if (w >> 9 & 1) w = 1 | (w >> 4 << 4);      /* cell was raised by the cross-lane I: treat as I, keep ext bits */
		st = carry ? carry : (w & 0xf);
		ext = (st >= 1 && st <= 5) ? (w >> (st + 3) & 1) : 0;
		switch (st) {
		case 0: push_op(r, opM, 1); i -= 3; --j; break;
		case 1: push_op(r, opI, 1); --j; break;
		case 2: push_op(r, opD, 1); i -= 3; break;
		case 3: push_op(r, opN, 1); --i; break;
		case 4: push_op(r, opU, 1); --i; if (!ext) --j; break;
		case 5: push_op(r, opV, 1); --i; if (!ext) --j; break;
		case 6: push_op(r, opF, 1); --i; break;
		case 7: push_op(r, opF, 2); i -= 2; break;
		case 8: push_op(r, opG, 1); --i; --j; break;
		case 9: push_op(r, opG, 2); i -= 2; --j; break;
		default: break;
		}
		carry = (st >= 1 && st <= 5 && ext) ? st : 0;
	}
	if (j > 0) push_op(r, opI, j);                  /* nasw-sse.c:77 (sic: j, not j+1) */
	if (i >= 0) {                                   /* nasw-sse.c:78-82 */
		int32_t l = (i + 1) / 3 * 3, t = (i + 1) % 3;
		if (l > 0) push_op(r, opD, l);
		if (t != 0) push_op(r, opF, t);
	}
	for (k = 0; k < r->n_cigar >> 1; ++k) {
		uint32_t t = r->cigar[k];
		r->cigar[k] = r->cigar[r->n_cigar - 1 - k], r->cigar[r->n_cigar - 1 - k] = t;
	}
	for (k = 0; k < r->n_cigar; ++k) {              /* ns_fix_tiny_UV */
		uint32_t op = r->cigar[k] & 0xf;
		if ((op == opU || op == opV) && r->cigar[k] >> 4 < 3)
			r->cigar[k] = r->cigar[k] >> 4 << 4 | opG;
	}
}

/* ---- the DP ------------------------------------------------------------------------------------ */

void mpo_nasw(const uint8_t *nt, int32_t nl, const uint8_t *aa, int32_t al, const mpo_nasw_opt_t *o, const uint8_t *ss, mpo_nasw_rst_t *r)
{
	const uint8_t *codon_tab = o->codon ? o->codon : mpo_codon_std();
	const int32_t is_ext = !!(o->flag & (MPO_F_EXT_LEFT | MPO_F_EXT_RIGHT));
	const int32_t want_tb = (o->flag & MPO_F_CIGAR) && !is_ext;
	const int32_t slen = (al + 7) / 8, ncol = slen * 8;
	const int32_t go = (int16_t)o->go, ge = (int16_t)o->ge, goe = (int16_t)(o->go + o->ge), io = (int16_t)o->io, fs = (int16_t)o->fs;
	prep_t p;
	int16_t *prof, *buf, *H[4], *D[4], *A, *B, *C, *Hbest, *h1, *i1;
	uint16_t *tb = 0;
	int32_t i, c, a, max_sc = INT32_MIN, max_sc_log = INT32_MIN, max_i = -1, last = 0;

	r->n_cigar = 0, r->m_cigar = 0, r->cigar = 0;
	r->nt_len = nl, r->aa_len = al, r->score = INT32_MIN;      /* NS_GEN_PREPARE, nasw-sse.c:234 */
	p.nl = nl, p.al = al, p.slen = slen, p.ncol = ncol;
	p.nas = (uint8_t*)malloc(nl + 1); p.aas = (uint8_t*)malloc(al + 1);
	p.donor = (int8_t*)malloc(nl + 2); p.acceptor = (int8_t*)malloc(nl + 2);
	if (o->flag & MPO_F_EXT_LEFT) prep_reverse(&p, nt, aa, o, ss, codon_tab);
	else prep_forward(&p, nt, aa, o, ss, codon_tab);

	/* query profile, ns_gen_prof nasw-sse.c:212-224: padded columns score -32768 */
	prof = (int16_t*)malloc(sizeof(int16_t) * 22 * (ncol + 1));
	for (a = 0; a < 22; ++a)
		for (c = 0; c < ncol; ++c)
			prof[a * ncol + c] = c < al ? o->mat[a * 22 + p.aas[c]] : NEG;

	/* rows are 4-deep rings for H and D; every array has a slot for column -1 at index 0 */
	buf = (int16_t*)malloc(sizeof(int16_t) * (ncol + 1) * 14);
	for (c = 0; c < (ncol + 1) * 14; ++c) buf[c] = NEG;        /* NS_GEN_INIT1, nasw-sse.c:253-255 */
	for (a = 0; a < 4; ++a) H[a] = buf + (ncol + 1) * a + 1, D[a] = buf + (ncol + 1) * (4 + a) + 1;
	A = buf + (ncol + 1) * 8 + 1, B = buf + (ncol + 1) * 9 + 1, C = buf + (ncol + 1) * 10 + 1;
	Hbest = buf + (ncol + 1) * 11 + 1, h1 = buf + (ncol + 1) * 12 + 1, i1 = buf + (ncol + 1) * 13 + 1;
	if (want_tb) tb = (uint16_t*)calloc((size_t)(nl > 0 ? nl : 1) * ncol, sizeof(uint16_t));

	for (i = 2; i < nl; ++i) {
		/* ring positions: row i -> H[i&3]; rows i-1, i-2, i-3 likewise */
		int16_t *Hc = H[i & 3], *H1 = H[(i - 1) & 3], *H2 = H[(i - 2) & 3], *H3 = H[(i - 3) & 3];
		int16_t *Dc = D[i & 3], *D3 = D[(i - 3) & 3];
		const int16_t *S = prof + (int32_t)p.nas[i] * ncol;
		const int32_t gei = p.nas[i] == 20 ? fs : ge;             /* nasw-sse.c:263 */
		const int32_t dm1 = p.donor[i-1], d0 = p.donor[i], dp1 = p.donor[i+1];
		const int32_t a0 = p.acceptor[i], am1 = p.acceptor[i-1], am2 = p.acceptor[i-2];
		int32_t G = NEG, row_max = NEG, tmp_sc, tmp_sc_log, end_sc;
		/* virtual column -1 (nasw-sse.c:256-258, 266-271): only row 2 sees H(-1,-1)=0, H(0,-1)=H(1,-1)=-fs */
		if (i == 2) H3[-1] = 0, H2[-1] = (int16_t)-o->fs, H1[-1] = (int16_t)-o->fs;
		else H3[-1] = H2[-1] = H1[-1] = NEG;
		for (c = 0; c < ncol; ++c) {
			const int32_t lane_start = (c % slen == 0);
			const int32_t hl = lane_start ? NEG : h1[c-1], il = lane_start ? NEG : i1[c-1];
			int32_t h, t, u, v, st = 0, w = 0;
			/* 0: H(i-3,c-1) + s */
			h = adds(H3[c-1], S[c]);
			/* 1: I, FIRST-PASS (lane-segmented) version, nasw-sse.c:458-463 */
			t = subs(hl, go);
			if (il > t) w |= 1 << 4;
			t = subs(mx(t, il), ge);
			i1[c] = (int16_t)t;
			if (t > h) st = 1, h = t;
			/* 2: D */
			u = subs(H3[c], go), v = D3[c];
			if (v > u) w |= 1 << 5;
			t = subs(mx(u, v), gei);
			Dc[c] = (int16_t)t;
			if (t > h) st = 2, h = t;
			/* 3: A (phase-0 intron) */
			u = subs(H1[c], io), v = A[c];
			t = subs(u, dm1);
			if (v > t) w |= 1 << 6;
			t = mx(t, v); A[c] = (int16_t)t;
			t = subs(t, a0);
			if (t > h) st = 3, h = t;
			/* 4: B (phase-1 intron) */
			u = subs(H1[c-1], io), v = B[c];
			t = subs(u, d0);
			if (v > t) w |= 1 << 7;
			t = mx(t, v); B[c] = (int16_t)t;
			t = subs(t, am2);
			if (t > h) st = 4, h = t;
			/* 5: C (phase-2 intron) */
			v = C[c];
			t = subs(u, dp1);
			if (v > t) w |= 1 << 8;
			t = mx(t, v); C[c] = (int16_t)t;
			t = subs(t, am1);
			if (t > h) st = 5, h = t;
			/* 6..9: frameshifts */
			t = subs(H1[c], fs);   if (t > h) st = 6, h = t;
			t = subs(H2[c], fs);   if (t > h) st = 7, h = t;
			t = subs(H1[c-1], fs); if (t > h) st = 8, h = t;
			t = subs(H2[c-1], fs); if (t > h) st = 9, h = t;
			h1[c] = (int16_t)h;
			/* cross-lane I ("lazy-F", nasw-sse.c:409-422 / 522-537) expressed as a running value G */
			if (c < slen) G = NEG;
			else if (lane_start) G = mx(mx(subs(h1[c-1], goe), subs(i1[c-1], ge)), subs(G, ge));
			else G = subs(G, ge);
			if (G > h) w |= 1 << 9, h = G;
			Hc[c] = (int16_t)h;
			if (h > row_max) row_max = h;
			if (tb) tb[(int64_t)i * ncol + c] = (uint16_t)(w | st);
		}
		last = i;
		if (want_tb) continue;
		/* score-only bookkeeping, nasw-sse.c:423-433 */
		tmp_sc = row_max;
		end_sc = Hc[al - 1] + o->end_bonus;
		if (end_sc > tmp_sc) tmp_sc = end_sc;
		tmp_sc_log = tmp_sc - (i - al * 3 < 2 ? 0 : (int32_t)(o->ie_coef * log2_approx((float)(i - al * 3)) + .5f));
		if (tmp_sc_log > max_sc_log) {
			max_sc = tmp_sc, max_sc_log = tmp_sc_log, max_i = i;
			memcpy(Hbest, Hc, sizeof(int16_t) * ncol);
		}
		if (max_sc_log - tmp_sc_log > o->xdrop) break;
	}
	if (is_ext) {                                               /* nasw-sse.c:435-443 */
		for (c = 0; c < al; ++c) {
			int32_t sc = Hbest[c];
			if (c == al - 1) sc += o->end_bonus;
			if (sc == max_sc) break;
		}
		/* the reference assert()s c < al here; we report c+1 as is */
		r->nt_len = max_i + 1, r->aa_len = c + 1, r->score = max_sc;
	} else {
		r->score = last >= 2 ? H[last & 3][al - 1] : NEG;         /* nasw-sse.c:543 */
	}
	if (tb) { walk_back(tb, nl, al, ncol, r); free(tb); }
	free(buf); free(prof); free(p.nas); free(p.aas); free(p.donor); free(p.acceptor);
}

/* ---- pieces of the above on their own, for the tests of the 32-bit sweep (tests/test_gs32_model.py) -------------------------
 * mpo_prep_rec: the per-row inputs of one call packed the way the device's prep kernel hands them to the DP kernels (dp_device.h):
 * byte0 2*nas[i], byte1 donor[i+1], byte2 ge -- or fs on a stop codon --, byte3 acceptor[i]; aas_out[al] = the aa20 codes of
 * the protein in column order (reversed for a left extension). */
void mpo_prep_rec(const uint8_t *nt, int32_t nl, const uint8_t *aa, int32_t al, const mpo_nasw_opt_t *o, const uint8_t *ss, uint32_t *rec, uint8_t *aas_out)
{
	const uint8_t *codon_tab = o->codon ? o->codon : mpo_codon_std();
	prep_t p;
	int32_t i;
	p.nl = nl, p.al = al, p.slen = (al + 7) / 8, p.ncol = p.slen * 8;
	p.nas = (uint8_t*)malloc(nl + 1); p.aas = (uint8_t*)malloc(al + 1);
	p.donor = (int8_t*)malloc(nl + 2); p.acceptor = (int8_t*)malloc(nl + 2);
	if (o->flag & MPO_F_EXT_LEFT) prep_reverse(&p, nt, aa, o, ss, codon_tab);
	else prep_forward(&p, nt, aa, o, ss, codon_tab);
	for (i = 0; i < nl; ++i)
		rec[i] = (uint32_t)(p.nas[i] * 2) | (uint32_t)(uint8_t)p.donor[i + 1] << 8 | (uint32_t)((p.nas[i] == 20 ? o->fs : o->ge) & 0xff) << 16 | (uint32_t)(uint8_t)p.acceptor[i] << 24;
	memcpy(aas_out, p.aas, al);
	free(p.nas); free(p.aas); free(p.donor); free(p.acceptor);
}

/* ns_backtrack (nasw-sse.c:40-89) over a [nl][ncol] matrix of traceback words */
void mpo_walk_back(const uint16_t *tb, int32_t nl, int32_t al, int32_t ncol, mpo_nasw_rst_t *r)
{
	r->n_cigar = 0, r->m_cigar = 0, r->cigar = 0;
	walk_back(tb, nl, al, ncol, r);
}
