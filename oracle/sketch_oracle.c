/* sketch_oracle.c -- restatement of miniprot's k-mer sketching and of its radix sort.
 *
 * TEST INFRASTRUCTURE ONLY (see mporacle.h).  Parity PINNED against oracle/_ref.
 *
 * Follows sketch.c:7-100 (hash, protein sketch, 3-frame ORF sketch) and ksort.h:109-162
 * (MSD in-place radix sort whose tie order for equal keys is part of the observable behaviour,
 * SURVEY.md A.5).
 */
#include <stdlib.h>
#include <string.h>
#include "mporacle.h"

/* ---- sort (ksort.h:109-162) -------------------------------------------------------------------- */

/* Elements are esz bytes; the sort key is the first 8 bytes (little-endian uint64). */
static inline uint64_t key_of(const uint8_t *p) { uint64_t k; memcpy(&k, p, 8); return k; }

static void insertion(uint8_t *beg, uint8_t *end, int esz)   /* rs_insertsort_*, stable */
{
	uint8_t tmp[16], *i, *j;
	for (i = beg + esz; i < end; i += esz) {
		if (key_of(i) < key_of(i - esz)) {
			memcpy(tmp, i, esz);
			for (j = i; j > beg && key_of(tmp) < key_of(j - esz); j -= esz)
				memcpy(j, j - esz, esz);
			memcpy(j, tmp, esz);
		}
	}
}

/* rs_sort_*: one MSD digit (8 bits at bit offset `shift`), American-flag in-place permutation.
 * The order in which the cycle leader walks buckets decides where equal keys end up, so the walk
 * below is the same walk: bucket by bucket from digit 0, always chasing the element just displaced. */
static void flag_pass(uint8_t *beg, uint8_t *end, int esz, int shift)
{
	uint8_t *head[256], *tail[256], hold[16], out[16];
	int64_t cnt[256];
	int d, k;
	uint8_t *p;
	memset(cnt, 0, sizeof(cnt));
	for (p = beg; p != end; p += esz) ++cnt[key_of(p) >> shift & 255];
	for (d = 0, p = beg; d < 256; ++d) head[d] = p, p += cnt[d] * esz, tail[d] = p;
	for (k = 0; k < 256;) {
		if (head[k] == tail[k]) { ++k; continue; }
		d = (int)(key_of(head[k]) >> shift & 255);
		if (d == k) { head[k] += esz; continue; }
		memcpy(hold, head[k], esz);
		do {
			memcpy(out, hold, esz);
			memcpy(hold, head[d], esz);
			memcpy(head[d], out, esz);
			head[d] += esz;
			d = (int)(key_of(hold) >> shift & 255);
		} while (d != k);
		memcpy(head[k], hold, esz);
		head[k] += esz;
	}
	if (shift) {
		int next = shift > 8 ? shift - 8 : 0;
		for (d = 0, p = beg; d < 256; ++d) {
			uint8_t *e = p + cnt[d] * esz;
			if (cnt[d] > 64) flag_pass(p, e, esz, next);
			else if (cnt[d] > 1) insertion(p, e, esz);
			p = e;
		}
	}
}

static void radix_any(uint8_t *beg, uint8_t *end, int esz)
{
	if ((end - beg) / esz <= 64) insertion(beg, end, esz);
	else flag_pass(beg, end, esz, 56);
}

void mpo_radix_sort64(uint64_t *beg, uint64_t *end) { radix_any((uint8_t*)beg, (uint8_t*)end, 8); }
void mpo_radix_sort128x(mpo128_t *beg, mpo128_t *end) { radix_any((uint8_t*)beg, (uint8_t*)end, 16); }

/* ---- hashing and sketching (sketch.c) ---------------------------------------------------------- */

uint32_t mpo_hash32_mask(uint32_t key, uint32_t mask)          /* sketch.c:7-16 */
{
	key = (key + ~(key << 15)) & mask;
	key ^= key >> 10;
	key = (key + (key << 3)) & mask;
	key ^= key >> 6;
	key = (key + ~(key << 11)) & mask;
	key ^= key >> 16;
	return key;
}

/* sketch.c:18-38.  out must hold len entries.  Entry = (hash>>mod_bit)<<32 | index of the k-mer's last residue. */
int64_t mpo_sketch_prot(const char *seq, int32_t len, int32_t kmer, int32_t mod_bit, uint64_t *out)
{
	const uint32_t mask_k = (1U << kmer * 4) - 1, mask_mod = (1U << mod_bit) - 1;
	uint32_t word = 0;
	int32_t i, run = 0;
	int64_t n = 0;
	for (i = 0; i < len; ++i) {
		uint32_t c = mpo_aa13((uint8_t)seq[i]);
		if (c >= 14) { word = 0, run = 0; continue; }          /* '*' and 'X' break the k-mer */
		word = (word << 4 | c) & mask_k;
		if (++run >= kmer) {
			uint32_t h = mpo_hash32_mask(word, mask_k);
			if ((h & mask_mod) == 0) out[n++] = (uint64_t)(h >> mod_bit) << 32 | (uint32_t)i;
		}
	}
	return n;
}

/* sketch.c:40-60: k-mers of one stop-free ORF [st,en) (multiple of 3 long, no N) */
static int64_t orf_kmers(const uint8_t *seq, int64_t st, int64_t en, int32_t kmer, int32_t mod_bit, int32_t bbit, int64_t boff, uint64_t *out, int64_t n, int64_t cap)
{
	const uint32_t mask_k = (1U << kmer * 4) - 1, mask_mod = (1U << mod_bit) - 1;
	const uint8_t *codon = mpo_codon_std();
	uint32_t word = 0;
	int32_t run = 0;
	int64_t i;
	for (i = st; i < en; i += 3) {
		uint32_t aa = codon[seq[i] << 4 | seq[i+1] << 2 | seq[i+2]];
		word = (word << 4 | mpo_aa13((uint8_t)aa)) & mask_k;
		if (++run >= kmer) {
			uint32_t h = mpo_hash32_mask(word, mask_k);
			if ((h & mask_mod) == 0 && n < cap)
				out[n++] = (uint64_t)(h >> mod_bit) << 32 | (uint64_t)(((i + 2) >> bbit) + boff);
		}
	}
	return n;
}

/* sketch.c:62-100: three-frame ORF scan of an nt4 string (one strand), sort, dedup */
int64_t mpo_sketch_nt4(const uint8_t *seq, int64_t len, int32_t min_aa_len, int32_t kmer, int32_t mod_bit, int32_t bbit, int64_t boff, uint64_t *out, int64_t cap)
{
	const uint8_t *codon_tab = mpo_codon_std();
	int64_t last_end[3] = { -1, -1, -1 }, n_codon[3] = { 0, 0, 0 }, i, n = 0, m;
	int32_t frame, q, run = 0;
	uint8_t codon = 0;
	/* frame of the codon ENDING at i is (i+1)%3 in the reference's bookkeeping (p starts at 1) */
	for (i = 0, frame = 1; i < len; ++i, ++frame) {
		if (frame == 3) frame = 0;
		if (seq[i] < 4) {
			codon = (codon << 2 | seq[i]) & 0x3f;
			if (++run >= 3) {
				if (codon_tab[codon] >= 20) {                  /* stop ends the ORF of this frame */
					if (n_codon[frame] >= min_aa_len)
						n = orf_kmers(seq, last_end[frame] + 1 - n_codon[frame] * 3, last_end[frame] + 1, kmer, mod_bit, bbit, boff, out, n, cap);
					n_codon[frame] = 0, last_end[frame] = -1;
				} else last_end[frame] = i, ++n_codon[frame];
			}
		} else {                                               /* N ends all three ORFs */
			for (q = 0; q < 3; ++q) {
				if (n_codon[q] >= min_aa_len)
					n = orf_kmers(seq, last_end[q] + 1 - n_codon[q] * 3, last_end[q] + 1, kmer, mod_bit, bbit, boff, out, n, cap);
				n_codon[q] = 0, last_end[q] = -1;
			}
			run = 0, codon = 0;
		}
	}
	for (q = 0; q < 3; ++q)
		if (n_codon[q] >= min_aa_len)
			n = orf_kmers(seq, last_end[q] + 1 - n_codon[q] * 3, last_end[q] + 1, kmer, mod_bit, bbit, boff, out, n, cap);
	if (n <= 1) return n;
	mpo_radix_sort64(out, out + n);
	for (i = 1, m = 0; i < n; ++i)
		if (out[m] != out[i]) out[++m] = out[i];
	return m + 1;
}
