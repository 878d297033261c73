// gs32_core.h -- ns_global_gs32b (nasw.h:131-132, nasw-sse.c:558-750): the spliced protein-to-genome DP in 32-bit scores, for
// score ranges the int16 kernels of dp_kernels.hip would saturate in.  Nothing on miniprot's mapping path calls it (align.c
// uses ns_global_gs16b only); it is part of the operator's interface (nasw.h:112-114) and exported for callers that link it.
//
// One source for the host and the device, like chain_core.h: the sweep is written for a TEAM (policy C: CoopSerial = one host
// thread, CoopWave = the 64 lanes of a wavefront) -- loops over the columns of a row are lane-strided, the gap chain is a prefix
// maximum over the team, decisions every lane must agree on are team reductions.  The host instance is a test model of the
// device code (tests/test_gs32_model.py pins it against the reference; the product runs the device instance only).
//
// WHAT is reproduced.  The reference's function is dead code with build-dependent behaviour:
//   * compiled as its Makefile does (x86-64, -O3, no -msse4.1) it returns garbage: the SSE2 fallback of _mm_insert_epi32
//     (nasw-sse.c:312-327) inserts `b << 16` as the upper half-word, i.e. zero, so every "-infinity" and "-fs" boundary value
//     becomes a small positive number (scores like 65669 for a 60-residue protein);
//   * compiled with -msse4.1 it is what nasw.h:112-114 describes -- the score of ns_global_gs16b on every input that stays inside
//     int16 (tests/test_gs32_model.py checks that too; about one CIGAR in ten differs between equally good paths, because the
//     first-pass gap chain restarts at the starts of 4 lanes instead of 8), the unsaturated score beyond;
//   * its lazy-F loop ends on ns_le_epi32(), which on every x86 build compares the vectors as EIGHT int16 (nasw-sse.c:277-284),
//     so for values outside int16 the loop can end early and leave H lower than the recurrence says.
// The parity target is the -msse4.1 build (oracle/_ref/libnasw_sse41.so, oracle/Makefile), bit for bit INCLUDING the
// eight-half-words comparison: score, CIGAR and every traceback word.
//
// The reference stripes the al columns over 4 SIMD lanes (vsize = 4, slen = ceil(al / 4)); column c is lane c / slen, vector
// c % slen, so "the vector to the left" of column c is column c - 1 for every c > 0 -- the sweep works on natural columns and
// only the gap chain (restarts at lane starts) and the lazy-F loop (walks vectors, all four lanes at once) see the striping.
#pragma once
#include <stdint.h>
#include "chain_core.h"

namespace mpa {

#define GS32_NEG (-0x40000000)

struct Gs32Params { int32_t go, ge, io, fs; int32_t wide_ge = 0; };   // wide_ge: the record's byte 2 is a stop flag, not the row's gap extension (dp_device.h DpConst)
// int32[W] each, W = 4 * ceil(al / 4): H rows i, i-1, i-2, i-3 (slot = row & 3), D rows (slot = row % 3, updated in place), A, B, C
// (in place), the best non-I state of the row, the first-pass I of the row; uint16[W]: the row's traceback words
struct Gs32Scratch { int32_t *H[4], *D[3], *A, *B, *C, *hn, *I1; uint16_t *zw; };
MPA_HD inline int64_t gs32_scratch_words(int32_t al) { return 13 * (int64_t)(4 * ((al + 3) / 4)); }   // int32 words (zw takes the last W / 2)
MPA_HD inline Gs32Scratch gs32_carve(int32_t *base, int32_t al)
{
	const int64_t W = 4 * ((al + 3) / 4);
	Gs32Scratch S;
	for (int k = 0; k < 4; ++k) S.H[k] = base + k * W;
	for (int k = 0; k < 3; ++k) S.D[k] = base + (4 + k) * W;
	S.A = base + 7 * W, S.B = base + 8 * W, S.C = base + 9 * W, S.hn = base + 10 * W, S.I1 = base + 11 * W;
	S.zw = (uint16_t*)(base + 12 * W);
	return S;
}

// wrapping int32 arithmetic (_mm_add_epi32 / _mm_sub_epi32)
MPA_HD inline int32_t gs32_add(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
MPA_HD inline int32_t gs32_sub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
MPA_HD inline int32_t gs32_max(int32_t a, int32_t b) { return a > b ? a : b; }
// ns_le_epi32 on x86 (nasw-sse.c:277-284): "no 16-bit half of a is greater than the same half of b"
MPA_HD inline bool gs32_le_halves(int32_t a, int32_t b)
{
	return (int16_t)(uint16_t)((uint32_t)a & 0xffffu) <= (int16_t)(uint16_t)((uint32_t)b & 0xffffu) && (a >> 16) <= (b >> 16);
}

// The sweep of one call.  rec: per-row records of the window (dp_device.h make_rec: 2*nas | donor[i+1] | gei | acceptor[i]);
// prof: int16 [22][pw] substitution scores of the protein's columns (columns >= al are not read: they score GS32_NEG);
// tb: uint16 [nl][ncol] or null.  Returns H(nl-1, al-1) (GS32_NEG when nl < 3), the same on every lane.
// Requires go >= 0 (the closed form of the gap chain), checked by the callers.
template<class C>
MPA_HD inline int32_t gs32_sweep(const Gs32Params p, const int32_t nl, const int32_t al, const uint32_t *rec, const int16_t *prof, const int32_t pw,
                                 const Gs32Scratch S, uint16_t *tb, const int32_t ncol)
{
	const int32_t slen = (al + 3) / 4, W = 4 * slen, NEG = GS32_NEG;
	const int32_t go = p.go, ge = p.ge, goe = gs32_add(p.go, p.ge), io = p.io, fs = p.fs;
	MPA_COOP_FOR(C, c, W) {                                              // NS_GEN_INIT1 (nasw-sse.c:250-255)
		for (int k = 0; k < 4; ++k) S.H[k][c] = NEG;
		for (int k = 0; k < 3; ++k) S.D[k][c] = NEG;
		S.A[c] = S.B[c] = S.C[c] = NEG;
	}
	C::sync();
	for (int32_t i = 2; i < nl; ++i) {
		const int32_t *H1 = S.H[(i - 1) & 3], *H2 = S.H[(i - 2) & 3], *H3 = S.H[(i - 3) & 3];
		int32_t *Hn = S.H[i & 3], *Dn = S.D[i % 3];
		const uint32_t r0 = rec[i], r1 = rec[i - 1], r2 = rec[i - 2];
		const int32_t aa = (int32_t)(r0 & 0xff) >> 1, gei_b = (int32_t)((r0 >> 16) & 0xff), gei = p.wide_ge ? (gei_b ? fs : ge) : gei_b;
		const int32_t dip1 = (int8_t)(r0 >> 8), di = (int8_t)(r1 >> 8), dim1 = (int8_t)(r2 >> 8);
		const int32_t ai = (int8_t)(r0 >> 24), aim1 = (int8_t)(r1 >> 24), aim2 = (int8_t)(r2 >> 24);
		// column -1 (NS_GEN_INIT1 for row 2, NS_GEN_INIT2 afterwards: lane 0 of the shifted-in vector is -infinity)
		const int32_t L3 = i == 2 ? 0 : NEG, L2 = i == 2 ? gs32_sub(0, fs) : NEG, L1 = L2;
		const int16_t *prow = prof + (int64_t)aa * pw;

		// ---- every state but I, each column on its own (nasw-sse.c:647-708 without the I lines)
		MPA_COOP_FOR(C, c, W) {
			const int32_t h1 = H1[c], h2 = H2[c], h3 = H3[c];
			const int32_t h1l = c ? H1[c - 1] : L1, h2l = c ? H2[c - 1] : L2, h3l = c ? H3[c - 1] : L3;
			int32_t h = gs32_add(h3l, c < al ? (int32_t)prow[c] : NEG), y = 0, t, u, v;
			uint32_t z = 0;
			u = gs32_sub(h3, go), v = Dn[c];                                // D
			if (v > u) z |= 1u << 5;
			t = gs32_sub(gs32_max(u, v), gei), Dn[c] = t;
			if (t > h) y = 2, h = t;
			u = gs32_sub(h1, io), v = S.A[c], t = gs32_sub(u, dim1);          // A
			if (v > t) z |= 1u << 6;
			t = gs32_max(t, v), S.A[c] = t, t = gs32_sub(t, ai);
			if (t > h) y = 3, h = t;
			u = gs32_sub(h1l, io), v = S.B[c], t = gs32_sub(u, di);           // B
			if (v > t) z |= 1u << 7;
			t = gs32_max(t, v), S.B[c] = t, t = gs32_sub(t, aim2);
			if (t > h) y = 4, h = t;
			v = S.C[c], t = gs32_sub(u, dip1);                               // C
			if (v > t) z |= 1u << 8;
			t = gs32_max(t, v), S.C[c] = t, t = gs32_sub(t, aim1);
			if (t > h) y = 5, h = t;
			t = gs32_sub(h1, fs); if (t > h) y = 6, h = t;                   // frameshifts
			t = gs32_sub(h2, fs); if (t > h) y = 7, h = t;
			t = gs32_sub(h1l, fs); if (t > h) y = 8, h = t;
			t = gs32_sub(h2l, fs); if (t > h) y = 9, h = t;
			S.hn[c] = h, S.zw[c] = (uint16_t)(z | (uint32_t)y);
		}
		C::sync();
		// ---- the gap chain of the first pass, lane by lane: I(j) = max(H(j-1) - go, I(j-1)) - ge from I = H = -infinity at the
		// lane's start; with go >= 0 that is max_{k<j}(nonI(k) + k ge) - go - j ge  or  -infinity - (j+1) ge, whichever is larger
		for (int s = 0; s < 4; ++s) {
			int64_t carry = INT64_MIN;
			for (int32_t base = 0; base < slen; base += C::width()) {
				const int32_t j = base + C::lane(), c = s * slen + j;
				const bool in = j < slen;
				int64_t tot;
				const int64_t below = C::scan_max_excl(in ? (int64_t)S.hn[c] + (int64_t)j * ge : INT64_MIN, &tot);
				const int64_t pm = below > carry ? below : carry;
				carry = tot > carry ? tot : carry;
				if (!in) continue;
				int64_t I = (int64_t)NEG - (int64_t)(j + 1) * ge;
				if (pm != INT64_MIN && pm - go - (int64_t)j * ge > I) I = pm - go - (int64_t)j * ge;
				const int32_t Ic = (int32_t)I;
				S.I1[c] = Ic;
				// I is the second state the reference tries (after M, before D): it wins ties against every state but M
				int32_t h = S.hn[c];
				uint32_t z = S.zw[c];
				if (Ic > h || (Ic == h && (z & 0xf) != 0)) z = (z & ~0xfu) | 1u, h = Ic;
				Hn[c] = h, S.zw[c] = (uint16_t)z;
			}
		}
		C::sync();
		// the "I(j-1) > H(j-1) - go" bit needs the neighbour's first-pass values: before the lazy-F loop touches H
		MPA_COOP_FOR(C, c, W) {
			const int32_t j = (int32_t)(c % slen);
			const int32_t pI = j ? S.I1[c - 1] : NEG, ph = j ? Hn[c - 1] : NEG;
			if (pI > gs32_sub(ph, go)) S.zw[c] |= 1u << 4;
		}
		int32_t Iv[4];
		for (int s = 0; s < 4; ++s) Iv[s] = gs32_max(gs32_sub(Hn[s * slen + slen - 1], goe), gs32_sub(S.I1[s * slen + slen - 1], ge));
		C::sync();
		// ---- the lazy-F loop (nasw-sse.c:711-728): up to four passes over the vectors, the lanes shifted by one each time, until
		// ns_le_epi32() says that no lane's I can raise anything any more
		for (int k = 0; k < 4; ++k) {
			Iv[3] = Iv[2], Iv[2] = Iv[1], Iv[1] = Iv[0], Iv[0] = NEG;
			int32_t first = 0;                                                // slen - (first vector at which the loop ends), 0 = it does not
			MPA_COOP_FOR(C, j, slen) {
				bool all = true;
				for (int s = 0; s < 4; ++s) {
					const int32_t Ib = gs32_sub(Iv[s], (int32_t)j * ge), hnew = gs32_max(Hn[s * slen + j], Ib);
					all = all && gs32_le_halves(gs32_sub(Ib, ge), gs32_sub(hnew, goe));
				}
				if (all && slen - (int32_t)j > first) first = slen - (int32_t)j;
			}
			first = C::reduce_max(first);
			const int32_t last = first ? slen - first : slen - 1;             // vectors 0 .. last are updated in this pass
			C::sync();
			MPA_COOP_FOR(C, j, last + 1) {
				for (int s = 0; s < 4; ++s) {
					const int32_t c = s * slen + (int32_t)j, Ib = gs32_sub(Iv[s], (int32_t)j * ge);
					if (Ib > Hn[c]) Hn[c] = Ib, S.zw[c] |= 1u << 9;
				}
			}
			C::sync();
			if (first) break;
			for (int s = 0; s < 4; ++s) Iv[s] = gs32_sub(Iv[s], slen * ge);
		}
		if (tb) { MPA_COOP_FOR(C, c, W) tb[(int64_t)i * ncol + c] = S.zw[c]; }
		C::sync();
	}
	return nl >= 3 ? S.H[(nl - 1) & 3][al - 1] : NEG;
}

} // namespace mpa
