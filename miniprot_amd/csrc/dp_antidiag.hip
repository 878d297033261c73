// dp_antidiag.hip -- the ANTI-DIAGONAL mapping of the score-only extension sweep (ns_global_gs16b, nasw-sse.c:349-443), the
// mapping BASELINE.json's north_star names (option A), for the 32-column class.  Included by dp_exec.hip.
//
// What it is for: a measured answer.  The production sweep (ext_narrow, dp_kernels.hip) is a ROW sweep: all columns of a row in
// lockstep, the horizontal gap chain as a max-plus prefix scan over the lanes.  Here lane j works on row t - j at step t: the
// cell (i, j) needs H(i, j-1) and I(i, j-1) -- what lane j-1 produced ONE step ago -- and H(i-1..i-3, j-1), which the lane keeps
// from the earlier steps, so the scan becomes two DPP shifts and the recurrence is the reference's own formula
//     I(i,j) = max(H(i,j-1) - go, I(i,j-1)) - ge                                  (nasw-sse.c:393-398)
// instead of the prefix maximum of h + j*ge.  The price: every lane is on a different row (per-lane record and ring addresses),
// the row maximum of the best-row / x-drop logic (nasw-sse.c:423-433) has to travel down the diagonal with the chain and is
// complete only in the call's last column, the H row of the best row has to be recovered from a per-lane history, and a call of
// c columns takes nl + c steps.  Same inputs (row records, profiles, ExtWave descriptors of class 1) and same outputs (ExtOut) as
// ext_narrow<32>, bit for bit; selected with mpa_dbg_antidiag() and timed by tools/time_wide.py (profiles/r04_antidiag_ns_per_row.txt).
// Two lane groups of 32 columns x two int16 halves = four calls per wave, as in ext_narrow<32>.
#pragma once

namespace mpa {

#define AD_G 32
#define AD_RING 128                                       /* rows of records per lane group in LDS (power of two) */
#define AD_HIST 32                                        /* steps of H history per lane: the best row's H(i, j) was computed ncol-1-j steps before its row maximum is complete */
#define EXT_ANTIDIAG_LDS ((((size_t)4 * AD_G * PROF_COL_STRIDE + 7) & ~(size_t)7) + (size_t)2 * AD_RING * 8 + (size_t)AD_HIST * 64 * 4 + 2 * MPA_PEN_MAX * 4)

__device__ __forceinline__ void ext_antidiag32(const ExtArgs &a, const ExtWave *wvp, const WavePos wp)
{
	constexpr int G = AD_G, NG = 2, NSLOT = 4;
	char *lds_prof = wp.lds;                                                              // [NSLOT][G columns][23] int16
	uint2 *lds_ring = (uint2*)(lds_prof + ((NSLOT * G * PROF_COL_STRIDE + 7) & ~7));       // [NG][AD_RING rows] {record of half 0, record of half 1}
	uint32_t *lds_hist = (uint32_t*)(lds_ring + NG * AD_RING);                            // [AD_HIST steps][64 lanes] packed H
	int32_t *lds_pen = (int32_t*)(lds_hist + AD_HIST * 64);                               // x[MPA_PEN_MAX], val[MPA_PEN_MAX]

	const int lane = wp.lane, grp = lane / G, col = lane % G;
	const DpConst c = a.c;
	const uint32_t *recbase = a.rec + wvp->rec_base;

	int32_t tid[2], nl[2], al[2], ncol[2];
	uint32_t roff[2];
	bool done[2];
#pragma unroll
	for (int h = 0; h < 2; ++h) {
		tid[h] = wvp->task[h * NG + grp];
		if (tid[h] >= 0) {
			const DTask *t = &a.tasks[tid[h]];
			nl[h] = t->nl, al[h] = t->al, ncol[h] = t->ncol;
			roff[h] = (uint32_t)(t->rec_off - wvp->rec_base);
			done[h] = false;
		} else nl[h] = 0, al[h] = 1, ncol[h] = G, roff[h] = 0, done[h] = true;
	}
	// ---- profiles: global [22][pw] int16 -> LDS [col][aa] (as ext_narrow)
	for (int s = 0; s < NSLOT; ++s) {
		const int t = wvp->task[s];
		if (t < 0) {
			for (int k = lane; k < G * PROF_COL_STRIDE / 2; k += 64) ((uint16_t*)(lds_prof + s * G * PROF_COL_STRIDE))[k] = 0;
			continue;
		}
		const DTask *tt = &a.tasks[t];
		const int16_t *src = a.prof + tt->prof_off;
		for (int k = lane; k < 22 * G; k += 64) {
			const int aa = k / G, cc = k - aa * G;
			*(int16_t*)(lds_prof + (s * G + cc) * PROF_COL_STRIDE + aa * PROF_AA_STRIDE) = src[aa * tt->pw + cc];
		}
	}
	for (int k = lane; k < MPA_PEN_MAX; k += 64) lds_pen[k] = a.pen.x[k], lds_pen[MPA_PEN_MAX + k] = a.pen.val[k];
	for (int k = lane; k < AD_HIST * 64; k += 64) lds_hist[k] = NEGP;
	// ---- record ring: AD_RING rows per lane group; lane (grp, col) moves row 32 k + col.  Rows [0, 64) go in now, rows [64, 96)
	// wait in `pf`; at every step t = 0 (mod 32) the rows [t + 32, t + 64) are published and the next 32 requested.  At step t the
	// lanes read rows t - 31 .. t + 2.
	uint2 *myring = lds_ring + grp * AD_RING;
	myring[col] = make_uint2(recbase[roff[0] + col], recbase[roff[1] + col]);
	myring[32 + col] = make_uint2(recbase[roff[0] + 32 + col], recbase[roff[1] + 32 + col]);
	uint2 pf = make_uint2(recbase[roff[0] + 64 + col], recbase[roff[1] + 64 + col]);
	wave_sync();

	// ---- per-lane constants
	const uint32_t ebv = pack16(col == al[0] - 1 ? c.end_bonus : 0, col == al[1] - 1 ? c.end_bonus : 0);
	// the lane in which a row's maximum is complete: the call's last (padded) column -- the reference's row maximum runs over
	// all 8 * ceil(al / 8) columns (nasw-sse.c:404,415)
	const uint32_t decP = (col == ncol[0] - 1 ? 0x8000u : 0u) | (col == ncol[1] - 1 ? 0x80000000u : 0u);
	const uint32_t goP = splat16(c.go), geP = splat16(c.ge), fsP = splat16(c.fs);
	const uint32_t ioP = pack16(tid[0] >= 0 ? a.tasks[tid[0]].io : 0, tid[1] >= 0 ? a.tasks[tid[1]].io : 0);
	const uint32_t pbase0 = (uint32_t)(((0 * NG + grp) * G + col) * PROF_COL_STRIDE);
	const uint32_t pbase1 = (uint32_t)(((1 * NG + grp) * G + col) * PROF_COL_STRIDE);

	// ---- DP state of the lane's column: rings of three rows (compile-time indexed), A/B/C persist
	uint32_t Hr[3], Hs[3], Dr[3], dn[3], ac[3], A = NEGP, B = NEGP, C = NEGP, Hbest = NEGP;
#pragma unroll
	for (int k = 0; k < 3; ++k) Hr[k] = Hs[k] = Dr[k] = NEGP, dn[k] = ac[k] = 0;
	// virtual column -1 as row 2 of column 0 sees it (nasw-sse.c:256-258): H(-1,-1) = 0, H(0,-1) = H(1,-1) = -fs
	if (col == 0) Hs[2] = 0u, Hs[1] = splat16(-c.fs), Hs[0] = splat16(-c.fs);
	uint32_t Hlast = NEGP, Ilast = NEGP, Mlast = NEGP;     // what this lane produced one step ago: H, I and the running row maximum
	uint32_t keepH = NEGP, keepI = NEGP, keepM = NEGP;     // fill registers of the three shifts (shift1_keep)

	// ---- bookkeeping of the reference's score-only loop (nasw-sse.c:423-433), per half; a call's values are uniform in its lane group
	int32_t Gs[2] = { NEG16, NEG16 }, maxi[2] = { -1, -1 }, maxsc[2] = { INT32_MIN, INT32_MIN };
	int32_t pk[2] = { 0, 0 }, pen[2] = { 0, 0 };
	uint32_t Gv = NEGP, Tv = NEGP, penP = 0;
	int32_t next_event = 2, limit = 0;
#pragma unroll
	for (int h = 0; h < 2; ++h) if (tid[h] >= 0) limit = limit > nl[h] + ncol[h] - 1 ? limit : nl[h] + ncol[h] - 1;
#pragma unroll
	for (int off = 1; off < 64; off <<= 1) { const int32_t o = __shfl_xor(limit, off); limit = limit > o ? limit : o; }
	limit = __builtin_amdgcn_readfirstlane(limit);         // steps: the last column of the longest call finishes its last row at nl + ncol - 2

	auto repack = [&]() {
		Gv = pack16(done[0] ? 32767 : Gs[0], done[1] ? 32767 : Gs[1]);
		Tv = pack16(done[0] ? NEG16 : sat16(Gs[0] - c.xdrop), done[1] ? NEG16 : sat16(Gs[1] - c.xdrop));
		penP = pack16(pen[0], pen[1]);
		if (__ballot(!done[0] || !done[1]) == 0) limit = 0;               // every call of the wave has finished
	};
	// task ends and steps of the length penalty, in terms of the row the DECIDING lane of a call is on: i_h = t - (ncol_h - 1)
	auto events = [&](int32_t t) {
		int32_t nx = INT32_MAX;
#pragma unroll
		for (int h = 0; h < 2; ++h) {
			const int32_t ih = t - (ncol[h] - 1);
			if (!done[h] && ih >= nl[h]) done[h] = true;
			if (!done[h]) {
				const int64_t x = (int64_t)ih - 3 * (int64_t)al[h];
				while (x >= (int64_t)lds_pen[pk[h] + 1]) ++pk[h];
				pen[h] = lds_pen[MPA_PEN_MAX + pk[h]];
				int64_t e = (int64_t)lds_pen[pk[h] + 1] + 3 * (int64_t)al[h];
				if (e > nl[h]) e = nl[h];
				e += ncol[h] - 1;                                                // ... as a step number
				nx = nx < (int32_t)e ? nx : (int32_t)e;
			}
		}
#pragma unroll
		for (int off = 1; off < 64; off <<= 1) { const int32_t o = __shfl_xor(nx, off); nx = nx < o ? nx : o; }
		next_event = __builtin_amdgcn_readfirstlane(nx);
		repack();
	};

	int32_t t = 2;                                          // step; this lane is on row i = t - col
	// per-lane ring address of the record of row i + 2 (bytes, inside the group's ring), advanced by one row per step
	uint32_t rpos = (uint32_t)((t - col + 2) & (AD_RING - 1)) * 8u;
	const char *ringb = (const char*)myring;
	// software pipeline over LDS latency as in ext_narrow: records of rows i and i + 1 in registers, profile scores of row i
	uint2 rcur = *(const uint2*)(ringb + (((uint32_t)(t - col) & (AD_RING - 1)) * 8u));
	uint2 rnext = *(const uint2*)(ringb + (((uint32_t)(t - col + 1) & (AD_RING - 1)) * 8u));
	uint32_t S = prof2(lds_prof + pbase0 + (rcur.x & 0xff), lds_prof + pbase1 + (rcur.y & 0xff));
	// records of rows i - 1 and i - 2 feed the donor / acceptor windows of row i; a lane that starts below row 0 picks them up on its way
	uint32_t hpos = (uint32_t)lane * 4u + (uint32_t)(t & (AD_HIST - 1)) * 256u;   // this lane's slot in the history line of step t (line t mod AD_HIST)
	{	// the donor / acceptor windows as if steps 0 and 1 had run: the records of rows 0 - col and 1 - col (columns 0 and 1 start
		// at row 2 and 1 and would otherwise never see rows 0 and 1; further right the lanes pick them up on their way)
		const uint2 q0 = *(const uint2*)(ringb + (((uint32_t)(0 - col) & (AD_RING - 1)) * 8u)), q1 = *(const uint2*)(ringb + (((uint32_t)(1 - col) & (AD_RING - 1)) * 8u));
		dn[1] = __builtin_amdgcn_perm(q0.y, q0.x, 0x0a050801u), dn[0] = __builtin_amdgcn_perm(q1.y, q1.x, 0x0a050801u);
		ac[1] = __builtin_amdgcn_perm(q0.y, q0.x, 0x0b070903u), ac[0] = __builtin_amdgcn_perm(q1.y, q1.x, 0x0b070903u);
	}

	auto step = [&](auto kc, auto pro) {
		constexpr int K = decltype(kc)::value;
		constexpr bool PROLOGUE = decltype(pro)::value;         // some lanes are still above row 2
		constexpr int R1 = (3 - K) % 3, R2 = (4 - K) % 3, R3 = (5 - K) % 3;   // ring slots of rows i-1, i-2, i-3
		if ((t & 31) == 0) {                                   // rows [t + 32, t + 64) published, [t + 64, t + 96) requested
			myring[((t + 32) & (AD_RING - 1)) + col] = pf;
			pf = make_uint2(recbase[roff[0] + (uint32_t)t + 64 + col], recbase[roff[1] + (uint32_t)t + 64 + col]);
		}
		const uint32_t Snext = prof2(lds_prof + pbase0 + (rnext.x & 0xff), lds_prof + pbase1 + (rnext.y & 0xff));
		const uint2 rnn = *(const uint2*)(ringb + rpos);
		rpos = (rpos + 8u) & (uint32_t)(AD_RING * 8 - 1);
		// what the left neighbour produced one step ago, on THIS lane's row: H(i, j-1), I(i, j-1), running maximum of row i
		const uint32_t Lh = shift1_keep<G>(keepH, Hlast, NEGP, lane);
		const uint32_t Li = shift1_keep<G>(keepI, Ilast, NEGP, lane);
		const uint32_t Lm = shift1_keep<G>(keepM, Mlast, NEGP, lane);
		dn[R3] = __builtin_amdgcn_perm(rcur.y, rcur.x, 0x0a050801u);
		ac[R3] = __builtin_amdgcn_perm(rcur.y, rcur.x, 0x0b070903u);
		const uint32_t gei = __builtin_amdgcn_perm(rcur.y, rcur.x, 0x0c060c02u);
		uint32_t h, x, u;
		h = p_adds(Hs[R3], S);                                             // H(i-3,j-1) + s(i,j)
		u = p_subs(Hr[R3], goP); x = p_max(u, Dr[R3]);                     // D
		x = p_subs(x, gei); Dr[R3] = x; h = p_max(h, x);
		u = p_subs(Hr[R1], ioP); x = p_subs(u, dn[R2]);                    // A: phase-0 intron, donor[i-1], acceptor[i]
		x = p_max(x, A); A = x; h = p_max(h, p_subs(x, ac[R3]));
		u = p_subs(Hs[R1], ioP); x = p_subs(u, dn[R1]);                    // B: phase-1 intron, donor[i], acceptor[i-2]
		x = p_max(x, B); B = x; h = p_max(h, p_subs(x, ac[R2]));
		x = p_subs(u, dn[R3]);                                             // C: phase-2 intron, donor[i+1], acceptor[i-1]
		x = p_max(x, C); C = x; h = p_max(h, p_subs(x, ac[R1]));
		x = p_max(p_max(Hr[R1], Hr[R2]), p_max(Hs[R1], Hs[R2]));          // four frameshift moves
		h = p_max(h, p_subs(x, fsP));
		const uint32_t ii = p_subs(p_max(p_subs(Lh, goP), Li), geP);       // I(i,j) = max(H(i,j-1) - go, I(i,j-1)) - ge
		h = p_max(h, ii);
		if (PROLOGUE && t - col < 2) h = NEGP;                             // rows 0 and 1 are not part of the matrix: -inf
		Hr[R3] = h, Hs[R3] = Lh;
		Hlast = h, Ilast = ii;
		// running maximum of row i over the columns 0..j (the last real column carries the end bonus), complete in lane ncol - 1
		const uint32_t mm = p_max(Lm, p_adds(h, ebv));
		Mlast = mm;
		*(uint32_t*)((char*)lds_hist + hpos) = h;                          // H(i, j) for the day row i turns out to be the best row
		hpos = (hpos + 256u) & (uint32_t)(AD_HIST * 256 - 1);
		// votes of the deciding lanes: sign bit of x set <=> the row beats the call's best; of y <=> it is below the x-drop threshold
		const uint32_t cand = p_subs(mm, penP);
		const uint32_t vx = p_subs(Gv, cand), vy = p_subs(cand, Tv);
		if (__ballot(((vx | vy) & decP) != 0) != 0) {
			// slow path: exact restatement of nasw-sse.c:423-433 for the row the call's deciding lane has just completed
#pragma unroll
			for (int hh = 0; hh < 2; ++hh) {
				const int32_t ih = t - (ncol[hh] - 1);
				const int32_t m = __shfl(half16(cand, hh), grp * G + ncol[hh] - 1);
				if (!done[hh] && ih >= 2 && ih < nl[hh]) {
					if (m > Gs[hh]) {                                          // new best row: its H(ih, j) was computed ncol-1-j steps ago
						Gs[hh] = m, maxi[hh] = ih, maxsc[hh] = m + pen[hh];
						const int32_t back = ncol[hh] - 1 - col;
						const uint32_t hv = back >= 0 ? lds_hist[(((uint32_t)(t - back)) & (AD_HIST - 1)) * 64 + lane] : NEGP;
						Hbest = hh ? ((Hbest & 0xffffu) | (hv & 0xffff0000u)) : ((Hbest & 0xffff0000u) | (hv & 0xffffu));
					}
					if (Gs[hh] - m > c.xdrop) done[hh] = true;                 // x-drop
				}
			}
			repack();
		}
		if (PROLOGUE && t == 2 && col == 0) Hs[R1] = NEGP, Hs[R2] = NEGP;   // column -1 is -inf from row 3 on
		S = Snext;
		rcur = rnext, rnext = rnn;
		++t;
	};

	using T0 = std::integral_constant<int, 0>; using T1 = std::integral_constant<int, 1>; using T2 = std::integral_constant<int, 2>;
	// the first 33 steps: lanes above row 2 hold -inf
	while (t < limit && t < 2 + 33) {
		if (t >= next_event) { events(t); if (t >= limit) break; }
		step(T0(), std::true_type());
		if (t >= limit) break;
		if (t >= next_event) { events(t); if (t >= limit) break; }
		step(T1(), std::true_type());
		if (t >= limit) break;
		if (t >= next_event) { events(t); if (t >= limit) break; }
		step(T2(), std::true_type());
	}
	while (t < limit) {                                    // (t = 35 here: 35 - 2 is a multiple of 3, the ring phase carries on)
		if (t >= next_event) { events(t); if (t >= limit) break; }
		step(T0(), std::false_type());
		if (t >= limit) break;
		if (t >= next_event) { events(t); if (t >= limit) break; }
		step(T1(), std::false_type());
		if (t >= limit) break;
		if (t >= next_event) { events(t); if (t >= limit) break; }
		step(T2(), std::false_type());
	}

	// ---- results (nasw-sse.c:435-443): first column of the best row that attains the best score
#pragma unroll
	for (int h = 0; h < 2; ++h) {
		if (tid[h] < 0) continue;
		int32_t first = INT32_MAX;
		if (maxi[h] >= 0) {
			int32_t v = half16(Hbest, h);
			if (col == al[h] - 1) v += c.end_bonus;
			const uint64_t m = __ballot(col < al[h] && v == maxsc[h]) & group_mask<G>(lane);
			if (m != 0) first = (__ffsll((unsigned long long)m) - 1) % G;
		}
		if (col == 0) {
			ExtOut o;
			o.nt_len = maxi[h] + 1;
			o.aa_len = maxi[h] < 0 ? 0 : (first == INT32_MAX ? al[h] + 1 : first + 1);
			o.score = maxsc[h];
			o.flags = (maxi[h] >= 0 && first == INT32_MAX) ? 1 : 0;
			a.out[tid[h]] = o;
		}
	}
}

// one wave per block: the class-1 wave descriptors [first, first + gridDim.x)
__global__ __launch_bounds__(64) void k_ext_antidiag(ExtArgs a, int first)
{
	extern __shared__ uint32_t lds_raw_ad[];
	ext_antidiag32(a, a.waves + first + blockIdx.x, whole_block((char*)lds_raw_ad));
}

} // namespace mpa
