// dp_kernels.hip -- hand-written CDNA4 (gfx950) kernels for miniprot's spliced protein-to-genome DP.
//
// What they replace in the reference (lh3/miniprot v0.18-r281):
//   k_prep_rows / k_prep_prof  ns_prep_seq, ns_prep_seq_left, ns_prep_nas, ns_gen_prof   nasw-sse.c:91-224
//   k_ext_narrow / k_ext_wide  ns_global_gs16b score-only / extension loop               nasw-sse.c:349-443
//   k_glob_narrow / _wide      ns_global_gs16b traceback loop                             nasw-sse.c:444-542
//   k_backtrack                ns_backtrack + ns_fix_tiny_UV                              nasw-sse.c:30-89
//
// Mapping to the hardware (DESIGN.md has the full argument):
//   * The DP matrix is tall and thin (rows = genomic window up to ~1e5, columns = the unaligned protein
//     end, ~25 on average), rows are strictly sequential, so the parallelism of ONE call is its columns.
//     A wavefront therefore sweeps rows with one column per lane ("row sweep"); the horizontal gap chain
//     I(i,j) <- H(i,j-1) becomes a max-plus prefix scan done with DPP row_shr / row_bcast steps.
//   * To fill 64 lanes x 2 int16 halves, k_ext packs up to 8 independent calls into one wave: lanes are
//     split into 64/G groups of G columns and each 32-bit register carries two calls in its halves, all
//     arithmetic being v_pk_{add,sub}_i16 clamp / v_pk_max_i16 -- the exact counterparts of the
//     reference's _mm_adds/_mm_subs/_mm_max_epi16.
//   * Per-row inputs (codon amino acid, donor/acceptor penalties) come from a 4-byte record per window row
//     written once by k_prep_rows straight from the 4-bit packed genome in HBM (reverse-complement and
//     reversal folded into the address), unpacked with v_perm_b32.  The 22 x ncol query profile sits in LDS.
//   * The per-row "best row / x-drop" bookkeeping of the reference needs a row maximum; instead of a wave
//     reduction per row the kernel compares every lane against two per-call thresholds and uses the
//     64-bit VOPC result as a wave-wide vote; the exact update runs in a rare slow path.
//   * k_glob has to reproduce artefacts of the reference's 8-lane striping in the traceback word
//     (SURVEY.md A.3): it computes in int32, one column per lane, with a SEGMENTED scan for the first-pass
//     I chain and a second scan for the cross-segment carry.  Matrices wider than 64 columns are processed
//     in column blocks with a per-row boundary record.
//
// No MFMA: this is integer max-plus DP, not a contraction.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "dp_device.h"

namespace mpa {

struct DevTables { uint8_t aa20[256]; uint8_t codon[64]; int8_t mat[484]; };

// Where a wavefront sits: every DP body below is written for "a group of NW waves working on one unit" and is told its LDS
// region, its lane, its wave index inside the group (scalar) and its thread index inside the group.  The stand-alone kernels
// are one group per workgroup; k_dp_round packs several groups (e.g. four independent one-wave units) into a 256-thread
// workgroup.
struct WavePos { char *lds; int lane, w, tg; };
__device__ __forceinline__ WavePos whole_block(char *lds)
{
	return WavePos{ lds, (int)(threadIdx.x & 63), __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), (int)threadIdx.x };
}
// ordering of LDS writes and reads among the lanes of ONE wave: the LDS executes a wave's instructions in order, so only the
// compiler has to be kept from moving them (a one-wave unit needs no s_barrier, and inside k_dp_round must not use one: the
// other waves of the workgroup are on different units)
__device__ __forceinline__ void wave_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }

// ------------------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------------------
typedef short v2s __attribute__((ext_vector_type(2)));
#define NEGP 0x80008000u
#define NEG16 (-32768)

__device__ __forceinline__ v2s as_v(uint32_t x) { return __builtin_bit_cast(v2s, x); }
__device__ __forceinline__ uint32_t as_u(v2s x) { return __builtin_bit_cast(uint32_t, x); }
__device__ __forceinline__ uint32_t p_adds(uint32_t a, uint32_t b) { return as_u(__builtin_elementwise_add_sat(as_v(a), as_v(b))); }
__device__ __forceinline__ uint32_t p_subs(uint32_t a, uint32_t b) { return as_u(__builtin_elementwise_sub_sat(as_v(a), as_v(b))); }
__device__ __forceinline__ uint32_t p_max(uint32_t a, uint32_t b) { return as_u(__builtin_elementwise_max(as_v(a), as_v(b))); }
__device__ __forceinline__ uint32_t splat16(int32_t x) { return ((uint32_t)x & 0xffffu) * 0x10001u; }
__device__ __forceinline__ uint32_t pack16(int32_t lo, int32_t hi) { return ((uint32_t)lo & 0xffffu) | ((uint32_t)hi << 16); }
__device__ __forceinline__ int32_t lo16(uint32_t x) { return (int32_t)(int16_t)(x & 0xffffu); }
__device__ __forceinline__ int32_t hi16(uint32_t x) { return (int32_t)x >> 16; }
__device__ __forceinline__ int32_t half16(uint32_t x, int h) { return h ? hi16(x) : lo16(x); }

template<int CTRL, int RMASK> __device__ __forceinline__ uint32_t dpp_mov(uint32_t old, uint32_t src)
{
	return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)src, CTRL, RMASK, 0xf, false);
}
#define DPP_ROW_SHR(n) (0x110 + (n))
#define DPP_WAVE_SHR1  0x138
#define DPP_BCAST15    0x142
#define DPP_BCAST31    0x143

// value of the left neighbour lane inside a group of G lanes; the first lane of each group gets `fill`
template<int G> __device__ __forceinline__ uint32_t shift1(uint32_t x, uint32_t fill, int lane)
{
	if (G == 16) return dpp_mov<DPP_ROW_SHR(1), 0xf>(fill, x);
	uint32_t r = dpp_mov<DPP_WAVE_SHR1, 0xf>(fill, x);
	if (G == 32) r = lane == 32 ? fill : r;
	return r;
}

// inclusive prefix max inside groups of G lanes, packed int16 (both halves at once)
template<int G> __device__ __forceinline__ uint32_t scan_max_pk(uint32_t x)
{
	x = p_max(x, dpp_mov<DPP_ROW_SHR(1), 0xf>(NEGP, x));
	x = p_max(x, dpp_mov<DPP_ROW_SHR(2), 0xf>(NEGP, x));
	x = p_max(x, dpp_mov<DPP_ROW_SHR(4), 0xf>(NEGP, x));
	x = p_max(x, dpp_mov<DPP_ROW_SHR(8), 0xf>(NEGP, x));
	if (G >= 32) x = p_max(x, dpp_mov<DPP_BCAST15, 0xa>(NEGP, x));
	if (G == 64) x = p_max(x, dpp_mov<DPP_BCAST31, 0xc>(NEGP, x));
	return x;
}

// The same with the fill registers of the first three steps kept by the caller (all NEGP before the first call, never written
// elsewhere): a DPP move leaves the lanes without a source untouched, so each register still holds NEGP exactly where its
// step needs it and that step's "v_mov fill" goes away (VOP3P has no DPP operand on gfx9: move + v_pk_max is the shortest
// step there is).  Three, not all six: registers are the scarcer resource in the kernel that inlines this (k_dp_round).
template<int G> __device__ __forceinline__ uint32_t scan_max_pk_keep(uint32_t x, uint32_t (&k)[3])
{
	k[0] = dpp_mov<DPP_ROW_SHR(1), 0xf>(k[0], x), x = p_max(x, k[0]);
	k[1] = dpp_mov<DPP_ROW_SHR(2), 0xf>(k[1], x), x = p_max(x, k[1]);
	k[2] = dpp_mov<DPP_ROW_SHR(4), 0xf>(k[2], x), x = p_max(x, k[2]);
	x = p_max(x, dpp_mov<DPP_ROW_SHR(8), 0xf>(NEGP, x));
	if (G >= 32) x = p_max(x, dpp_mov<DPP_BCAST15, 0xa>(NEGP, x));
	if (G == 64) x = p_max(x, dpp_mov<DPP_BCAST31, 0xc>(NEGP, x));
	return x;
}

// wave-wide maximum of unsigned words, valid in lane 63.  VOP2 takes the DPP operand directly; a lane without a source (or in
// a masked row) is not written, i.e. keeps its value -- no fill needed (cf. scan_max_i32).
__device__ __forceinline__ uint32_t reduce_max_u32(uint32_t x)
{
	asm("s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
	    "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
	    "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
	    "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
	    "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
	    "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf" : "+v"(x));
	return x;
}

// six such reductions at once, step by step across the six: every DPP read is five instructions behind the write it depends
// on, so the chains need no wait states between their steps
__device__ __forceinline__ void reduce_max_u32_x6(uint32_t &a, uint32_t &b, uint32_t &c, uint32_t &d, uint32_t &e, uint32_t &f)
{
#define MPA_STEP6(ctrl) \
	"v_max_u32_dpp %0, %0, %0 " ctrl "\n\tv_max_u32_dpp %1, %1, %1 " ctrl "\n\tv_max_u32_dpp %2, %2, %2 " ctrl "\n\t" \
	"v_max_u32_dpp %3, %3, %3 " ctrl "\n\tv_max_u32_dpp %4, %4, %4 " ctrl "\n\tv_max_u32_dpp %5, %5, %5 " ctrl "\n\t"
	asm("s_nop 1\n\t"
	    MPA_STEP6("row_shr:1 row_mask:0xf bank_mask:0xf")
	    MPA_STEP6("row_shr:2 row_mask:0xf bank_mask:0xf")
	    MPA_STEP6("row_shr:4 row_mask:0xf bank_mask:0xf")
	    MPA_STEP6("row_shr:8 row_mask:0xf bank_mask:0xf")
	    MPA_STEP6("row_bcast:15 row_mask:0xa bank_mask:0xf")
	    MPA_STEP6("row_bcast:31 row_mask:0xc bank_mask:0xf")
	    : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));
#undef MPA_STEP6
}

// barrier that orders LDS traffic only: outstanding global stores/atomics (vmcnt) keep draining across it
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#define NEG32 (-(1 << 29))
__device__ __forceinline__ int32_t imax(int32_t a, int32_t b) { return a > b ? a : b; }
// The compiler emits mov + hazard nop + v_mov_b32_dpp + v_max per step; VOP2 takes the DPP operand directly, and a lane
// whose source is out of range (or whose row is masked) is simply not written, i.e. keeps x -- no fill value needed.
template<int G> __device__ __forceinline__ int32_t scan_max_i32(int32_t x)
{
	asm("s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
	    "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
	    "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
	    "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf" : "+v"(x));
	if (G >= 32) asm("s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(x));
	if (G == 64) asm("s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf" : "+v"(x));
	return x;
}
template<int G> __device__ __forceinline__ int32_t shift1_i32(int32_t x, int32_t fill, int lane)
{
	return (int32_t)shift1<G>((uint32_t)x, (uint32_t)fill, lane);
}

// shift1 with a constant fill whose register is kept across calls: the DPP move leaves the lanes without a source untouched,
// so a register that held `fill` everywhere once still holds it in exactly those lanes after any number of shifts -- no
// "v_mov fill" per use (keep must start as fill and must only ever be written here)
template<int G> __device__ __forceinline__ uint32_t shift1_keep(uint32_t &keep, uint32_t x, uint32_t fill, int lane)
{
	if (G == 16) { keep = dpp_mov<DPP_ROW_SHR(1), 0xf>(keep, x); return keep; }
	keep = dpp_mov<DPP_WAVE_SHR1, 0xf>(keep, x);
	if (G == 32) return lane == 32 ? fill : keep;
	return keep;
}

template<int G> __device__ __forceinline__ int32_t group_max(int32_t v)
{
#pragma unroll
	for (int off = 1; off < G; off <<= 1) v = imax(v, __shfl_xor(v, off));
	return v;
}
template<int G> __device__ __forceinline__ uint64_t group_mask(int lane)
{
	if (G == 64) return ~0ULL;
	return ((1ULL << G) - 1) << (lane / G * G);
}
// does the 64-bit vote contain a group of G lanes with no bit set?
template<int G> __device__ __forceinline__ bool has_empty_group(uint64_t m)
{
	if (G == 64) return m == 0;
	if (G == 32) return (uint32_t)m == 0 || (uint32_t)(m >> 32) == 0;
	return ((m - 0x0001000100010001ULL) & ~m & 0x8000800080008000ULL) != 0;
}

__device__ __forceinline__ int32_t sat16(int32_t x) { return x < -32768 ? -32768 : x > 32767 ? 32767 : x; }
__device__ __forceinline__ int32_t s_sub(int32_t a, int32_t b) { return sat16(a - b); }
__device__ __forceinline__ int32_t s_add(int32_t a, int32_t b) { return sat16(a + b); }

// ------------------------------------------------------------------------------------------------
// K3: per-row records and query profiles
// ------------------------------------------------------------------------------------------------
struct PrepChunk { int32_t task; int32_t row0; };
#define PROF_AA_STRIDE_REC 2   /* byte0 of a record = amino acid * 2 = byte offset into an int16 profile column (k_ext) */

// base of the strand-oriented contig at strand-local position x (ntseq.c:89-106 folded into addressing; the scans of seed_exec.hip)
__device__ __forceinline__ uint32_t strand_base(const uint8_t *seq, int64_t off, int64_t len, int rev, int64_t x)
{
	int64_t p = rev ? off + len - 1 - x : off + x;
	uint32_t b = (seq[p >> 1] >> ((p & 1) * 4)) & 0xf;
	return rev && b < 4 ? 3 - b : b;
}

// Sixteen consecutive bases of the packed genome as the nibbles of one word: nibble j = the base at genome position
// p_first + dir * j (dir = +1 / -1), complemented (3 - b for the codes below 4) when `comp`.  Three aligned words cover them;
// positions outside the genome buffer (which is padded by 16 bytes) read as garbage -- the caller masks what it does not own.
__device__ __forceinline__ uint64_t packed_window16(const uint8_t *seq, int64_t l_seq, int64_t p_first, int dir, int comp)
{
	const int64_t lo = dir > 0 ? p_first : p_first - 15;
	const int64_t seq_bytes = (l_seq + 1) >> 1, amax = (seq_bytes + 4) & ~(int64_t)3;
	int64_t a0 = (lo >> 1) & ~(int64_t)3;
	a0 = a0 < 0 ? 0 : a0 > amax ? amax : a0;
	const uint32_t *wp = (const uint32_t*)(seq + a0);
	const uint32_t W0 = wp[0], W1 = wp[1], W2 = wp[2];
	const int64_t n0 = lo - 2 * a0;                                // first nibble (0..7 unless the address was clamped)
	const uint64_t lo64 = (uint64_t)W0 | (uint64_t)W1 << 32;
	uint64_t nib;
	if (n0 >= 0 && n0 <= 7) nib = n0 ? (lo64 >> (4 * n0)) | ((uint64_t)W2 << (64 - 4 * n0)) : lo64;
	else if (n0 < 0 && n0 >= -15) nib = lo64 << (4 * (-n0));
	else nib = 0;
	if (dir < 0) {                                                 // descending: reverse the sixteen nibbles
		nib = ((nib & 0x0f0f0f0f0f0f0f0fULL) << 4) | ((nib >> 4) & 0x0f0f0f0f0f0f0f0fULL);
		nib = __builtin_bswap64(nib);
	}
	if (comp) {                                                    // minus strand: complement the codes below 4 (3 - b = b ^ 3)
		const uint64_t m = ~((nib >> 2) | (nib >> 3)) & 0x1111111111111111ULL;
		nib ^= m * 3;
	}
	return nib;
}

// donor/acceptor/nas of window row i (the window fetch, ntseq.c:89-114 -- 4-bit unpack, reverse complement -- is folded into the
// addressing).  Forward: ns_prep_seq (nasw-sse.c:106-155); left extension:
// ns_prep_seq_left (nasw-sse.c:157-210).  int8 wrap-around of the reference's arrays is preserved.
// Four rows per thread (rows i, i + 256, i + 512, i + 768 of a 1 024-row chunk): the kernel is bound by the chain of dependent loads
// chunk -> task -> contig -> genome words and by how many waves the chip holds at once, not by bandwidth or arithmetic (round 4:
// 1.8 ms alone for 200 M rows = 0.5 TB/s, and 2-6 x that next to the other kernels of the stream, on every round's critical
// path) -- a thread that fetches the words of four rows together pays the chain once for four records.
#define MPA_PREP_ROWS (MPA_PREP_CHUNK_ROWS / 256)
__global__ __launch_bounds__(256) void k_prep_rows(DevGenome g, const DTask *tasks, const PrepChunk *chunks, uint32_t *rec, DpConst c, DevTables tb)
{
	MPA_SHORT_KERNEL();
	const PrepChunk ch = chunks[blockIdx.x];
	const DTask t = tasks[ch.task];
	const int32_t nl = t.nl;
	const int cid = t.vid >> 1, rev = t.vid & 1, left = (t.flag & 2) != 0;
	const int64_t off = g.ctg_off[cid], len = g.ctg_len[cid];
	// w[j] = b[i-6+j], j=0..15, where b[] is the (possibly reversed) window; 15 = outside the window.
	// The sixteen bases are consecutive in the packed genome (ascending, or descending for a reversed window / the minus strand):
	// three aligned words cover them, and the sixteen nibbles are shifted out of those -- instead of sixteen dependent byte loads
	// with 64-bit address arithmetic each.  The genome buffer is padded by 16 bytes; what lies outside the
	// window is masked to 15 whatever was read for it.
	uint64_t nibs[MPA_PREP_ROWS];
	const int64_t x0 = t.nt_off + (left ? nl - 1 : 0);             // strand position of window position 0
	const int64_t p0 = rev ? off + len - 1 - x0 : off + x0;        // ... its position in the packed genome
	const int dir = (left != (rev != 0)) ? -1 : 1;
#pragma unroll
	for (int r = 0; r < MPA_PREP_ROWS; ++r) {
		const int32_t i = ch.row0 + (int32_t)threadIdx.x + 256 * r;
		nibs[r] = i < nl ? packed_window16(g.seq, g.l_seq, p0 + (int64_t)dir * (i - 6), dir, rev) : 0;
	}
#pragma unroll
	for (int r = 0; r < MPA_PREP_ROWS; ++r) {
	const int32_t i = ch.row0 + (int32_t)threadIdx.x + 256 * r;
	if (i >= nl) continue;
	uint64_t nib = nibs[r];
	{
		const int32_t jlo = 6 - i > 0 ? (6 - i > 16 ? 16 : 6 - i) : 0;                        // w[j], j < jlo: before the window
		const int32_t jhi = nl - i + 6 < 16 ? (nl - i + 6 < 0 ? 0 : nl - i + 6) : 16;       // w[j], j >= jhi: behind it
		uint64_t valid = jhi >= 16 ? ~0ULL : ((1ULL << (4 * jhi)) - 1);
		valid &= jlo >= 16 ? 0ULL : ~((1ULL << (4 * jlo)) - 1);
		nib = (nib & valid) | ~valid;
	}
	uint32_t w[16];
#pragma unroll
	for (int j = 0; j < 16; ++j) w[j] = (uint32_t)(nib >> (4 * j)) & 15u;
#define B_(d) w[6 + (d)]   /* b[i+d] */
	uint32_t nas = 21;
	int32_t don, acc;
	const int32_t sp3 = (int8_t)c.sp[3];
	if (!left) {
		if (i >= 2 && B_(0) < 4 && B_(-1) < 4 && B_(-2) < 4) nas = tb.codon[B_(-2) << 4 | B_(-1) << 2 | B_(0)];
		{ // donor[k], k = i+1
			const int32_t k = i + 1;
			don = sp3;
			if (k < nl - 3) {
				int tt = 3;
				if (B_(2) == 2 && B_(3) == 3) tt = (B_(4) == 0 || B_(4) == 2) ? (B_(1) == 2 ? -1 : 4) : 0;
				else if (B_(2) == 2 && B_(3) == 1 && B_(1) == 2) tt = 1;
				else if (B_(2) == 0 && B_(3) == 3) tt = 2;
				don = tt < 0 ? 0 : (int8_t)c.sp[tt];
			}
		}
		{ // acceptor[i]
			acc = sp3;
			if (i >= 1) {
				int tt = 3, pen_y = 0;
				if (B_(-1) == 0 && B_(0) == 2) {
					tt = (i >= 2 && (B_(-2) == 1 || B_(-2) == 3)) ? -1 : 0;
#pragma unroll
					for (int d = 4; d <= 6; ++d)
						if (i - d >= 0 && B_(-d) != 1 && B_(-d) != 3) pen_y += c.sp[5];
				} else if (B_(-1) == 0 && B_(0) == 1) tt = 2;
				acc = tt < 0 ? 0 : (int8_t)c.sp[tt];
				if (tt == -1 || tt == 0) acc = (int8_t)(acc + pen_y);
			}
		}
	} else {
		if (i >= 2 && B_(0) < 4 && B_(-1) < 4 && B_(-2) < 4) nas = tb.codon[B_(0) << 4 | B_(-1) << 2 | B_(-2)];
		{ // "donor"[k] of the reversed string, k = i+1 (really the acceptor signal read backwards)
			const int32_t k = i + 1;
			don = sp3;
			if (k < nl - 3) {
				int tt = 3, pen_y = 0;
				if (B_(2) == 2 && B_(3) == 0) {
					tt = (B_(4) == 1 || B_(4) == 3) ? -1 : 0;
#pragma unroll
					for (int d = 6; d <= 8; ++d)       // j = k+5 .. k+7  ->  b[i+6 .. i+8]
						if (i + d < nl && B_(d) != 1 && B_(d) != 3) pen_y += c.sp[5];
				} else if (B_(2) == 1 && B_(3) == 0) tt = 2;
				don = tt < 0 ? 0 : (int8_t)c.sp[tt];
				if (tt == -1 || tt == 0) don = (int8_t)(don + pen_y);
			}
		}
		{ // "acceptor"[i] of the reversed string (really the donor signal read backwards)
			acc = sp3;
			if (i >= 1) {
				int tt = 3;
				if (B_(-1) == 3 && B_(0) == 2)
					tt = (i >= 2 && (B_(-2) == 0 || B_(-2) == 2)) ? ((i + 1 < nl && B_(1) == 2) ? -1 : 4) : 0;
				else if (B_(-1) == 1 && B_(0) == 2 && i + 1 < nl && B_(1) == 1) tt = 1;
				else if (B_(-1) == 3 && B_(0) == 0) tt = 2;
				acc = tt < 0 ? 0 : (int8_t)c.sp[tt];
			}
		}
	}
#undef B_
	if (g.spsc) {
		// --spsc: the score track adjusts the penalties (nasw-sse.c:138-152, left: 189-203).  ss[p] belongs to window position p
		// (forward); forward calls shift it onto donor/acceptor[p-1], the reversed left extension onto [nl-1-p] with the two
		// roles swapped.  int8 wrap-around as in the reference's arrays.
		const uint8_t *track = g.spsc + (rev ? g.l_seq : 0) + off;
		const int32_t max_spsc = (t.io + 1) / 2 - 1;
		auto ss_at = [&](int32_t p) -> int32_t {             // -1: outside what the reference reads
			if (p < (left ? 0 : 1) || p >= nl) return -1;
			if (p == 0 && (t.flag & 8)) return 0xff;           // MPA_F_SS_SKIP0
			return track[t.nt_off + p];
		};
		auto adjust = [&](int32_t v, int32_t ss, bool for_acceptor_entry) -> int32_t {
			if (ss < 0) return v;
			if (ss == 0xff) return (int8_t)(v - c.sp_null_bonus);
			if (((ss & 1) != 0) != for_acceptor_entry) return v;
			int32_t spsc = (int8_t)(ss >> 1) - 64;
			if (spsc > max_spsc) spsc = max_spsc;
			return (int8_t)(v - spsc);
		};
		if (!left) {
			don = adjust(don, ss_at(i + 2), false);          // donor[i+1] <- ss[i+2] (donor entries)
			acc = adjust(acc, ss_at(i + 1), true);           // acceptor[i] <- ss[i+1] (acceptor entries)
		} else {
			don = adjust(don, ss_at(nl - 2 - i), true);      // "donor"[i+1] of the reversed string <- acceptor entries at nl-1-(i+1)
			acc = adjust(acc, ss_at(nl - 1 - i), false);     // "acceptor"[i] <- donor entries at nl-1-i
		}
	}
	// (byte 2: the row's gap extension -- fs on a stop codon, nasw-sse.c:263,370 -- or, when the penalties do not fit a byte, just the stop flag)
	rec[t.rec_off + i] = make_rec(nas * PROF_AA_STRIDE_REC, don, c.wide_ge ? (uint32_t)(nas == 20) : nas == 20 ? (uint32_t)c.fs : (uint32_t)c.ge, acc);
	}
}

// query profile prof[a][col] = mat[a][aa(col)] (ns_gen_prof nasw-sse.c:212-224); columns >= al score -32768
__global__ __launch_bounds__(256) void k_prep_prof(const DTask *tasks, const char *qseq, int16_t *prof, DevTables tb)
{
	MPA_SHORT_KERNEL();
	const DTask t = tasks[blockIdx.x];
	const int left = (t.flag & 2) != 0;
	for (int idx = threadIdx.x; idx < 22 * t.pw; idx += blockDim.x) {
		int a = idx / t.pw, col = idx - a * t.pw;
		int16_t v = NEG16;
		if (col < t.al) {
			uint8_t ch = (uint8_t)qseq[t.q_off + (left ? t.al - 1 - col : col)];
			v = tb.mat[a * 22 + tb.aa20[ch]];
		}
		prof[t.prof_off + idx] = v;
	}
}

// ------------------------------------------------------------------------------------------------
// K1: extension (score only).  See file header.
//
// What one row costs (round 6, tools/ubench/valu_rate.hip on the MI355X): every packed-int16 instruction, DPP move, v_perm,
// SDWA or VOP3 instruction takes 4.15 cycles of its SIMD's VALU (only plain 32-bit VOP2 adds / logic ops take 2.3), and a wave
// that is alone on its SIMD issues ONE instruction of any kind (VALU, SALU, s_nop, LDS) per ~5 cycles.  A saturated machine
// is therefore bound by the VALU instructions per row, a lone wave (the 100 000-row calls that set a round's length) by ALL
// instructions per row -- including the `s_nop 0` hipcc puts between a packed instruction and any VALU instruction that reads
// its result (LLVM's dst_sel forwarding hazard, which takes VOP3P's default op_sel_hi bit for VOP3's dst op_sel;
// tools/ubench/pk_hazard.hip: 524 288 lanes x 144 000 dependent packed instructions, identical with and without the nop) and
// the `s_nop 1` in front of every DPP move (that hazard is real: two wait states between a VALU write and a DPP read).
// Round 5's row was ~60 VALU + ~17 s_nop + ~20 SALU = ~95 slots (255 ns alone, 554 ns per wave-row on a saturated machine).
//
// The row is now two hand-scheduled asm blocks (`row_head`, `row_tail<G>`; 45 VALU + 1 s_nop for 32 lanes per call):
//   * the row records arrive DECODED: the lanes that refill the LDS ring (12 rows at a time) split a record into packed
//     {donor, acceptor, gap extension, profile offsets} once, so a row reads one ds_read_b128 and does no v_perm;
//   * the four frameshift moves use M(r) = max(H(r, j), H(r, j-1)) kept per row (one max when the row is made);
//   * the horizontal gap chain is a scan inside rows of 16 lanes, then ONE exclusive step (row_shr:1 + row_bcast:15), which
//     also removes the lane-32 select of a 32-lane group; all fill registers are kept (a DPP move leaves lanes without a
//     source untouched);
//   * the two wait states in front of each DPP move are filled with the part of the NEXT row that does not depend on this one
//     (its D state, its diagonal term, the LDS addresses of the row after), so no slot is idle;
//   * the best-row / x-drop vote is a clamp and a compare: every lane carries thresholds (lo, hi) such that lo <= H <= hi
//     proves that the row neither beats the call's best score nor falls below its x-drop line (see repack()).
// The exact bookkeeping (nasw-sse.c:423-433), task ends, steps of the length penalty and the first rows run through the
// plain C++ row below ("generic row"), which shares all state with the asm rows.
// ------------------------------------------------------------------------------------------------
struct ExtArgs {
	const DTask *tasks;
	const ExtWave *waves;
	const uint32_t *rec;
	const int16_t *prof;
	ExtOut *out;
	DpConst c;
	PenTable pen;
	// checkpointed traceback (ext_narrow<G, true>; dp_device.h): extension-bit words, checkpoints, H(nl-1, al-1) per call
	uint32_t *lite;
	uint32_t *ckpt;
	int32_t *score;
};

#define PROF_AA_STRIDE 2                       /* bytes per amino acid inside one column of the LDS profile */
#define PROF_COL_STRIDE 46                     /* bytes per column: 22 scores + 1 pad, an odd number of 16-bit words */

// two int16 profile scores into the halves of one register
__device__ __forceinline__ uint32_t prof2(const char *lo, const char *hi)
{
	v2s r;
	r.x = *(const int16_t*)lo, r.y = *(const int16_t*)hi;
	return as_u(r);
}
__device__ __forceinline__ uint32_t p_min(uint32_t a, uint32_t b) { return as_u(__builtin_elementwise_min(as_v(a), as_v(b))); }

// One decoded row record of a lane group in the LDS ring: the two calls (int16 halves) of the group side by side.
//   x donor[r+1] (sign-extended)   y acceptor[r]   z gap extension of the row   w byte offset of the row's amino acid in a profile
//   column, call 0 in the low half, call 1 in the high half
__device__ __forceinline__ uint4 ring_entry(const uint32_t r0, const uint32_t r1)
{
	return make_uint4(__builtin_amdgcn_perm(r1, r0, 0x0a050801u), __builtin_amdgcn_perm(r1, r0, 0x0b070903u),
	                  __builtin_amdgcn_perm(r1, r0, 0x0c060c02u), __builtin_amdgcn_perm(r1, r0, 0x0c040c00u));
}
#define EXT_RING_ROWS 24                       /* two blocks of 12 rows: one is read while the other is refilled */
#define EXT_RING_SLOTS 25                      /* + a copy of slot 0 behind slot 23: the asm rows fetch three rows ahead with a constant offset */
#define EXT_BLOCK_ROWS 12
#define EXT_NARROW_LDS ((size_t)2 * 64 * PROF_COL_STRIDE + (size_t)4 * EXT_RING_SLOTS * 16 + 8 * 32 + 2 * MPA_PEN_MAX * 4)   /* bytes per wave; a multiple of 16 */
typedef const __attribute__((address_space(3))) int16_t *lds_s16p;
// two int16 profile scores (absolute LDS byte addresses) into the halves of one register
__device__ __forceinline__ uint32_t prof2s(const uint32_t lo, const uint32_t hi)
{
	v2s r;
	r.x = *(lds_s16p)(uintptr_t)lo, r.y = *(lds_s16p)(uintptr_t)hi;
	return as_u(r);
}

// The first asm block of a row: the three intron states, the frameshift moves and the choice among everything but the
// horizontal gap.  hD = max(H(i-3,j-1) + s(i,j), D(i,j)) was made during the row before.  x = h + j*ge feeds the scan.
#define MPA_ROW_HEAD(Hr1, Hs1, dn1, dn2, dn3, ac1, ac2, ac3, M1, M2) \
	asm volatile( \
		"v_pk_sub_i16 %[uA], %[hr1], %[io] clamp\n\t" \
		"v_pk_sub_i16 %[uB], %[hs1], %[io] clamp\n\t" \
		"v_pk_sub_i16 %[uA], %[uA], %[d2] clamp\n\t" \
		"v_pk_sub_i16 %[t1], %[uB], %[d1] clamp\n\t" \
		"v_pk_sub_i16 %[uB], %[uB], %[d3] clamp\n\t" \
		"v_pk_max_i16 %[A], %[uA], %[A]\n\t" \
		"v_pk_max_i16 %[B], %[t1], %[B]\n\t" \
		"v_pk_max_i16 %[C], %[uB], %[C]\n\t" \
		"v_pk_sub_i16 %[uA], %[A], %[a3] clamp\n\t" \
		"v_pk_sub_i16 %[t1], %[B], %[a2] clamp\n\t" \
		"v_pk_sub_i16 %[uB], %[C], %[a1] clamp\n\t" \
		"v_pk_max_i16 %[uA], %[uA], %[t1]\n\t" \
		"v_pk_max_i16 %[t1], %[m1], %[m2]\n\t" \
		"v_pk_sub_i16 %[t1], %[t1], %[fs] clamp\n\t" \
		"v_pk_max_i16 %[uB], %[uB], %[t1]\n\t" \
		"v_pk_max_i16 %[h], %[hD], %[uA]\n\t" \
		"v_pk_max_i16 %[h], %[h], %[uB]\n\t" \
		"v_pk_add_i16 %[x], %[h], %[jge] clamp" \
		: [uA] "=&v"(tA_), [uB] "=&v"(tB_), [t1] "=&v"(tC_), [h] "=&v"(h), [x] "=&v"(x), [A] "+v"(A), [B] "+v"(B), [C] "+v"(C) \
		: [hr1] "v"(Hr1), [hs1] "v"(Hs1), [io] "v"(ioP), [d1] "v"(dn1), [d2] "v"(dn2), [d3] "v"(dn3), [a1] "v"(ac1), [a2] "v"(ac2), [a3] "v"(ac3), \
		  [m1] "v"(M1), [m2] "v"(M2), [hD] "v"(hD), [jge] "v"(jge), [fs] "s"(fsP))

// The second asm block: the gap scan with the next row's independent part in its wait states (D state from H(i-2), Dr(i-2) and the
// next record's gap extension; the diagonal term from Hs(i-2) and the two profile scores fetched a row ago; the LDS addresses of
// the scores of the row after from the record fetched at the top of this row), the final H, the vote clamp, H shifted by one
// column, and M.  Wait states: a DPP move needs two instructions between it and the VALU write of its source.
#define MPA_SCAN4 \
		"v_pk_sub_i16 %[t2], %[hr2], %[go] clamp\n\t" \
		"v_pk_max_i16 %[t2], %[t2], %[dr2]\n\t" \
		"v_mov_b32_dpp %[k0], %[x] row_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
		"v_pk_max_i16 %[x], %[x], %[k0]\n\t" \
		"v_pk_sub_i16 %[dr2], %[t2], %[gei] clamp\n\t" \
		"v_perm_b32 %[t2], %[s1], %[s0], %[sel]\n\t" \
		"v_mov_b32_dpp %[k1], %[x] row_shr:2 row_mask:0xf bank_mask:0xf\n\t" \
		"v_pk_max_i16 %[x], %[x], %[k1]\n\t" \
		"v_pk_add_i16 %[t2], %[hs2], %[t2] clamp\n\t" \
		"v_pk_max_i16 %[hDn], %[t2], %[dr2]\n\t" \
		"v_mov_b32_dpp %[k2], %[x] row_shr:4 row_mask:0xf bank_mask:0xf\n\t" \
		"v_pk_max_i16 %[x], %[x], %[k2]\n\t" \
		"v_add_u32_sdwa %[a0], %[pb0], %[aw] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n\t" \
		"v_add_u32_sdwa %[a1], %[pb1], %[aw] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t" \
		"v_mov_b32_dpp %[k3], %[x] row_shr:8 row_mask:0xf bank_mask:0xf\n\t" \
		"v_pk_max_i16 %[x], %[x], %[k3]\n\t"
#define MPA_EX16 \
		"s_nop 1\n\t" \
		"v_mov_b32_dpp %[ke], %[x] row_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
		"v_pk_sub_i16 %[t2], %[ke], %[gojge] clamp\n\t"
#define MPA_EX32 \
		"s_nop 1\n\t" \
		"v_mov_b32_dpp %[ke], %[x] row_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
		"v_mov_b32_dpp %[kc], %[x] row_bcast:15 row_mask:0xa bank_mask:0xf\n\t" \
		"v_pk_max_i16 %[t2], %[ke], %[kc]\n\t" \
		"v_pk_sub_i16 %[t2], %[t2], %[gojge] clamp\n\t"
#define MPA_EX64 \
		"v_mov_b32 %[t2], 0x80008000\n\t" \
		"s_nop 0\n\t" \
		"v_mov_b32_dpp %[t2], %[x] row_bcast:15 row_mask:0xa bank_mask:0xf\n\t" \
		"v_pk_max_i16 %[x], %[x], %[t2]\n\t" \
		"v_mov_b32 %[t2], 0x80008000\n\t" \
		"s_nop 0\n\t" \
		"v_mov_b32_dpp %[t2], %[x] row_bcast:31 row_mask:0xc bank_mask:0xf\n\t" \
		"v_pk_max_i16 %[x], %[x], %[t2]\n\t" \
		"s_nop 1\n\t" \
		"v_mov_b32_dpp %[ke], %[x] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
		"v_pk_sub_i16 %[t2], %[ke], %[gojge] clamp\n\t"
#define MPA_FIN \
		"v_pk_max_i16 %[h], %[h], %[t2]\n\t" \
		"v_pk_min_i16 %[vt], %[h], %[gl]\n\t" \
		"v_pk_max_i16 %[vt], %[vt], %[tl]\n\t"
#define MPA_HS16 "v_mov_b32_dpp %[hs3], %[h] row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define MPA_HS32 "v_mov_b32_dpp %[hs3], %[h] wave_shr:1 row_mask:0xb bank_mask:0xf\n\tv_mov_b32_dpp %[hs3], %[h] row_shr:1 row_mask:0x4 bank_mask:0xf\n\t"
#define MPA_HS64 "v_mov_b32_dpp %[hs3], %[h] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define MPA_ROW_TAIL(EX, HS, Hr2, Hs2, Dr2, Hs3, M3) \
	asm volatile(MPA_SCAN4 EX MPA_FIN HS "v_pk_max_i16 %[m3], %[h], %[hs3]" \
		: [x] "+v"(x), [h] "+v"(h), [k0] "+v"(k0), [k1] "+v"(k1), [k2] "+v"(k2), [k3] "+v"(k3), [ke] "+v"(ke), [kc] "+v"(kc), [dr2] "+v"(Dr2), [hs3] "+v"(Hs3), \
		  [t2] "=&v"(tA_), [hDn] "=&v"(hDn), [a0] "=&v"(a0), [a1] "=&v"(a1), [vt] "=&v"(vt), [m3] "=&v"(M3) \
		: [hr2] "v"(Hr2), [gei] "v"(rnext.z), [s0] "v"(sraw0), [s1] "v"(sraw1), [hs2] "v"(Hs2), [pb0] "v"(pb0), [pb1] "v"(pb1), [aw] "v"(rnn.w), \
		  [gojge] "v"(gojge), [gl] "v"(GvL), [tl] "v"(TvL), [go] "s"(goP), [sel] "s"(sel_lo))

// DUAL extension rows (one call of 65..128 columns per wave, see MPA_EX64_LD below): the tail with the vote and the two carries
#define MPA_ROW_TAIL_D(Hr2, Hs2, Dr2, Hs3, M3) \
	asm volatile(MPA_SCAN4 \
		"v_mov_b32 %[t2], 0x80008000\n\t" \
		"s_nop 0\n\t" \
		"v_mov_b32_dpp %[t2], %[x] row_bcast:15 row_mask:0xa bank_mask:0xf\n\t" \
		"v_pk_max_i16 %[x], %[x], %[t2]\n\t" \
		"v_mov_b32 %[t2], 0x80008000\n\t" \
		"s_nop 0\n\t" \
		"v_mov_b32_dpp %[t2], %[x] row_bcast:31 row_mask:0xc bank_mask:0xf\n\t" \
		"v_pk_max_i16 %[x], %[x], %[t2]\n\t" \
		"s_nop 0\n\t" \
		"v_readlane_b32 %[st], %[x], 63\n\t" \
		"s_lshl_b32 %[st], %[st], 16\n\t" \
		"s_or_b32 %[st], %[st], 0x8000\n\t" \
		"v_mov_b32_dpp %[ke], %[x] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
		"v_pk_max_i16 %[t2], %[ke], %[st]\n\t" \
		"v_pk_sub_i16 %[t2], %[t2], %[gojge] clamp\n\t" \
		"v_pk_max_i16 %[h], %[h], %[t2]\n\t" \
		"v_pk_min_i16 %[vt], %[h], %[gl]\n\t" \
		"v_pk_max_i16 %[vt], %[vt], %[tl]\n\t" \
		"v_readlane_b32 %[st], %[h], 63\n\t" \
		"s_lshl_b32 %[st], %[st], 16\n\t" \
		"s_or_b32 %[st], %[st], 0x8000\n\t" \
		"v_mov_b32_dpp %[hs3], %[h] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
		"v_writelane_b32 %[hs3], %[st], 0\n\t" \
		"v_pk_max_i16 %[m3], %[h], %[hs3]" \
		: [x] "+v"(x), [h] "+v"(h), [k0] "+v"(k0), [k1] "+v"(k1), [k2] "+v"(k2), [k3] "+v"(k3), [ke] "+v"(ke), [dr2] "+v"(Dr2), [hs3] "+v"(Hs3), \
		  [t2] "=&v"(tA_), [hDn] "=&v"(hDn), [a0] "=&v"(a0), [a1] "=&v"(a1), [vt] "=&v"(vt), [m3] "=&v"(M3), [st] "=&s"(stmp) \
		: [hr2] "v"(Hr2), [gei] "v"(rnext.z), [s0] "v"(sraw0), [s1] "v"(sraw1), [hs2] "v"(Hs2), [pb0] "v"(pb0), [pb1] "v"(pb1), [aw] "v"(rnn.w), \
		  [gojge] "v"(gojge), [gl] "v"(GvL), [tl] "v"(TvL), [go] "s"(goP), [sel] "s"(sel_lo) \
		: "scc")
// The rows of the checkpointed traceback's sweep (ext_narrow<G, true>): the same two blocks without the vote, with the four
// "fresh value minus carried value" differences whose signs are the extension bits of the reference's traceback word (bit 5: D(i-3,j)
// > H(i-3,j) - q, nasw-sse.c:455; bits 6..8: A, B, C carried rather than opened, :462-485), and with the new D state in a register
// of its own (the old one is what a checkpoint stores).  dD belongs to the NEXT row, like the D state it comes with.
#define MPA_ROW_HEAD_L(Hr1, Hs1, dn1, dn2, dn3, ac1, ac2, ac3, M1, M2) \
	asm volatile( \
		"v_pk_sub_i16 %[uA], %[hr1], %[io] clamp\n\t" \
		"v_pk_sub_i16 %[uB], %[hs1], %[io] clamp\n\t" \
		"v_pk_sub_i16 %[uA], %[uA], %[d2] clamp\n\t" \
		"v_pk_sub_i16 %[t1], %[uB], %[d1] clamp\n\t" \
		"v_pk_sub_i16 %[uB], %[uB], %[d3] clamp\n\t" \
		"v_pk_sub_i16 %[dA], %[uA], %[A] clamp\n\t" \
		"v_pk_sub_i16 %[dB], %[t1], %[B] clamp\n\t" \
		"v_pk_sub_i16 %[dC], %[uB], %[C] clamp\n\t" \
		"v_pk_max_i16 %[A], %[uA], %[A]\n\t" \
		"v_pk_max_i16 %[B], %[t1], %[B]\n\t" \
		"v_pk_max_i16 %[C], %[uB], %[C]\n\t" \
		"v_pk_sub_i16 %[uA], %[A], %[a3] clamp\n\t" \
		"v_pk_sub_i16 %[t1], %[B], %[a2] clamp\n\t" \
		"v_pk_sub_i16 %[uB], %[C], %[a1] clamp\n\t" \
		"v_pk_max_i16 %[uA], %[uA], %[t1]\n\t" \
		"v_pk_max_i16 %[t1], %[m1], %[m2]\n\t" \
		"v_pk_sub_i16 %[t1], %[t1], %[fs] clamp\n\t" \
		"v_pk_max_i16 %[uB], %[uB], %[t1]\n\t" \
		"v_pk_max_i16 %[h], %[hD], %[uA]\n\t" \
		"v_pk_max_i16 %[h], %[h], %[uB]\n\t" \
		"v_pk_add_i16 %[x], %[h], %[jge] clamp" \
		: [uA] "=&v"(tA_), [uB] "=&v"(tB_), [t1] "=&v"(tC_), [h] "=&v"(h), [x] "=&v"(x), [dA] "=&v"(dA), [dB] "=&v"(dB), [dC] "=&v"(dC), \
		  [A] "+v"(A), [B] "+v"(B), [C] "+v"(C) \
		: [hr1] "v"(Hr1), [hs1] "v"(Hs1), [io] "v"(ioP), [d1] "v"(dn1), [d2] "v"(dn2), [d3] "v"(dn3), [a1] "v"(ac1), [a2] "v"(ac2), [a3] "v"(ac3), \
		  [m1] "v"(M1), [m2] "v"(M2), [hD] "v"(hD), [jge] "v"(jge), [fs] "s"(fsP))
#define MPA_SCAN4_L \
		"v_pk_sub_i16 %[t2], %[hr2], %[go] clamp\n\t" \
		"v_pk_sub_i16 %[dDn], %[t2], %[dr2] clamp\n\t" \
		"v_mov_b32_dpp %[k0], %[x] row_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
		"v_pk_max_i16 %[x], %[x], %[k0]\n\t" \
		"v_pk_max_i16 %[t2], %[t2], %[dr2]\n\t" \
		"v_pk_sub_i16 %[drn], %[t2], %[gei] clamp\n\t" \
		"v_mov_b32_dpp %[k1], %[x] row_shr:2 row_mask:0xf bank_mask:0xf\n\t" \
		"v_pk_max_i16 %[x], %[x], %[k1]\n\t" \
		"v_perm_b32 %[t2], %[s1], %[s0], %[sel]\n\t" \
		"v_pk_add_i16 %[t2], %[hs2], %[t2] clamp\n\t" \
		"v_mov_b32_dpp %[k2], %[x] row_shr:4 row_mask:0xf bank_mask:0xf\n\t" \
		"v_pk_max_i16 %[x], %[x], %[k2]\n\t" \
		"v_pk_max_i16 %[hDn], %[t2], %[drn]\n\t" \
		"v_add_u32_sdwa %[a0], %[pb0], %[aw] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n\t" \
		"v_mov_b32_dpp %[k3], %[x] row_shr:8 row_mask:0xf bank_mask:0xf\n\t" \
		"v_pk_max_i16 %[x], %[x], %[k3]\n\t" \
		"v_add_u32_sdwa %[a1], %[pb1], %[aw] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t"
#define MPA_EX16_L \
		"s_nop 0\n\t" \
		"v_mov_b32_dpp %[ke], %[x] row_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
		"v_pk_sub_i16 %[t2], %[ke], %[gojge] clamp\n\t"
#define MPA_EX32_L \
		"s_nop 0\n\t" \
		"v_mov_b32_dpp %[ke], %[x] row_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
		"v_mov_b32_dpp %[kc], %[x] row_bcast:15 row_mask:0xa bank_mask:0xf\n\t" \
		"v_pk_max_i16 %[t2], %[ke], %[kc]\n\t" \
		"v_pk_sub_i16 %[t2], %[t2], %[gojge] clamp\n\t"
#define MPA_EX64_L \
		"v_mov_b32 %[t2], 0x80008000\n\t" \
		"v_mov_b32_dpp %[t2], %[x] row_bcast:15 row_mask:0xa bank_mask:0xf\n\t" \
		"v_pk_max_i16 %[x], %[x], %[t2]\n\t" \
		"v_mov_b32 %[t2], 0x80008000\n\t" \
		"s_nop 0\n\t" \
		"v_mov_b32_dpp %[t2], %[x] row_bcast:31 row_mask:0xc bank_mask:0xf\n\t" \
		"v_pk_max_i16 %[x], %[x], %[t2]\n\t" \
		"s_nop 1\n\t" \
		"v_mov_b32_dpp %[ke], %[x] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
		"v_pk_sub_i16 %[t2], %[ke], %[gojge] clamp\n\t"
// DUAL (one call of 65..128 columns in a wave: column c in the low half of lane c, column c + 64 in the high half of the same lane):
// the gap scan's total over the low halves (lane 63) is the carry into every high half, and H of column 63 is the left neighbour
// of column 64 (lane 0, high half); one v_readlane each (one wait state behind the VALU write of what it reads: gfx940 family),
// the two scalar instructions that move the value into the high half sit in the DPP move's wait states.
#define MPA_EX64_LD \
		"v_mov_b32 %[t2], 0x80008000\n\t" \
		"v_mov_b32_dpp %[t2], %[x] row_bcast:15 row_mask:0xa bank_mask:0xf\n\t" \
		"v_pk_max_i16 %[x], %[x], %[t2]\n\t" \
		"v_mov_b32 %[t2], 0x80008000\n\t" \
		"s_nop 0\n\t" \
		"v_mov_b32_dpp %[t2], %[x] row_bcast:31 row_mask:0xc bank_mask:0xf\n\t" \
		"v_pk_max_i16 %[x], %[x], %[t2]\n\t" \
		"s_nop 0\n\t" \
		"v_readlane_b32 %[st], %[x], 63\n\t" \
		"s_lshl_b32 %[st], %[st], 16\n\t" \
		"s_or_b32 %[st], %[st], 0x8000\n\t" \
		"v_mov_b32_dpp %[ke], %[x] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
		"v_pk_max_i16 %[t2], %[ke], %[st]\n\t" \
		"v_pk_sub_i16 %[t2], %[t2], %[gojge] clamp\n\t"
#define MPA_FIN_HS64_LD \
		"v_pk_max_i16 %[h], %[h], %[t2]\n\t" \
		"s_nop 0\n\t" \
		"v_readlane_b32 %[st], %[h], 63\n\t" \
		"s_lshl_b32 %[st], %[st], 16\n\t" \
		"s_or_b32 %[st], %[st], 0x8000\n\t" \
		"v_mov_b32_dpp %[hs3], %[h] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
		"v_writelane_b32 %[hs3], %[st], 0\n\t"
#define MPA_ROW_TAIL_LD(Hr2, Hs2, Dr2, Hs3, M3) \
	asm volatile(MPA_SCAN4_L MPA_EX64_LD MPA_FIN_HS64_LD "v_pk_max_i16 %[m3], %[h], %[hs3]" \
		: [x] "+v"(x), [h] "+v"(h), [k0] "+v"(k0), [k1] "+v"(k1), [k2] "+v"(k2), [k3] "+v"(k3), [ke] "+v"(ke), [hs3] "+v"(Hs3), \
		  [t2] "=&v"(tA_), [hDn] "=&v"(hDn), [a0] "=&v"(a0), [a1] "=&v"(a1), [m3] "=&v"(M3), [drn] "=&v"(drn), [dDn] "=&v"(dDn), [st] "=&s"(stmp) \
		: [hr2] "v"(Hr2), [dr2] "v"(Dr2), [gei] "v"(rnext.z), [s0] "v"(sraw0), [s1] "v"(sraw1), [hs2] "v"(Hs2), [pb0] "v"(pb0), [pb1] "v"(pb1), [aw] "v"(rnn.w), \
		  [gojge] "v"(gojge), [go] "s"(goP), [sel] "s"(sel_lo) \
		: "scc")
#define MPA_ROW_TAIL_L(EX, HS, Hr2, Hs2, Dr2, Hs3, M3) \
	asm volatile(MPA_SCAN4_L EX "v_pk_max_i16 %[h], %[h], %[t2]\n\ts_nop 1\n\t" HS "v_pk_max_i16 %[m3], %[h], %[hs3]" \
		: [x] "+v"(x), [h] "+v"(h), [k0] "+v"(k0), [k1] "+v"(k1), [k2] "+v"(k2), [k3] "+v"(k3), [ke] "+v"(ke), [kc] "+v"(kc), [hs3] "+v"(Hs3), \
		  [t2] "=&v"(tA_), [hDn] "=&v"(hDn), [a0] "=&v"(a0), [a1] "=&v"(a1), [m3] "=&v"(M3), [drn] "=&v"(drn), [dDn] "=&v"(dDn) \
		: [hr2] "v"(Hr2), [dr2] "v"(Dr2), [gei] "v"(rnext.z), [s0] "v"(sraw0), [s1] "v"(sraw1), [hs2] "v"(Hs2), [pb0] "v"(pb0), [pb1] "v"(pb1), [aw] "v"(rnn.w), \
		  [gojge] "v"(gojge), [go] "s"(goP), [sel] "s"(sel_lo))
// the row's nibble (D | A << 1 | B << 2 | C << 3 in bits 0..3 of each half) from the signs of the four differences, pushed into the
// three-row accumulator (FIRST: the row opens a new word)
#define MPA_NIBBLE(LAST_OP) \
	asm volatile( \
		"v_pk_ashrrev_i16 %[u], 15, %[dD] op_sel_hi:[0,1]\n\t" \
		"v_pk_ashrrev_i16 %[v], 15, %[dA] op_sel_hi:[0,1]\n\t" \
		"v_and_b32 %[u], %[c1], %[u]\n\t" \
		"v_pk_ashrrev_i16 %[w], 15, %[dB] op_sel_hi:[0,1]\n\t" \
		"v_and_or_b32 %[u], %[v], %[c2], %[u]\n\t" \
		"v_pk_ashrrev_i16 %[v], 15, %[dC] op_sel_hi:[0,1]\n\t" \
		"v_and_or_b32 %[u], %[w], %[c4], %[u]\n\t" \
		LAST_OP \
		: [u] "=&v"(tA_), [v] "=&v"(tB_), [w] "=&v"(tC_), [acc] "+v"(acc) \
		: [dD] "v"(dDc), [dA] "v"(dA), [dB] "v"(dB), [dC] "v"(dC), [c1] "s"(nb1), [c2] "s"(nb2), [c4] "s"(nb4), [c8] "s"(nb8))
#define MPA_NIB_FIRST "v_and_or_b32 %[acc], %[v], %[c8], %[u]"
#define MPA_NIB_NEXT  "v_and_or_b32 %[u], %[v], %[c8], %[u]\n\tv_lshl_or_b32 %[acc], %[acc], 4, %[u]"

// What the exact bookkeeping keeps per CALL (not per lane): in LDS, read and written only by the rare paths (task ends, steps of
// the length penalty, rows that vote), so that the sweep's registers hold nothing but the DP state.
struct ExtCall { int32_t Gs, maxi, maxsc, pk, pen, wit, nl, al; };     // wit: witness lane of the group, -1 none yet, -2 the call is done
#define EXT_DONE (-2)

// LITE: the packed sweep of the checkpointed traceback (dp_device.h): a global alignment (no best row, no x-drop: every row of
// every call is swept), extension bits and checkpoints out, H(nl-1, al-1) as the score.
// DUAL (G = 64): ONE call of 65..128 columns per wave, column c + 64 in the high half of lane c (slot 0 of the wave).
template<int G, bool LITE = false, bool DUAL = false>
__device__ __forceinline__ void ext_narrow(const ExtArgs &a, const ExtWave *wvp, const WavePos wp)
{
	static_assert(!DUAL || G == 64, "DUAL is the 128-column class: both halves of the 64 lanes");
	constexpr int NG = 64 / G;               // lane groups per wave
	constexpr int NSLOT = 2 * NG;            // calls per wave: two int16 halves per group
	char *lds_prof = wp.lds;                                           // [NSLOT][G columns][23] int16
	uint4 *lds_ring = (uint4*)(lds_prof + NSLOT * G * PROF_COL_STRIDE);  // [NG][24 rows] decoded records (ring_entry)
	ExtCall *lds_call = (ExtCall*)(lds_ring + 4 * EXT_RING_SLOTS);     // [NSLOT]
	int32_t *lds_pen = (int32_t*)(lds_call + 8);                       // x[MPA_PEN_MAX], val[MPA_PEN_MAX]

	const int lane = wp.lane, grp = lane / G, col = lane % G;
	const DpConst c = a.c;
	const uint32_t *recbase = a.rec + wvp->rec_base;
	const int32_t max_nl = wvp->max_nl;

	uint32_t roff[2], ioP;
	{
		int32_t io[2];
#pragma unroll
		for (int h = 0; h < 2; ++h) {
			const int32_t tid = wvp->task[DUAL ? 0 : h * NG + grp];
			ExtCall e;
			e.Gs = NEG16, e.maxi = -1, e.maxsc = INT32_MIN, e.pk = 0, e.pen = 0;
			if (tid >= 0) {
				const DTask *t = &a.tasks[tid];
				e.nl = t->nl, e.al = t->al, e.wit = -1, io[h] = t->io;
				roff[h] = (uint32_t)(t->rec_off - wvp->rec_base);
			} else e.nl = 0, e.al = 1, e.wit = EXT_DONE, io[h] = 0, roff[h] = 0;
			if (DUAL && h == 1) e.nl = 0, e.al = 1, e.wit = EXT_DONE;        // (the high halves belong to the call of slot 0)
			if (col == 0) lds_call[h * NG + grp] = e;
		}
		ioP = pack16(io[0], io[1]);
	}
	// ---- stage the profiles: global [22][pw] int16 -> LDS [col][aa]
	for (int s = 0; s < NSLOT; ++s) {
		const int t = wvp->task[DUAL ? 0 : s];
		const int c0 = DUAL ? s * G : 0;                                   // first profile column of the slot
		if (t < 0) {   // an empty slot scores zero everywhere (its half never reaches a result)
			for (int k = lane; k < G * PROF_COL_STRIDE / 2; k += 64) ((uint16_t*)(lds_prof + s * G * PROF_COL_STRIDE))[k] = 0;
			continue;
		}
		const DTask *tt = &a.tasks[t];
		const int16_t *src = a.prof + tt->prof_off;
		for (int k = lane; k < 22 * G; k += 64) {
			const int aa = k / G, cc = k - aa * G;
			*(int16_t*)(lds_prof + (s * G + cc) * PROF_COL_STRIDE + aa * PROF_AA_STRIDE) = src[aa * tt->pw + c0 + cc];
		}
	}
	for (int k = lane; k < MPA_PEN_MAX; k += 64) lds_pen[k] = a.pen.x[k], lds_pen[MPA_PEN_MAX + k] = a.pen.val[k];
	// ---- record ring: 24 decoded rows per lane group, row r in slot (r + 20) % 24, so that the record two rows ahead of row i
	// is in slot rs = (i - 2) % 24 and a block of 12 rows reads 12 consecutive slots (slot 24 repeats slot 0).  Rows [0, 16) go in now, rows [16, 28) wait in
	// `pf`; whenever rs reaches 0 or 12 the lanes col < 12 publish the 12 rows in `pf` into the OTHER half of the ring and request
	// the 12 rows after them (the records are padded behind the last call: dp_exec.hip).
	uint4 *ring = lds_ring + grp * EXT_RING_SLOTS;
	uint2 pf = make_uint2(0, 0);
	if (col < 16) ring[col < 4 ? col + 20 : col - 4] = ring_entry(recbase[roff[0] + col], recbase[roff[1] + col]);
	if (col < EXT_BLOCK_ROWS) pf = make_uint2(recbase[roff[0] + 16 + col], recbase[roff[1] + 16 + col]);
	wave_sync();

	// ---- per-lane constants
	const uint32_t jge = DUAL ? pack16(col * c.ge, (col + 64) * c.ge) : splat16(col * c.ge);
	const uint32_t gojge = DUAL ? pack16(c.go + col * c.ge, c.go + (col + 64) * c.ge) : splat16(c.go + col * c.ge);
	uint32_t stmp;                                                         // (DUAL: scalar scratch of the asm rows)
	const uint32_t goP = __builtin_amdgcn_readfirstlane(splat16(c.go)), fsP = __builtin_amdgcn_readfirstlane(splat16(c.fs));
	const uint32_t pbase0 = (uint32_t)(((0 * NG + grp) * G + col) * PROF_COL_STRIDE);
	const uint32_t pbase1 = (uint32_t)(((1 * NG + grp) * G + col) * PROF_COL_STRIDE);
	// (absolute LDS addresses of this lane's two profile columns, for the asm rows)
	const uint32_t pb0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds_prof + pbase0, pb1 = pb0 - pbase0 + pbase1;
	const uint32_t sel_lo = __builtin_amdgcn_readfirstlane(0x05040100u);   // v_perm_b32: {low half of src0, low half of src1}
	const uint32_t nb1 = __builtin_amdgcn_readfirstlane(0x00010001u), nb2 = __builtin_amdgcn_readfirstlane(0x00020002u),
	               nb4 = __builtin_amdgcn_readfirstlane(0x00040004u), nb8 = __builtin_amdgcn_readfirstlane(0x00080008u);
	// LITE: the three-row accumulator of extension-bit nibbles, the difference that holds row i's D bit, the D state the row before
	// row i's precomputation replaced (what a checkpoint wants), the wave's slice of the bit words (one dword per lane and 3 rows)
	uint32_t acc = 0, dDc = 0, Dold = NEGP;
	uint32_t *lite_p = LITE ? a.lite + wvp->lite_off + lane : nullptr;

	// ---- DP state: rings of three rows (compile-time indexed), A/B/C persist
	uint32_t Hr[3], Hs[3], Dr[3], dn[3], ac[3], M[3], A = NEGP, B = NEGP, C = NEGP, Hbest = NEGP;
#pragma unroll
	for (int k = 0; k < 3; ++k) Hr[k] = Hs[k] = Dr[k] = M[k] = NEGP;
	uint32_t k0 = NEGP, k1 = NEGP, k2 = NEGP, k3 = NEGP, ke = NEGP, kc = NEGP;   // fill registers of the asm rows' scan: only ever written by DPP moves
	// virtual column -1 as row 2 sees it (nasw-sse.c:256-258): H(-1,-1)=0, H(0,-1)=H(1,-1)=-fs
	if (col == 0) {
		if (DUAL) Hs[2] = pack16(0, NEG16), Hs[1] = Hs[0] = pack16(-c.fs, NEG16);   // (lane 0's high half is column 64)
		else Hs[2] = 0u, Hs[1] = splat16(-c.fs), Hs[0] = splat16(-c.fs);
	}
	// record r carries donor[r+1] and acceptor[r]
	{
		const uint4 q0 = ring[20], q1 = ring[21];
		dn[1] = q0.x, dn[0] = q1.x, ac[1] = q0.y, ac[0] = q1.y, dn[2] = ac[2] = 0;
	}

	// Vote thresholds, per lane and half: the exact bookkeeping (nasw-sse.c:423-433) runs for a row unless TvL <= H <= GvL in every
	// lane.  Start: always.
	uint32_t GvL = NEGP, TvL = 0x7fff7fffu;
	int32_t next_event = 2, limit = max_nl;

	// The rare paths work on the two calls of this lane's group, fetched from LDS (every lane of the group computes the same; lane 0
	// of the group writes back).  The lane index is made opaque in each of them: otherwise their lane-dependent constants (masks,
	// shuffle addresses) are hoisted out of the sweep and held in registers across it.
	auto load_calls = [&](ExtCall (&e)[2], int &lane_, int &grp_, int &col_) {
		lane_ = lane;
		asm volatile("" : "+v"(lane_));
		grp_ = lane_ / G, col_ = lane_ % G;
		e[0] = lds_call[grp_], e[1] = lds_call[NG + grp_];
	};
	auto store_calls = [&](const ExtCall (&e)[2], const int grp_, const int col_) {
		if (col_ == 0) lds_call[grp_] = e[0], lds_call[NG + grp_] = e[1];
		wave_sync();
	};

	// For a call that is not done, row i needs the bookkeeping iff some live lane has cand > Gs (a new best row) or the row
	// maximum m < Gs - xdrop, where cand = sat(sat(H + eb) - pen), eb = the end bonus in column al - 1, pen >= 0.
	//   * cand <= H + eb - pen unless the lower clamp raised it to -32768, which beats no Gs: cand > Gs implies H > Gs + pen - eb.
	//   * the witness lane w held the row maximum at the last bookkeeping; cand_w >= min(32767, H_w + eb) - pen, so
	//     H_w >= Gs - xdrop + pen - eb (and 32767 - pen >= Gs - xdrop) proves m >= cand_w >= Gs - xdrop.
	// Whenever a bound does not fit int16 (absurd ie_coef / x-drop: never with the reference's defaults), or a call has no witness
	// yet, the thresholds are set so that EVERY row votes: lo = 32767, hi = -32768 makes the clamp return 32767 in every lane,
	// and H stays below 32 000 in the packed kernels (dp_exec.hip refuses calls that could saturate).
	auto repack = [&](const ExtCall (&e)[2], const int col_) {
		int32_t gl[2], tl[2];
		bool force = false, any = false;
		if (LITE) {
			if (__ballot(e[0].wit != EXT_DONE || e[1].wit != EXT_DONE) == 0) limit = 0;
			return;
		}
#pragma unroll
		for (int h = 0; h < 2; ++h) {
			gl[h] = 32767, tl[h] = NEG16;
			const ExtCall &ec = e[DUAL ? 0 : h];                             // (DUAL: both halves are columns of the call in slot 0)
			const int cidx = DUAL ? col_ + 64 * h : col_;
			if (ec.wit == EXT_DONE) continue;
			any = true;
			const int32_t eb = cidx == ec.al - 1 ? c.end_bonus : 0;
			if (cidx < (ec.al + 7) / 8 * 8) {
				const int32_t g = ec.Gs + ec.pen - eb;
				if (g < NEG16) force = true;
				gl[h] = g < NEG16 ? NEG16 : g > 32767 ? 32767 : g;
			}
			if (ec.wit < 0 || 32767 - ec.pen < ec.Gs - c.xdrop) force = true;
			if (ec.wit == cidx) {
				const int32_t t = ec.Gs - c.xdrop + ec.pen - eb;
				if (t > 32767 || t < NEG16) force = true;
				tl[h] = t < NEG16 ? NEG16 : t > gl[h] ? gl[h] : t;
			}
		}
		GvL = pack16(gl[0], gl[1]), TvL = pack16(tl[0], tl[1]);
		if (__ballot(force) != 0) GvL = NEGP, TvL = 0x7fff7fffu;
		if (__ballot(any) == 0) limit = 0;                                  // every call of the wave has finished
	};

	// task ends and steps of the length penalty: rare, handled before the row is computed.  h_prev = the H row of row i - 1.
	auto events = [&](const int32_t i, const uint32_t h_prev) {
		ExtCall e[2];
		int lane_, grp_, col_;
		load_calls(e, lane_, grp_, col_);
		int32_t nx = INT32_MAX;
#pragma unroll
		for (int h = 0; h < 2; ++h) {
			if (e[h].wit != EXT_DONE && i >= e[h].nl) {
				e[h].wit = EXT_DONE;
				// LITE: row nl - 1 was the call's last: its score is H(nl-1, al-1) (nasw-sse.c:543); a call of fewer than three rows has none
				if (LITE && !DUAL && col_ == e[h].al - 1) a.score[wvp->task[h * NG + grp_]] = e[h].nl < 3 ? NEG16 : half16(h_prev, h);
				if (LITE && DUAL && col_ == (e[h].al - 1) % 64) a.score[wvp->task[0]] = e[h].nl < 3 ? NEG16 : half16(h_prev, (e[h].al - 1) / 64);
			}
			if (e[h].wit != EXT_DONE) {
				int64_t nxt = e[h].nl;
				if (!LITE) {
					const int64_t x = (int64_t)i - 3 * (int64_t)e[h].al;
					while (x >= (int64_t)lds_pen[e[h].pk + 1]) ++e[h].pk;
					e[h].pen = lds_pen[MPA_PEN_MAX + e[h].pk];
					nxt = (int64_t)lds_pen[e[h].pk + 1] + 3 * (int64_t)e[h].al;
					if (nxt > e[h].nl) nxt = e[h].nl;
				}
				nx = nx < (int32_t)nxt ? nx : (int32_t)nxt;
			}
		}
#pragma unroll
		for (int off = 1; off < 64; off <<= 1) { const int32_t o = __shfl_xor(nx, off); nx = nx < o ? nx : o; }
		next_event = __builtin_amdgcn_readfirstlane(nx);
		store_calls(e, grp_, col_);
		repack(e, col_);
	};

	// exact restatement of nasw-sse.c:423-433 for row i, whose final H row is h
	auto book = [&](const uint32_t h, const int32_t i) {
		ExtCall e[2];
		int lane_, grp_, col_;
		load_calls(e, lane_, grp_, col_);
		if (DUAL) {
			// one call in both halves: its row maximum is over the live columns of both, the witness the first column holding it
			ExtCall &ec = e[0];
			const uint32_t ebv = pack16(col_ == ec.al - 1 ? c.end_bonus : 0, col_ + 64 == ec.al - 1 ? c.end_bonus : 0);
			const uint32_t cand = p_subs(p_adds(h, ebv), splat16(ec.pen));
			const int32_t ncol_ = (ec.al + 7) / 8 * 8, cv0 = lo16(cand), cv1 = hi16(cand);
			const bool live0 = col_ < ncol_, live1 = col_ + 64 < ncol_;
			const int32_t m = group_max<G>(imax(live0 ? cv0 : NEG16, live1 ? cv1 : NEG16));
			if (ec.wit != EXT_DONE) {
				if (m > ec.Gs) ec.Gs = m, ec.maxi = i, ec.maxsc = m + ec.pen, Hbest = h;
				if (ec.Gs - m > c.xdrop) ec.wit = EXT_DONE;
				else {
					const uint64_t at0 = __ballot(live0 && cv0 == m), at1 = __ballot(live1 && cv1 == m);
					ec.wit = at0 ? __ffsll((unsigned long long)at0) - 1 : 64 + __ffsll((unsigned long long)at1) - 1;
				}
			}
			store_calls(e, grp_, col_);
			repack(e, col_);
			return;
		}
		const uint32_t ebv = pack16(col_ == e[0].al - 1 ? c.end_bonus : 0, col_ == e[1].al - 1 ? c.end_bonus : 0);
		const uint32_t cand = p_subs(p_adds(h, ebv), pack16(e[0].pen, e[1].pen));
#pragma unroll
		for (int hh = 0; hh < 2; ++hh) {
			const int32_t cv = half16(cand, hh);
			const bool live = col_ < (e[hh].al + 7) / 8 * 8;
			const int32_t m = group_max<G>(live ? cv : NEG16);
			if (e[hh].wit == EXT_DONE) continue;
			if (m > e[hh].Gs) {                                              // new best row
				e[hh].Gs = m, e[hh].maxi = i, e[hh].maxsc = m + e[hh].pen;
				Hbest = hh ? ((Hbest & 0xffffu) | (h & 0xffff0000u)) : ((Hbest & 0xffff0000u) | (h & 0xffffu));
			}
			if (e[hh].Gs - m > c.xdrop) e[hh].wit = EXT_DONE;                // x-drop
			else {                                                           // witness: the first lane holding the row maximum
				const uint64_t at = __ballot(live && cv == m) & group_mask<G>(lane_);
				e[hh].wit = (__ffsll((unsigned long long)at) - 1) % G;
			}
		}
		store_calls(e, grp_, col_);
		repack(e, col_);
	};

	int32_t i = 2, rs = 0;                                                 // rs = (i - 2) % 24: ring slot of the record of row i + 2
	bool have_hD = false;                                                  // hD (below) already holds row i's max(diagonal, D)
	uint32_t hD = NEGP;
	uint4 rcur = ring[22], rnext = ring[23];                               // records of rows i and i + 1
	uint32_t S = prof2s(pb0 + (rcur.w & 0xffff), pb1 + (rcur.w >> 16));   // profile scores of row i

	auto refill = [&]() {                                                  // rs is 0 or 12: rows [i + 14, i + 26) published, the next 12 requested
		int lane_ = lane;
		asm volatile("" : "+v"(lane_));
		const int col_ = lane_ % G;
		if (col_ < EXT_BLOCK_ROWS) {
			const uint4 e = ring_entry(pf.x, pf.y);
			uint4 *rg = lds_ring + lane_ / G * EXT_RING_SLOTS;
			rg[(rs ^ 12) + col_] = e;
			if (rs == 12 && col_ == 0) rg[EXT_RING_ROWS] = e;                // slot 0 again behind slot 23
			pf = make_uint2(recbase[roff[0] + (uint32_t)i + 26 + col_], recbase[roff[1] + (uint32_t)i + 26 + col_]);
		}
	};

	// LITE: the DP state at the top of row i = 2 + k * MPA_TB_BLOCK (a row with K = 0: slots 0, 1, 2 hold rows i-1, i-2, i-3)
	auto checkpoint = [&](const int32_t i, const uint32_t d3) {
		uint32_t *p = a.ckpt + wvp->ck_off + (int64_t)((i - 2) / MPA_TB_BLOCK - 1) * 9 * 64 + lane;
		p[0] = Hr[0], p[64] = Hr[1], p[128] = Hr[2], p[192] = Dr[0], p[256] = Dr[1], p[320] = d3, p[384] = A, p[448] = B, p[512] = C;
	};

	// ---- the generic row: plain C++, any row.  (The lane index is opaque here as well: the row runs a few dozen times per call, and
	// what it derives from the lane must not be kept in registers across the asm rows.)
	auto row = [&](auto kc_) {
		constexpr int K = decltype(kc_)::value;
		constexpr int R1 = (3 - K) % 3, R2 = (4 - K) % 3, R3 = (5 - K) % 3;   // ring slots of rows i-1, i-2, i-3
		int lane_ = lane;
		asm volatile("" : "+v"(lane_));
		const int col_ = lane_ % G;
		if (rs == 0 || rs == 12) refill();
		if (LITE && K == 0 && i > 2 && (uint32_t)(i - 2) % MPA_TB_BLOCK == 0) checkpoint(i, have_hD ? Dold : Dr[R3]);
		// next row's profile scores and the record of row i+2: issued first, consumed one row later
		const uint32_t Snext = prof2s(pb0 + (rnext.w & 0xffff), pb1 + (rnext.w >> 16));
		const uint4 rnn = ring[rs];
		dn[R3] = rcur.x, ac[R3] = rcur.y;                                  // donor[i+1], acceptor[i]
		// now dn[R3]=donor[i+1], dn[R1]=donor[i], dn[R2]=donor[i-1]; ac[R3]=acceptor[i], ac[R1]=acceptor[i-1], ac[R2]=acceptor[i-2]
		uint32_t h, t, u, dA = 0, dB = 0, dC = 0;
		if (have_hD) h = hD;
		else {
			h = p_adds(Hs[R3], S);                                         // H(i-3,j-1) + s(i,j)
			u = p_subs(Hr[R3], goP);                                       // D
			if (LITE) dDc = p_subs(u, Dr[R3]);
			t = p_max(u, Dr[R3]);
			t = p_subs(t, rcur.z); Dr[R3] = t; h = p_max(h, t);
		}
		u = p_subs(Hr[R1], ioP); t = p_subs(u, dn[R2]);                    // A: phase-0 intron, donor[i-1], acceptor[i]
		if (LITE) dA = p_subs(t, A);
		t = p_max(t, A); A = t; h = p_max(h, p_subs(t, ac[R3]));
		u = p_subs(Hs[R1], ioP); t = p_subs(u, dn[R1]);                    // B: phase-1 intron, donor[i], acceptor[i-2]
		if (LITE) dB = p_subs(t, B);
		t = p_max(t, B); B = t; h = p_max(h, p_subs(t, ac[R2]));
		t = p_subs(u, dn[R3]);                                             // C: phase-2 intron, donor[i+1], acceptor[i-1]
		if (LITE) dC = p_subs(t, C);
		t = p_max(t, C); C = t; h = p_max(h, p_subs(t, ac[R1]));
		if (LITE) {                                                        // the row's extension bits into the three-row word
			const uint32_t nib = (dDc >> 15 & 0x00010001u) | (dA >> 14 & 0x00020002u) | (dB >> 13 & 0x00040004u) | (dC >> 12 & 0x00080008u);
			acc = K == 0 ? nib : (acc << 4 | nib);
			if (K == 2) *lite_p = acc, lite_p += 64;
		}
		t = p_max(p_max(Hr[R1], Hr[R2]), p_max(Hs[R1], Hs[R2]));          // four frameshift moves
		h = p_max(h, p_subs(t, fsP));
		// horizontal gap chain as a prefix max of h + j*ge
		const uint32_t y = scan_max_pk<G>(p_adds(h, jge));
		uint32_t ex = shift1<G>(y, NEGP, lane_);
		if (DUAL) ex = p_max(ex, (uint32_t)__builtin_amdgcn_readlane((int)y, 63) << 16 | 0x8000u);   // the low halves' total into every high half
		h = p_max(h, p_subs(ex, gojge));
		Hr[R3] = h, Hs[R3] = shift1<G>(h, NEGP, lane_);
		if (DUAL) { const uint32_t h63 = (uint32_t)__builtin_amdgcn_readlane((int)h, 63); if (lane_ == 0) Hs[R3] = (Hs[R3] & 0xffffu) | h63 << 16; }
		if (!LITE) { if (__ballot(p_max(p_min(h, GvL), TvL) != h) != 0) book(h, i); }
		if (i == 2 && col_ == 0) Hs[R1] = NEGP, Hs[R2] = NEGP;              // column -1 is -inf from row 3 on
		S = Snext;
		rcur = rnext, rnext = rnn;
		++i, rs = rs == EXT_RING_ROWS - 1 ? 0 : rs + 1, have_hD = false;
	};

	// ---- the asm rows: blocks of 12 rows that contain no event, entered at rs = 0 or 12 (where i - 2 is a multiple of 3)
	auto fast_blocks = [&]() {
		uint32_t tA_, tB_, tC_, h, x, hDn, a0, a1, vt, dA, dB, dC, drn, dDn;
		v2s sraw0, sraw1;                                                  // (.x = a profile score; .y is never set: v_perm_b32 takes the low halves)
		M[0] = p_max(Hr[0], Hs[0]), M[1] = p_max(Hr[1], Hs[1]);
		if (!have_hD) {                                                    // row i's diagonal term and D state (slot R3 = 2)
			const uint32_t u = p_subs(Hr[2], goP), t = p_max(u, Dr[2]);
			if (LITE) dDc = p_subs(u, Dr[2]), Dold = Dr[2];
			Dr[2] = p_subs(t, rcur.z);
			hD = p_max(p_adds(Hs[2], S), Dr[2]);
		}
		sraw0.x = *(lds_s16p)(uintptr_t)(pb0 + (rnext.w & 0xffff)), sraw1.x = *(lds_s16p)(uintptr_t)(pb1 + (rnext.w >> 16));
		uint4 rnn = ring[rs];                                              // record of row i + 2; from here on fetched a row before it is needed
		do {
			// (every outstanding global load is 12 rows old here -- or a register reload at the entry: said once, outside the rows, so
			// that no row waits for the loads refill() is about to issue)
			__builtin_amdgcn_s_waitcnt(0x0f70);                            // vmcnt(0)
			refill();
			if (LITE && (uint32_t)(i - 2) % MPA_TB_BLOCK == 0) checkpoint(i, Dold);
			const uint4 *rb = ring + rs;
#pragma unroll 1
			for (int g = 0; g < EXT_BLOCK_ROWS / 3; ++g, rb += 3) {
#define MPA_FAST_ROW(K, R1, R2, R3) { \
				if constexpr (!LITE) { \
					MPA_ROW_HEAD(Hr[R1], Hs[R1], dn[R1], dn[R2], rcur.x, ac[R1], ac[R2], rcur.y, M[R1], M[R2]); \
					dn[R3] = rcur.x, ac[R3] = rcur.y; \
					if constexpr (G == 16) MPA_ROW_TAIL(MPA_EX16, MPA_HS16, Hr[R2], Hs[R2], Dr[R2], Hs[R3], M[R3]); \
					else if constexpr (G == 32) MPA_ROW_TAIL(MPA_EX32, MPA_HS32, Hr[R2], Hs[R2], Dr[R2], Hs[R3], M[R3]); \
					else if constexpr (DUAL) MPA_ROW_TAIL_D(Hr[R2], Hs[R2], Dr[R2], Hs[R3], M[R3]); \
					else MPA_ROW_TAIL(MPA_EX64, MPA_HS64, Hr[R2], Hs[R2], Dr[R2], Hs[R3], M[R3]); \
				} else { \
					MPA_ROW_HEAD_L(Hr[R1], Hs[R1], dn[R1], dn[R2], rcur.x, ac[R1], ac[R2], rcur.y, M[R1], M[R2]); \
					dn[R3] = rcur.x, ac[R3] = rcur.y; \
					if constexpr (G == 16) MPA_ROW_TAIL_L(MPA_EX16_L, MPA_HS16, Hr[R2], Hs[R2], Dr[R2], Hs[R3], M[R3]); \
					else if constexpr (G == 32) MPA_ROW_TAIL_L(MPA_EX32_L, MPA_HS32, Hr[R2], Hs[R2], Dr[R2], Hs[R3], M[R3]); \
					else if constexpr (DUAL) MPA_ROW_TAIL_LD(Hr[R2], Hs[R2], Dr[R2], Hs[R3], M[R3]); \
					else MPA_ROW_TAIL_L(MPA_EX64_L, MPA_HS64, Hr[R2], Hs[R2], Dr[R2], Hs[R3], M[R3]); \
					if constexpr (K == 0) MPA_NIBBLE(MPA_NIB_FIRST); else MPA_NIBBLE(MPA_NIB_NEXT); \
					Dold = Dr[R2], Dr[R2] = drn, dDc = dDn; \
					if constexpr (K == 2) *lite_p = acc, lite_p += 64; \
				} \
				sraw0.x = *(lds_s16p)(uintptr_t)a0, sraw1.x = *(lds_s16p)(uintptr_t)a1; \
				rcur = rnext, rnext = rnn, rnn = rb[K + 1]; \
				Hr[R3] = h, hD = hDn; \
				if constexpr (!LITE) { if (__ballot(vt != h) != 0) book(h, i); } \
				++i; }
				MPA_FAST_ROW(0, 0, 1, 2)
				MPA_FAST_ROW(1, 2, 0, 1)
				MPA_FAST_ROW(2, 1, 2, 0)
#undef MPA_FAST_ROW
			}
			rs ^= 12;
		} while (i + EXT_BLOCK_ROWS <= (limit < next_event ? limit : next_event));
		have_hD = true;
	};

	// (slot of row i - 1 before a row with K = 0, 1, 2: 0, 2, 1)
	while (i < limit) {
		if (i >= next_event) { events(i, Hr[0]); if (i >= limit) break; }
		if ((rs == 0 || rs == 12) && i >= 14 && i + EXT_BLOCK_ROWS <= (limit < next_event ? limit : next_event)) { fast_blocks(); continue; }
		row(std::integral_constant<int, 0>());
		if (i >= limit) break;
		if (i >= next_event) { events(i, Hr[2]); if (i >= limit) break; }
		row(std::integral_constant<int, 1>());
		if (i >= limit) break;
		if (i >= next_event) { events(i, Hr[1]); if (i >= limit) break; }
		row(std::integral_constant<int, 2>());
	}
	if (LITE) {
		// the last word may be partial (the sweep ended inside a group of three rows), and calls that end with the wave's last row
		// have not met their task-end event yet
		const int32_t part = (i - 2) % 3;
		if (part != 0) *lite_p = acc << (4 * (3 - part));
		if (i >= next_event && limit != 0) events(i, part == 0 ? Hr[0] : part == 1 ? Hr[2] : Hr[1]);
		return;
	}

	// ---- results (nasw-sse.c:435-443): first column of the best row that attains the best score
	if (DUAL) {
		const int32_t tid = wvp->task[0];
		if (tid < 0) return;
		const ExtCall e = lds_call[0];
		int32_t first = INT32_MAX;
		if (e.maxi >= 0) {
			int32_t v0 = lo16(Hbest), v1 = hi16(Hbest);
			if (col == e.al - 1) v0 += c.end_bonus;
			if (col + 64 == e.al - 1) v1 += c.end_bonus;
			const uint64_t m0 = __ballot(col < e.al && v0 == e.maxsc), m1 = __ballot(col + 64 < e.al && v1 == e.maxsc);
			if (m0) first = __ffsll((unsigned long long)m0) - 1;
			else if (m1) first = 64 + __ffsll((unsigned long long)m1) - 1;
		}
		if (col == 0) {
			ExtOut o;
			o.nt_len = e.maxi + 1;
			o.aa_len = e.maxi < 0 ? 0 : (first == INT32_MAX ? e.al + 1 : first + 1);
			o.score = e.maxsc;
			o.flags = (e.maxi >= 0 && first == INT32_MAX) ? 1 : 0;
			a.out[tid] = o;
		}
		return;
	}
#pragma unroll
	for (int h = 0; h < 2; ++h) {
		const int32_t tid = wvp->task[h * NG + grp];
		if (tid < 0) continue;
		const ExtCall e = lds_call[h * NG + grp];
		int32_t first = INT32_MAX;
		if (e.maxi >= 0) {
			int32_t v = half16(Hbest, h);
			if (col == e.al - 1) v += c.end_bonus;
			const uint64_t m = __ballot(col < e.al && v == e.maxsc) & group_mask<G>(lane);
			if (m != 0) first = (__ffsll((unsigned long long)m) - 1) % G;
		}
		if (col == 0) {
			ExtOut o;
			o.nt_len = e.maxi + 1;
			o.aa_len = e.maxi < 0 ? 0 : (first == INT32_MAX ? e.al + 1 : first + 1);
			o.score = e.maxsc;
			o.flags = (e.maxi >= 0 && first == INT32_MAX) ? 1 : 0;        // the reference assert()s here
			a.out[tid] = o;
		}
	}
}

// One launch for the three narrow shapes (16 / 32 / 64 lanes per call): wave b of the launch belongs to the class whose
// range of wave descriptors contains it.  Fewer, larger launches matter because every concurrently running kernel
// occupies a hardware queue and a stream of batches keeps several rounds in flight (dp_exec.hip).
struct NarrowMap { int32_t first[4], cnt[4]; };



// ------------------------------------------------------------------------------------------------
// K1-wide: extension calls wider than 64 columns.  One workgroup of NW waves per pair of calls (the two
// int16 halves); wave w owns columns [64w, 64w+64) and runs ONE ROW BEHIND wave w-1 (a wavefront over
// waves), so the horizontal carry and the left-neighbour column it needs were produced in the previous step
// and arrive through LDS with a single barrier per step.  The critical path is nl + NW steps instead of
// nl * NW block-rows of a single wave.
//
// Bookkeeping differs from k_ext: every (row, block) posts max(H + end bonus) together with the first
// column attaining it into a per-row key (atomicMax), and the reference's sequential best-row / x-drop
// logic (nasw-sse.c:423-433) is replayed over the keys afterwards -- exact, because the x-drop only ever
// stops the sweep, it never changes a row that was computed.
// ------------------------------------------------------------------------------------------------
struct ExtWideArgs {
	const DTask *tasks;
	const ExtWave *waves;     // task[0], task[1] = the two halves
	const uint32_t *rec;
	const int16_t *prof;
	ExtOut *out;
	uint32_t *rowkey;         // [2][max_nl] per group, zero-initialised; offset = group * 2 * key_stride
	int64_t key_stride;
	DpConst c;
	PenTable pen;
	// split classes only (k_ext_wide_split): column-block boundaries travel between workgroups through HBM
	unsigned long long *xg;   // [boundary][key_stride rows][2] granules {payload, row + 1}, zero-initialised
	int32_t *done;            // [group] workgroups of the group that have finished, zero-initialised
	int32_t *ticket;          // [group] workgroups of the group that have STARTED, zero-initialised: a workgroup's column block is its ticket
	int32_t *err;             // set to 1 if a hand-off never arrived (bounded spin)
};

// where a workgroup of the split kernel sits: group = pair of calls, blk = which quarter (or half) of its columns
struct SplitPos { int32_t group, blk, n_blk, xg_first; };

// One 8-byte granule {payload, tag} per value and row, written by ONE write-through store and polled with loads that bypass
// the reader's L1 (MI355X_MICROARCH.md, hand-off by data-tagged granules): no fence, no separate flag.
__device__ __forceinline__ void granule_put(unsigned long long *p, uint32_t payload, uint32_t tag)
{
	__hip_atomic_store(p, (unsigned long long)tag << 32 | payload, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long granule_get(const unsigned long long *p)
{
	return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// NWB = waves of this workgroup.  SPLIT: the pair of calls spans pos.n_blk workgroups of NWB waves each; workgroup pos.blk
// owns the columns [256 blk, 256 blk + 256) and receives the carry of the gap scan and the H of the column to its left, row
// by row, from workgroup blk - 1 through HBM (it runs about fifty rows behind it), so that a 1024-column call occupies four
// CUs with one wave per SIMD instead of one CU with four (782 -> ~370 ns per row: the tail of a DP round at genome scale).
//
// Round 6: the rows from row 14 on are the hand-scheduled rows of ext_narrow (MPA_ROW_HEAD and a tail without the vote): these
// calls sweep windows of max_intron / 2 rows (align.c:259-261) that the x-drop practically never cuts short, so a round lasts as
// long as its longest wide call -- 98 600 rows at 425 ns in the compiled rows of rounds 2-5 (198 instructions per row), which is
// what the round's other 8 000 units were waiting for.  What a wave takes from its left neighbour -- the carry of the gap scan
// and H of the column before its first -- enters through LANE 0 only: folded into lane 0's scan input (the inclusive scan then
// carries it to every lane and lane 63 holds the carry for the wave to the right), and preset in lane 0 of the two registers the
// wave_shr:1 moves leave untouched there.  Every lane reads the 8-byte exchange slot {carry, H} of its row with one ds_read_b64
// (lane 0 the neighbour's slot -- or the incoming ring of a split workgroup's first wave --, the others a slot of -inf) and writes
// {scan, H} with one ds_write_b64 (lane 63 into the wave's slot, the others into a dump area): no exec masking in the rows.
#define EXT_WIDE_RING 48                       /* rows of decoded records (ring_entry) in LDS; slot 48 repeats slot 0 */
#define EXT_WIDE_XIN 96                        /* rows of incoming boundary values kept in LDS (two fetches of 48) */
#define EXT_WIDE_KROWS 96                      /* rows of the key ring: 64 per flush + the lead of the first wave over the last; a multiple of 3 like the other two */
#define EXT_WIDE_LDS(NW) ((size_t)(NW) * 2 * 64 * PROF_COL_STRIDE + (EXT_WIDE_RING + 1) * 16 + 256 + 32 + (64 * 8 + 32) + EXT_WIDE_XIN * 8 + EXT_WIDE_KROWS * 2 * (NW) * 4)   /* bytes per group; a multiple of 16 */
typedef uint32_t v2u __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) v2u *lds_u2p;
#define MPA_ROW_TAIL_W(Hr2, Hs2, Dr2, Hs3, M3, CX, CH) \
	asm volatile( \
		"v_pk_max_i16 %[x], %[x], %[cx]\n\t" \
		MPA_SCAN4 \
		"v_mov_b32 %[t2], 0x80008000\n\t" \
		"v_mov_b32 %[ke], %[cx]\n\t" \
		"v_mov_b32_dpp %[t2], %[x] row_bcast:15 row_mask:0xa bank_mask:0xf\n\t" \
		"v_pk_max_i16 %[x], %[x], %[t2]\n\t" \
		"v_mov_b32 %[t2], 0x80008000\n\t" \
		"v_mov_b32 %[hs3], %[ch]\n\t" \
		"v_mov_b32_dpp %[t2], %[x] row_bcast:31 row_mask:0xc bank_mask:0xf\n\t" \
		"v_pk_max_i16 %[x], %[x], %[t2]\n\t" \
		"s_nop 1\n\t" \
		"v_mov_b32_dpp %[ke], %[x] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
		"v_pk_sub_i16 %[t2], %[ke], %[gojge] clamp\n\t" \
		"v_pk_max_i16 %[h], %[h], %[t2]\n\t" \
		"s_nop 1\n\t" \
		"v_mov_b32_dpp %[hs3], %[h] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
		"v_pk_max_i16 %[m3], %[h], %[hs3]" \
		: [x] "+v"(x), [h] "+v"(h), [k0] "+v"(k0), [k1] "+v"(k1), [k2] "+v"(k2), [k3] "+v"(k3), [ke] "+v"(ke), [dr2] "+v"(Dr2), [hs3] "+v"(Hs3), \
		  [t2] "=&v"(tA_), [hDn] "=&v"(hDn), [a0] "=&v"(a0), [a1] "=&v"(a1), [m3] "=&v"(M3) \
		: [hr2] "v"(Hr2), [gei] "v"(rnext.z), [s0] "v"(sraw0), [s1] "v"(sraw1), [hs2] "v"(Hs2), [pb0] "v"(pb0), [pb1] "v"(pb1), [aw] "v"(rnn.w), \
		  [gojge] "v"(gojge), [cx] "v"(CX), [ch] "v"(CH), [go] "s"(goP), [sel] "s"(sel_lo))
template<int NWB, bool SPLIT>
__device__ __forceinline__ void ext_wide_body(const ExtWideArgs &a, const int group_idx, const SplitPos pos, const WavePos wp)
{
	constexpr int NW = NWB;
	// per wave: profile of its 64 columns for both halves [2][64 columns][23] int16; then the record ring, the exchange slots
	// [2 step parities][NW waves][3 rows] of {scan, H of the wave's last column}, 24 bytes of -inf, the dump area, the incoming
	// boundary ring and the key ring
	char *lds_prof = wp.lds;
	constexpr int RING = EXT_WIDE_RING, XIN = EXT_WIDE_XIN;
	uint4 *ring = (uint4*)(lds_prof + NW * 2 * 64 * PROF_COL_STRIDE);  // [RING + 1] decoded records of both halves
	uint2 *xch = (uint2*)(ring + RING + 1);                          // [2][16] slots: parity * 16 + 3 w + k
	uint2 *xneg = xch + 32;                                          // [4] {-inf, -inf}
	uint2 *xdump = xneg + 4;                                         // [64 + 4]
	uint2 *xin = xdump + 68;                                         // [XIN] {carry, H} from the workgroup to the left (SPLIT only)
	uint32_t *kring = (uint32_t*)(xin + XIN);                        // [NW][96 rows][2 halves] per-row, per-block keys (see keys()): row r at (r - 2) % 96
	const int lane = wp.lane, w = wp.w;                              // w: wave index inside the group, scalar
	// SPLIT: which column block this workgroup sweeps is decided when it STARTS -- the k-th workgroup of the group to start takes
	// block k -- not by its position in the grid: the workgroup it will wait for (block k - 1) is then certain to be running
	// already, whatever order the dispatcher hands workgroups to the XCDs in and whatever else competes for the CUs, so the
	// hand-off cannot wait for a workgroup that has no slot yet.
	int blk = 0;
	if (SPLIT) {
		int32_t *slot = (int32_t*)wp.lds;
		if (wp.tg == 0) *slot = atomicAdd(&a.ticket[pos.group], 1);
		__syncthreads();
		blk = __builtin_amdgcn_readfirstlane(*slot);
		__syncthreads();                                           // (the slot is profile space from here on)
	}
	const int gw = SPLIT ? blk * NWB + w : w;                    // wave index inside the call: owns columns [64 gw, 64 gw + 64)
	const bool has_left = SPLIT && blk > 0, has_right = SPLIT && blk + 1 < pos.n_blk;
	const ExtWave *wvp = &a.waves[group_idx];
	const DpConst c = a.c;
	const uint32_t *recbase = a.rec + wvp->rec_base;
	uint32_t *rowkey = a.rowkey + (int64_t)group_idx * 2 * a.key_stride;
	unsigned long long *xg_out = SPLIT ? a.xg + ((int64_t)(pos.xg_first + blk) * a.key_stride) * 2 : nullptr;
	const unsigned long long *xg_in = SPLIT ? a.xg + ((int64_t)(pos.xg_first + blk - 1) * a.key_stride) * 2 : nullptr;

	int32_t tid[2], nl[2], al[2], ncol[2];
	uint32_t roff[2];
#pragma unroll
	for (int h = 0; h < 2; ++h) {
		tid[h] = wvp->task[h];
		if (tid[h] >= 0) {
			const DTask *t = &a.tasks[tid[h]];
			nl[h] = t->nl, al[h] = t->al, ncol[h] = t->ncol, roff[h] = (uint32_t)(t->rec_off - wvp->rec_base);
		} else nl[h] = 0, al[h] = 1, ncol[h] = 0, roff[h] = 0;
	}
	// profile columns of this wave: global [22][pw] int16 -> LDS [column][amino acid]
	for (int h = 0; h < 2; ++h) {
		char *dst = lds_prof + (size_t)(w * 2 + h) * 64 * PROF_COL_STRIDE;
		if (tid[h] < 0) {   // (an empty half scores zero everywhere; its keys are never read)
			for (int k = lane; k < 64 * PROF_COL_STRIDE / 2; k += 64) ((uint16_t*)dst)[k] = 0;
			continue;
		}
		const DTask *t = &a.tasks[tid[h]];
		const int16_t *src = a.prof + t->prof_off;
		for (int k = lane; k < 22 * 64; k += 64) {
			const int aidx = k >> 6, cc = k & 63, gcc = gw * 64 + cc;
			*(int16_t*)(dst + cc * PROF_COL_STRIDE + aidx * PROF_AA_STRIDE) = gcc < t->pw ? src[aidx * t->pw + gcc] : (int16_t)NEG16;
		}
	}
	if (wp.tg < 36) xch[wp.tg] = make_uint2(NEGP, NEGP);           // the exchange slots and, behind them, the slots of -inf
	// record ring, filled by the leading wave: row r in slot (r + 44) % 48, so that the record two rows ahead of row i is in slot
	// (i - 2) % 48; rows [0, 16) now, rows [16, 28) wait in `pf`; at every twelfth row the lanes < 12 publish `pf` and request the
	// twelve rows after it (ext_narrow's scheme; the last wave runs 3 (NW - 1) rows behind the first)
	const bool loader = w == 0 && lane < 12;
	uint2 pf = make_uint2(0, 0);
	if (w == 0 && lane < 16) {
		const uint4 e = ring_entry(recbase[roff[0] + lane], recbase[roff[1] + lane]);
		ring[lane < 4 ? lane + 44 : lane - 4] = e;
		if (lane == 4) ring[RING] = e;
	}
	if (loader) pf = make_uint2(recbase[roff[0] + 16 + lane], recbase[roff[1] + 16 + lane]);
	__syncthreads();

	const int gc = gw * 64 + lane;
	const uint32_t jge = splat16(gc * c.ge), gojge = splat16(c.go + gc * c.ge);
	const uint32_t ebv = pack16(gc == al[0] - 1 ? c.end_bonus : 0, gc == al[1] - 1 ? c.end_bonus : 0);
	// live halves keep their value, dead halves read as -32768 in the row maximum
	const uint32_t keepP = (gc < ncol[0] ? 0xffffu : 0u) | (gc < ncol[1] ? 0xffff0000u : 0u);
	const uint32_t flipP = ((gc < ncol[0] ? 0u : 0x8000u) | (gc < ncol[1] ? 0u : 0x80000000u)) ^ 0x80008000u;   // (key order: value + 32768, dead halves 0)
	const uint32_t goP = __builtin_amdgcn_readfirstlane(splat16(c.go)), fsP = __builtin_amdgcn_readfirstlane(splat16(c.fs));
	const uint32_t ioP = pack16(tid[0] >= 0 ? a.tasks[tid[0]].io : 0, tid[1] >= 0 ? a.tasks[tid[1]].io : 0);
	const uint32_t sel_lo = __builtin_amdgcn_readfirstlane(0x05040100u);
	// absolute LDS addresses: this lane's two profile columns; the exchange slot it reads / writes in a step of parity 0
	const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)wp.lds;
	const uint32_t pb0 = lds0 + (uint32_t)(((w * 2 + 0) * 64 + lane) * PROF_COL_STRIDE), pb1 = pb0 + 64 * PROF_COL_STRIDE;
	const uint32_t xch0 = lds0 + (uint32_t)((char*)xch - wp.lds), xneg0 = lds0 + (uint32_t)((char*)xneg - wp.lds),
	               xdump0 = lds0 + (uint32_t)((char*)xdump - wp.lds), xin0 = lds0 + (uint32_t)((char*)xin - wp.lds);
	const uint32_t xwr_even = lane == 63 ? xch0 + (uint32_t)(3 * w) * 8 : xdump0 + (uint32_t)lane * 8;
	const uint32_t xwr_flip = lane == 63 ? 128u : 0u;              // (the other parity's slots: + 16 slots)
	const bool takes_left = lane == 0 && (w > 0 || has_left);

	uint32_t Hr[3], Hs[3], Dr[3], dn[3], ac[3], M[3], A = NEGP, B = NEGP, C = NEGP;
#pragma unroll
	for (int k = 0; k < 3; ++k) Hr[k] = Hs[k] = Dr[k] = M[k] = NEGP;
	uint32_t k0 = NEGP, k1 = NEGP, k2 = NEGP, k3 = NEGP, ke = NEGP;    // fill registers of the asm rows' scan: only ever written by DPP moves
	if (gw == 0 && lane == 0) Hs[2] = 0u, Hs[1] = splat16(-c.fs), Hs[0] = splat16(-c.fs);
	{
		const uint4 q0 = ring[44], q1 = ring[45];
		dn[1] = q0.x, dn[0] = q1.x, ac[1] = q0.y, ac[0] = q1.y, dn[2] = ac[2] = 0;
	}
	const int32_t max_nl = wvp->max_nl;
	uint4 rcur = ring[46], rnext = ring[47];                           // records of rows i and i + 1
	uint32_t S = prof2s(pb0 + (rcur.w & 0xffff), pb1 + (rcur.w >> 16));   // profile scores of row i (generic rows)
	bool have_hD = false;                                              // hD already holds row i's max(diagonal, D) (behind the asm rows)
	uint32_t hD = NEGP;

	uint32_t hxv[3];
	// Per-row key of this block: the best (H + end bonus) and its column, ties to the smallest column, per half, as
	// (value + 32768) << 6 | (63 - lane) so that one unsigned wave maximum yields both.  The reductions of the three rows
	// of a step are independent dependency chains and are issued together; lane 63 drops the results into an LDS ring
	// [row][block], from which the last wave -- the last to finish a row -- assembles 64 rows at a time (flush_keys) the
	// per-row keys the replay below consumes.  (Global atomics per row and wave, or assembling the key with scalar code in
	// the row loop, each cost a quarter of this kernel's time.)
	const uint32_t lane_tag = 63u - (uint32_t)lane;
	auto keys = [&](const int r96, int n_rows) {                       // r96 = (i - 2) % 96, a multiple of 3: the step's rows are consecutive ring rows
		uint32_t k0_[3], k1_[3];
#pragma unroll
		for (int k = 0; k < 3; ++k) {                                      // (hxv holds the values with their sign bits flipped already)
			k0_[k] = ((hxv[k] & 0xffffu) << 6) | lane_tag;
			k1_[k] = ((hxv[k] >> 16) << 6) | lane_tag;
		}
		reduce_max_u32_x6(k0_[0], k1_[0], k0_[1], k1_[1], k0_[2], k1_[2]);
		if (lane == 63) {
			uint32_t *slot = kring + (w * EXT_WIDE_KROWS + r96) * 2;
#pragma unroll
			for (int k = 0; k < 3; ++k) {
				if (k >= n_rows) break;
				slot[2 * k] = k0_[k], slot[2 * k + 1] = k1_[k];
			}
		}
	};
	// SPLIT, leading wave of a workgroup that has a left neighbour: the boundary values of rows [i0, i0 + 48) into the LDS
	// ring, one row per lane, polling until the neighbour has written them (it is a few dozen rows ahead; the spin is bounded)
	auto fetch_left = [&](int32_t i0) {
		const int32_t r = i0 + lane;
		const bool want = lane < 48 && r < max_nl;
		unsigned long long gy = 0, gh = 0;
		for (int spin = 0;; ++spin) {
			if (want) gy = granule_get(xg_in + 2 * (int64_t)r), gh = granule_get(xg_in + 2 * (int64_t)r + 1);
			const bool ok = !want || ((uint32_t)(gy >> 32) == (uint32_t)r + 1 && (uint32_t)(gh >> 32) == (uint32_t)r + 1);
			if (__ballot(!ok) == 0) break;
			if (spin > (1 << 22)) { if (lane == 0) *a.err = 1; break; }    // seconds: the producer is gone; give up loudly
			__builtin_amdgcn_s_sleep(4);
		}
		if (want) xin[(r - 2) % XIN] = make_uint2((uint32_t)gy, (uint32_t)gh);   // (a step's three rows never wrap: 96 = 32 steps)
		wave_sync();
	};
	// the leading wave, every twelfth row (rs = (i - 2) % 48 is 0, 12, 24 or 36): rows [i + 14, i + 26) published, the next 12 requested
	auto refill = [&](const int32_t i, const int rs) {
		if (loader) {
			const uint4 e = ring_entry(pf.x, pf.y);
			const int at = (rs + 12) % RING + lane;
			ring[at] = e;
			if (at == 0) ring[RING] = e;                                       // slot 0 again behind slot 47
			pf = make_uint2(recbase[roff[0] + (uint32_t)i + 26 + lane], recbase[roff[1] + (uint32_t)i + 26 + lane]);
		}
	};
	// what the wave to the left produced for the three rows of this step -- in the previous step (same workgroup, through LDS) or
	// some dozens of rows ago (the workgroup to the left, through HBM and the xin ring): lane 0 takes it, the others take -inf
	v2u cv[3];
	uint32_t xwr = 0;
	auto step_begin = [&](const int r96, const int par) {
		uint32_t at = xneg0;
		if (takes_left) at = w > 0 ? xch0 + (uint32_t)(((par ^ 1) * 16 + 3 * (w - 1)) * 8) : xin0 + (uint32_t)r96 * 8;
#pragma unroll
		for (int k = 0; k < 3; ++k) cv[k] = *(lds_u2p)(uintptr_t)(at + 8 * k);
		xwr = xwr_even + (par ? xwr_flip : 0u);
	};
	// the workgroup to the right reads the step's boundary values from HBM: lanes 0..5 pass on what lane 63 has just put into LDS
	auto step_end = [&](const int32_t i, const int par, const int n_rows) {
		if (has_right && w == NW - 1) {
			wave_sync();
			if (lane < 2 * n_rows) {
				const uint32_t v = ((const uint32_t*)(xch + par * 16 + 3 * w))[lane];
				granule_put(xg_out + 2 * (int64_t)i + lane, v, (uint32_t)(i + (lane >> 1)) + 1);
			}
		}
	};
	// the generic row (plain C++): the first twelve rows and the last one or two
	auto row = [&](auto kc, const int32_t i, const int rs) {
		constexpr int K = decltype(kc)::value;
		constexpr int R1 = (3 - K) % 3, R2 = (4 - K) % 3, R3 = (5 - K) % 3;
		const uint32_t Snext = prof2s(pb0 + (rnext.w & 0xffff), pb1 + (rnext.w >> 16));
		const uint4 rnn = ring[rs + K];                                    // record of row i + 2
		dn[R3] = rcur.x, ac[R3] = rcur.y;                                  // donor[i+1], acceptor[i]
		uint32_t h, t, u;
		if (have_hD) h = hD;
		else {
			h = p_adds(Hs[R3], S);
			u = p_subs(Hr[R3], goP); t = p_max(u, Dr[R3]);
			t = p_subs(t, rcur.z); Dr[R3] = t; h = p_max(h, t);
		}
		u = p_subs(Hr[R1], ioP); t = p_subs(u, dn[R2]);
		t = p_max(t, A); A = t; h = p_max(h, p_subs(t, ac[R3]));
		u = p_subs(Hs[R1], ioP); t = p_subs(u, dn[R1]);
		t = p_max(t, B); B = t; h = p_max(h, p_subs(t, ac[R2]));
		t = p_subs(u, dn[R3]);
		t = p_max(t, C); C = t; h = p_max(h, p_subs(t, ac[R1]));
		t = p_max(p_max(Hr[R1], Hr[R2]), p_max(Hs[R1], Hs[R2]));
		h = p_max(h, p_subs(t, fsP));
		const uint32_t y = scan_max_pk<64>(p_max(p_adds(h, jge), cv[K].x));   // (the incoming carry through lane 0)
		const uint32_t ex = shift1<64>(y, cv[K].x, lane);
		h = p_max(h, p_subs(ex, gojge));
		Hr[R3] = h, Hs[R3] = shift1<64>(h, cv[K].y, lane);
		*(lds_u2p)(uintptr_t)(xwr + 8 * K) = v2u{ y, h };
		hxv[K] = (p_adds(h, ebv) & keepP) ^ flipP;
		if (i == 2 && gw == 0 && lane == 0) Hs[R1] = NEGP, Hs[R2] = NEGP;
		S = Snext;
		rcur = rnext, rnext = rnn, have_hD = false;
	};

	// Wave w runs 3 w rows behind wave 0: one barrier per step of three rows (the exchange slots hold three rows), w
	// barriers of delay first and NW-1-w at the end so that every wave passes the same number.  Step parity = (step + w) & 1.
	for (int k = 0; k < w; ++k) lds_barrier();
	{
		int par = w & 1;
		int32_t i = 2, flushed = 2;                                        // keys of rows [2, flushed) are in HBM
		int rs = 0, rs12 = 0, r96 = 0;                                     // (i - 2) % 48, % 12, % 96 at the top of a step
		auto flush_keys = [&](int32_t upto) {                              // last wave only: rows [flushed, upto), 64 at a time
			for (; flushed < upto; flushed += 64) {
				const int32_t r = flushed + lane;
				if (r < upto) {
					const uint32_t *slot = kring + ((r - 2) % EXT_WIDE_KROWS) * 2;
					uint32_t b0 = slot[0], b1 = slot[1];
					int32_t w0 = 0, w1 = 0;
#pragma unroll
					for (int ww = 1; ww < NW; ++ww) {                          // best block, ties to the leftmost
						const uint32_t v0 = slot[ww * 2 * EXT_WIDE_KROWS], v1 = slot[ww * 2 * EXT_WIDE_KROWS + 1];
						if ((v0 >> 6) > (b0 >> 6)) b0 = v0, w0 = ww;
						if ((v1 >> 6) > (b1 >> 6)) b1 = v1, w1 = ww;
					}
					const int32_t wb = SPLIT ? blk * NWB : 0;
					const uint32_t key0 = (b0 >> 6) << 16 | (0xffffu - (uint32_t)((wb + w0) * 64 + 63 - (int32_t)(b0 & 63)));
					const uint32_t key1 = (b1 >> 6) << 16 | (0xffffu - (uint32_t)((wb + w1) * 64 + 63 - (int32_t)(b1 & 63)));
					if (SPLIT) {                                               // best value, then smallest column, over the workgroups too
						atomicMax(&rowkey[r], key0);
						atomicMax(&rowkey[a.key_stride + r], key1);
					} else rowkey[r] = key0, rowkey[a.key_stride + r] = key1;
				}
			}
		};
		auto advance = [&]() { i += 3, par ^= 1, rs = rs == RING - 3 ? 0 : rs + 3, rs12 = rs12 == 9 ? 0 : rs12 + 3, r96 = r96 == EXT_WIDE_KROWS - 3 ? 0 : r96 + 3; };
		// a step of compiled rows (1..3 of them): rows [2, 14) -- the virtual column -1 of rows 2 and 3 lives there --, and what the
		// blocks of asm rows leave over at the end
		auto generic_step = [&](const int n_rows) {
			if (has_left && w == 0 && rs == 0) fetch_left(i);
			if (w == 0 && rs12 == 0) refill(i, rs);
			step_begin(r96, par);
			row(std::integral_constant<int, 0>(), i, rs);
			if (n_rows > 1) row(std::integral_constant<int, 1>(), i + 1, rs);
			if (n_rows > 2) row(std::integral_constant<int, 2>(), i + 2, rs);
			keys(r96, n_rows);
			step_end(i, par, n_rows);
			if (w == NW - 1 && i + 3 - flushed >= 64) flush_keys(flushed + 64);
			lds_barrier();
			advance();
		};
		hxv[0] = hxv[1] = hxv[2] = NEGP;
		while (i < 14 && i + 3 <= max_nl) generic_step(3);
		// ---- the asm rows, in blocks of four steps (i = 2 + 12 k: the record ring is refilled, the boundary values are fetched and
		// the keys are flushed at block starts only)
		if (i + 12 <= max_nl) {
			uint32_t tA_, tB_, tC_, h, x, hDn, a0, a1;
			v2s sraw0, sraw1;                                              // (.x = a profile score; .y is never set: v_perm_b32 takes the low halves)
			M[0] = p_max(Hr[0], Hs[0]), M[1] = p_max(Hr[1], Hs[1]);
			{                                                              // row i's diagonal term and D state (slot R3 = 2)
				const uint32_t u = p_subs(Hr[2], goP), t = p_max(u, Dr[2]);
				Dr[2] = p_subs(t, rcur.z);
				hD = p_max(p_adds(Hs[2], S), Dr[2]);
			}
			sraw0.x = *(lds_s16p)(uintptr_t)(pb0 + (rnext.w & 0xffff)), sraw1.x = *(lds_s16p)(uintptr_t)(pb1 + (rnext.w >> 16));
			uint4 rnn = ring[rs];                                          // record of row i + 2; from here on fetched a row before it is needed
			do {
				if (has_left && w == 0 && rs == 0) fetch_left(i);
				if (w == 0) refill(i, rs);
				if (w == NW - 1 && i - flushed >= 64) flush_keys(flushed + 64);
				const uint4 *rb = ring + rs;
#pragma unroll 1
				for (int st = 0; st < 4; ++st, rb += 3) {
					step_begin(r96, par);
#define MPA_WIDE_ROW(K, R1, R2, R3) { \
					MPA_ROW_HEAD(Hr[R1], Hs[R1], dn[R1], dn[R2], rcur.x, ac[R1], ac[R2], rcur.y, M[R1], M[R2]); \
					dn[R3] = rcur.x, ac[R3] = rcur.y; \
					MPA_ROW_TAIL_W(Hr[R2], Hs[R2], Dr[R2], Hs[R3], M[R3], cv[K].x, cv[K].y); \
					{ v2s f0, f1; f0.x = *(lds_s16p)(uintptr_t)a0, f1.x = *(lds_s16p)(uintptr_t)a1; sraw0 = f0, sraw1 = f1; } \
					rcur = rnext, rnext = rnn, rnn = rb[K + 1]; \
					Hr[R3] = h, hD = hDn; \
					*(lds_u2p)(uintptr_t)(xwr + 8 * K) = v2u{ x, h }; \
					hxv[K] = (p_adds(h, ebv) & keepP) ^ flipP; }
					MPA_WIDE_ROW(0, 0, 1, 2)
					MPA_WIDE_ROW(1, 2, 0, 1)
					MPA_WIDE_ROW(2, 1, 2, 0)
#undef MPA_WIDE_ROW
					keys(r96, 3);
					step_end(i, par, 3);
					lds_barrier();
					i += 3, par ^= 1, r96 += 3;
				}
				rs = rs == RING - 12 ? 0 : rs + 12, r96 = r96 == EXT_WIDE_KROWS ? 0 : r96;
			} while (i + 12 <= max_nl);
			have_hD = true;
		}
		while (i + 3 <= max_nl) generic_step(3);
		if (i < max_nl) generic_step(max_nl - i);
		if (w == NW - 1) flush_keys(max_nl);
	}
	for (int k = w; k < NW - 1; ++k) lds_barrier();
	__threadfence();
	__syncthreads();
	if (SPLIT) {
		// the replay needs the keys of every workgroup of the group: the last one to get here does it
		int32_t *flag = (int32_t*)xin;
		if (wp.tg == 0) *flag = atomicAdd(&a.done[pos.group], 1);
		__syncthreads();
		if (*flag != pos.n_blk - 1) return;
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
	}

	// ---- replay of nasw-sse.c:423-443 over the per-row keys, one wave per half
	if (w < 2 && tid[w] >= 0) {
		const int h = w;
		const uint32_t *key = rowkey + (int64_t)h * a.key_stride;
		const int64_t pen_len = 3 * (int64_t)al[h];
		int32_t G = INT32_MIN, best_i = -1, best_sc = INT32_MIN, best_col = -1;
		bool stopped = false;
		for (int32_t base = 2; base < nl[h] && !stopped; base += 64) {
			const int32_t i = base + lane;
			const bool ok = i < nl[h];
			const uint32_t kv = ok ? __builtin_nontemporal_load(&key[i]) : 0u;
			// tmp_sc_log of row i: row maximum minus the length penalty (looked up in the step table)
			int32_t pen = 0;
			{
				const int64_t x = (int64_t)i - pen_len;
				int k = 0;
				while (k + 1 < MPA_PEN_MAX && x >= (int64_t)a.pen.x[k + 1]) ++k;
				pen = a.pen.val[k];
			}
			const int32_t sc = (int32_t)(kv >> 16) - 32768;
			const int32_t v = ok ? sc - pen : INT32_MIN;
			// inclusive running maximum (over lanes, seeded with G)
			int32_t m = v;
#pragma unroll
			for (int off = 1; off < 64; off <<= 1) { const int32_t o = __shfl_up(m, off); if (lane >= off) m = imax(m, o); }
			m = imax(m, G);
			const int32_t up = __shfl_up(m, 1);                    // cross-lane reads must not sit inside a divergent select
			const bool imp = ok && v > (lane == 0 ? G : up);
			const bool brk = ok && (int64_t)m - (int64_t)v > c.xdrop;
			const uint64_t bm = __ballot(brk);
			const int first_brk = bm ? __ffsll((unsigned long long)bm) - 1 : 64;
			const uint64_t im = __ballot(imp) & (first_brk >= 63 ? ~0ULL : ((2ULL << first_brk) - 1));
			if (im) {
				const int last = 63 - __clzll((long long)im);
				best_i = base + last;
				G = __shfl(v, last);
				best_sc = __shfl(sc, last);
				best_col = 0xffff - (int32_t)(__shfl((int)kv, last) & 0xffff);
			}
			if (bm) stopped = true;
		}
		if (lane == 0) {
			ExtOut o;
			o.nt_len = best_i + 1;
			o.aa_len = best_i < 0 ? 0 : best_col + 1;
			o.score = best_sc;
			o.flags = (best_i >= 0 && best_col >= al[h]) ? 1 : 0;
			a.out[tid[h]] = o;
		}
	}
}





// ------------------------------------------------------------------------------------------------
// K2: global alignment with traceback.  int32 arithmetic, one column per lane, one call per group of
// G lanes; matrices wider than 64 columns are swept in column blocks (block-major), each block leaving
// a 16-byte boundary record per row for its right neighbour.
// ------------------------------------------------------------------------------------------------
struct GlobArgs {
	const DTask *tasks;
	const GlobWave *waves;
	const uint32_t *rec;
	const int16_t *prof;
	uint16_t *tb;
	int4 *bnd;            // {prefix of the I scan, prefix of the carry scan, H | h1<<16 (int16 each), I1}
	int32_t *score;       // per task
	DpConst c;
	unsigned long long *rowkey64;   // extension calls wider than 1024 columns only: per-row keys at t.tb_off (see k_ext_huge)
};

#define SEG_BIG (1 << 20)

// (x << 16 | 15 - k): the maximum over such keys is the best value and, among equal values, the smallest state index k --
// the outcome of the reference's chain of strict '>' comparisons (nasw-sse.c:451-519) without ten compare/select pairs.  x is
// the LOW half of v (one v_lshl_or_b32, whatever the high half holds).
__device__ __forceinline__ int32_t st_key(uint32_t v, int k) { return (int32_t)((v << 16) | (uint32_t)(15 - k)); }
__device__ __forceinline__ int32_t imax3(int32_t a, int32_t b, int32_t c) { return imax(imax(a, b), c); }
// (acc << 1) | (d < 0): collects "a > b" flags as the sign bits of b - a, one v_alignbit_b32 each
__device__ __forceinline__ uint32_t push_neg(uint32_t acc, uint32_t d) { return __builtin_amdgcn_alignbit(acc, d, 31); }
// an int16-range value in both halves of a register
__device__ __forceinline__ uint32_t spl(int32_t x) { return __builtin_amdgcn_perm((uint32_t)x, (uint32_t)x, 0x01000100u); }

// The state a traceback lane carries from row to row.  Every member holds its int16 value in BOTH halves of the register
// ("splat"): the reference computes these in saturating int16 (nasw-sse.c:444-542), and on a splat register one
// v_pk_{add,sub}_i16 clamp IS that operation (an int32 lane needs a subtract and a clamp), the sign of the value is bit 31,
// and signed int32 comparison of two splats orders them like the values.  Rows are processed in groups of three with the
// slot of row i fixed at compile time (i mod 3), so nothing is moved between rows.
struct GlobState { uint32_t H[3], Hs[3], D[3], A, B, C, dn[3], ac[3]; };
__device__ __forceinline__ void glob_state_init(GlobState &s, const uint32_t q0, const uint32_t q1, const bool origin, const int32_t fs)
{
	for (int k = 0; k < 3; ++k) s.H[k] = s.Hs[k] = s.D[k] = NEGP, s.dn[k] = s.ac[k] = 0;
	s.A = s.B = s.C = NEGP;
	// rows 0 and 1 of the first column's left neighbour (the matrix's column -1): H(-1,-1) = 0, frameshifted starts
	if (origin) s.Hs[2] = 0, s.Hs[0] = s.Hs[1] = splat16((int16_t)-fs);
	s.dn[0] = __builtin_amdgcn_perm(q0, q0, 0x08010801u), s.dn[1] = __builtin_amdgcn_perm(q1, q1, 0x08010801u);
	s.ac[0] = __builtin_amdgcn_perm(q0, q0, 0x09030903u), s.ac[1] = __builtin_amdgcn_perm(q1, q1, 0x09030903u);
}
// Row i (K = i mod 3) up to the choice among the states other than I: returns the key of the best of them, leaves the new D, A,
// B, C in the state and the four "old value kept" differences (sign bit = direction bit) in dD..dC.  rcur = record of row i,
// S = profile score of the row's amino acid in this lane's column (splat).
// WIDE: gap-extension / frameshift penalties above 255 (-E / -F of the reference's command line, main.c:133,136) do not fit the
// record's byte: the byte then only says whether the row's codon is a stop, and the penalty (ge, or fs on a stop: nasw-sse.c:263,
// 370) is chosen here.  Only the stand-alone traceback / block-major kernels are instantiated with it (dp_exec.hip routes every
// call of such a run to them); the round kernel and its packed extension bodies never pay for it.
template<int K, bool WIDE = false>
__device__ __forceinline__ int32_t glob_cands(GlobState &s, const uint32_t rcur, const uint32_t S, const uint32_t go, const uint32_t io, const uint32_t fs,
                                              uint32_t &dD, uint32_t &dA, uint32_t &dB, uint32_t &dC, const uint32_t ge_wide = 0)
{
	constexpr int R1 = (K + 2) % 3, R2 = (K + 1) % 3, R3 = K;      // slots of rows i-1, i-2 and i-3 (the one row i overwrites)
	s.dn[R3] = __builtin_amdgcn_perm(rcur, rcur, 0x08010801u);      // donor(i), acceptor(i): sign-extended bytes 1 and 3
	s.ac[R3] = __builtin_amdgcn_perm(rcur, rcur, 0x09030903u);
	uint32_t gei = __builtin_amdgcn_perm(rcur, rcur, 0x0c020c02u);
	if (WIDE) gei = gei ? fs : ge_wide;
	const uint32_t H1 = s.H[R1], H2 = s.H[R2], H3 = s.H[R3], H1s = s.Hs[R1], H2s = s.Hs[R2], H3s = s.Hs[R3];
	const uint32_t M = p_adds(H3s, S);
	uint32_t u = p_subs(H3, go), tt;
	dD = p_subs(u, s.D[R3]);                                          // direction bits: "old value wins" = sign of new - old
	const uint32_t Dn = p_subs(p_max(u, s.D[R3]), gei);
	s.D[R3] = Dn;
	tt = p_subs(p_subs(H1, io), s.dn[R2]); dA = p_subs(tt, s.A);
	s.A = p_max(tt, s.A); const uint32_t Av = p_subs(s.A, s.ac[R3]);
	u = p_subs(H1s, io);
	tt = p_subs(u, s.dn[R1]); dB = p_subs(tt, s.B);
	s.B = p_max(tt, s.B); const uint32_t Bv = p_subs(s.B, s.ac[R2]);
	tt = p_subs(u, s.dn[R3]); dC = p_subs(tt, s.C);
	s.C = p_max(tt, s.C); const uint32_t Cv = p_subs(s.C, s.ac[R1]);
	const uint32_t F6 = p_subs(H1, fs), F7 = p_subs(H2, fs), F8 = p_subs(H1s, fs), F9 = p_subs(H2s, fs);
	return imax3(imax3(st_key(M, 0), st_key(Dn, 2), st_key(Av, 3)), imax3(st_key(Bv, 4), st_key(Cv, 5), st_key(F6, 6)),
	             imax3(st_key(F7, 7), st_key(F8, 8), st_key(F9, 9)));
}

// MB: the matrix may span several 64-column blocks (block-major sweep).  EXT: score-only extension call swept with this
// kernel's exact int32 arithmetic (calls wider than the 1024 columns k_ext_wide covers): no traceback words, instead the
// per-row key (best H + end bonus, first column) that k_ext_replay consumes.
#define GLOB_NARROW_LDS ((size_t)22 * 64 * 2 + (size_t)4 * 32 * 4)   /* bytes per wave */
// One block of rows of a call swept again from a checkpoint of the packed sweep (the checkpointed traceback, dp_device.h; k_walk):
// the sweep's row 2 is the call's row row_off + 2, the state comes from the checkpoint (nullptr: the call's first block, standard
// start), the traceback words go to tb[(row - 2) * ncol + column] in LDS.  One call, in lane group 0.
struct GlobResume {
	int32_t row_off, n_rows;
	const uint32_t *ck;
	int32_t ck_lane0, half;      // (block-major sweep of a 65..128-column call: lanes from 0, the column block is the half)
	uint16_t *tb;
	int4 *bnd;                   // block-major sweep: the boundary records of the block's rows, [n_rows + 2]
};
template<int G, bool MB, bool EXT = false, bool WIDE = false>
__device__ __forceinline__ void glob_narrow(const GlobArgs &a, const GlobWave &wv, const WavePos wp, const GlobResume *rz = nullptr)
{
	constexpr int NG = 64 / G;
	int16_t *lds_prof = (int16_t*)wp.lds;            // [NG][22][G] for the current column block
	const int lane = wp.lane, grp = lane / G, col = lane % G;
	const DpConst c = a.c;
	const int32_t tid = wv.task[grp];
	DTask t;
	if (tid >= 0) t = a.tasks[tid];
	else { t.nl = 0, t.al = 1, t.ncol = 8, t.pw = G, t.rec_off = 0, t.prof_off = 0, t.tb_off = 0, t.bnd_off = 0, t.io = 0; }
	const int32_t nl = rz ? (t.nl - rz->row_off < rz->n_rows + 2 ? t.nl - rz->row_off : rz->n_rows + 2) : t.nl, ncol = t.ncol, slen = ncol >> 3;
	const int32_t nblk = MB ? (ncol + 63) / 64 : 1;
	const int32_t go = c.go, ge = c.ge, goe = (int16_t)(c.go + c.ge), io = t.io, fs = c.fs;
	const uint32_t *rec = a.rec + t.rec_off + (rz ? rz->row_off : 0);
	uint16_t *tb = rz ? rz->tb - 2 * ncol : a.tb + (EXT ? 0 : t.tb_off);          // (rz: row 2 of the sweep is the block's first row)
	unsigned long long *rowkey = EXT ? a.rowkey64 + t.tb_off : nullptr;
	int4 *bnd = rz ? rz->bnd : a.bnd + t.bnd_off;

	for (int32_t blk = 0; blk < nblk; ++blk) {
		const int32_t gc = blk * G + col;                 // global column of this lane
		const bool live = tid >= 0 && gc < ncol;
		// the previous block's boundary records were written by this wave: make them visible (and drop L1).  (Not for the walk's
		// recomputed blocks, whose boundary records are in LDS: an agent-scope release is a write-back of the whole L2, and the walk
		// recomputes tens of thousands of blocks per round.)
		if (MB && blk > 0 && !rz) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
		// stage this block's profile columns
		wave_sync();
		for (int s = 0; s < NG; ++s) {
			const int ts = wv.task[s];
			if (ts < 0) continue;
			const DTask *tt = &a.tasks[ts];
			const int16_t *src = a.prof + tt->prof_off;
			for (int k = lane; k < 22 * G; k += 64) {
				int aidx = k / G, cc = k - aidx * G, gcc = blk * G + cc;
				lds_prof[s * 22 * G + k] = gcc < tt->pw ? src[aidx * tt->pw + gcc] : (int16_t)NEG16;
			}
		}
		wave_sync();
		// record ring (see k_ext): rows [0,16) now, rows [16,32) wait in pf, then 16 rows per refill
		uint32_t *ring = (uint32_t*)(lds_prof + NG * 22 * G) + grp * 32;
		const bool loader = col < 16;
		uint32_t pf = 0;
		if (loader) ring[col] = rec[col], pf = rec[16 + col];
		wave_sync();
		const int16_t *myprof = lds_prof + grp * 22 * G + col;
		const int32_t seg = slen > 0 ? gc / slen : 0;
		const bool seg_start = slen > 0 && gc % slen == 0;
		const int32_t cge = gc * ge, yoff = seg * SEG_BIG + cge;
		const bool first_blk = !MB || blk == 0, last_blk = !MB || blk == nblk - 1;

		GlobState gs;
		glob_state_init(gs, ring[0], ring[1], first_blk && col == 0 && !(rz && rz->ck), fs);
		if (rz && rz->ck) {
			// the packed sweep's registers at the top of this row: H and D of rows i-1, i-2, i-3 (this sweep's slots 1, 0, 2), A, B, C;
			// H shifted by one column follows (column -1 is -inf behind row 2)
			const uint32_t *p = rz->ck + rz->ck_lane0 + (gc < ncol ? col : 0);
			const int sh = (MB ? blk : rz->half) * 16;
			auto half_splat = [&](const uint32_t x) { return spl((int32_t)(x >> sh)); };
			gs.H[1] = half_splat(p[0]), gs.H[0] = half_splat(p[64]), gs.H[2] = half_splat(p[128]);
			gs.D[1] = half_splat(p[192]), gs.D[0] = half_splat(p[256]), gs.D[2] = half_splat(p[320]);
			gs.A = half_splat(p[384]), gs.B = half_splat(p[448]), gs.C = half_splat(p[512]);
			if (!live) { for (int k = 0; k < 3; ++k) gs.H[k] = gs.D[k] = NEGP; gs.A = gs.B = gs.C = NEGP; }
			// (block-major: the left neighbour of a later block's first column is the last column of the block before, lane 63's lower half)
			for (int k = 0; k < 3; ++k) gs.Hs[k] = shift1<G>(gs.H[k], MB && blk > 0 ? spl((int32_t)(rz->ck[(k == 0 ? 64 : k == 1 ? 0 : 128) + 63] >> ((blk - 1) * 16))) : NEGP, lane);
		}
		const uint32_t go_s = splat16(go), io_s = splat16(io), fs_s = splat16(fs), ge_s = splat16(ge);
		const char *profb = (const char*)myprof;                // + byte0(record) * G = this lane's score for the row's amino acid
		uint32_t r[3];                                          // records of rows i, i+1 (and, once fetched, i+2) at slot row mod 3
		r[2] = ring[2], r[0] = ring[3], r[1] = 0;
		const bool carry_src = seg_start && gc >= slen;         // lanes that feed the carry scan
		uint32_t keep_py = (uint32_t)NEG32, keep_h1 = (uint32_t)NEG16, keep_i1 = (uint32_t)NEG16, keep_hs = NEGP;   // !MB: see shift1_keep
		int32_t hfin = NEG16;                                   // H of the last live row

		const int32_t max_nl = rz ? nl : wv.max_nl;        // (a register copy: through the reference it is re-loaded, and waited for, every row)
		auto row = [&](auto kc, const int32_t i) {
			constexpr int K = decltype(kc)::value;
			const bool rowlive = i < nl;
			if ((i & 15) == 12) { if (loader) ring[((i + 4) & 31) + col] = pf, pf = rec[i + 20 + col]; }   // rows [i+4, i+20) published, the next 16 requested
			r[(K + 2) % 3] = ring[(i + 2) & 31];                           // two rows ahead: LDS latency is off the critical path
			const uint32_t rcur = r[K];
			const char *sp = profb + (rcur & 0xff) * G;
			const uint32_t S = prof2(sp, sp);
			// boundary record of the block to the left
			int4 bin = make_int4(NEG32, NEG32, (int)NEGP, NEG16);
			if (!first_blk && rowlive) bin = bnd[i];
			const int32_t Hb = lo16((uint32_t)bin.z), h1b = hi16((uint32_t)bin.z), I1b = bin.w;

			// candidates other than I
			uint32_t dD, dA, dB, dC;
			const int32_t knon = glob_cands<K, WIDE>(gs, rcur, S, go_s, io_s, fs_s, dD, dA, dB, dC, ge_s);
			const int32_t nonI = knon >> 16;
			// first-pass I: the chain restarts at every lane start of the striped reference (segments of slen columns)
			const int32_t py = scan_max_i32<G>(nonI + yoff);
			int32_t pex = MB ? shift1_i32<G>(py, NEG32, lane) : (int32_t)shift1_keep<G>(keep_py, (uint32_t)py, (uint32_t)NEG32, lane);
			pex = imax(pex, bin.x);
			const int32_t py_tot = imax(bin.x, __shfl(py, grp * G + G - 1));
			const int32_t I1 = imax(pex - yoff - go, NEG16);
			// ordered choice of the state, strict '>' keeps the earlier one (nasw-sse.c:451-519)
			const int32_t kbest = imax(knon, st_key((uint32_t)I1, 1));
			const int32_t h1 = kbest >> 16;
			// left neighbour's first-pass (h, I)
			const int32_t hl_raw = MB ? shift1_i32<G>(h1, first_blk ? NEG16 : h1b, lane) : (int32_t)shift1_keep<G>(keep_h1, (uint32_t)h1, (uint32_t)NEG16, lane);
			const int32_t il_raw = MB ? shift1_i32<G>(I1, first_blk ? NEG16 : I1b, lane) : (int32_t)shift1_keep<G>(keep_i1, (uint32_t)I1, (uint32_t)NEG16, lane);
			// at a lane start of the striped reference both are -inf: (sat(-inf - go)) - (-inf) = 0
			const int32_t dI = seg_start ? 0 : s_sub(hl_raw, go) - il_raw;
			// carry of I across lane starts (the reference's lazy-F loop) as a second prefix max
			const int32_t E = imax(s_sub(hl_raw, goe), s_sub(il_raw, ge));
			const int32_t z = carry_src ? E + cge : NEG32;
			int32_t pz = scan_max_i32<G>(z);
			pz = imax(pz, bin.y);
			const int32_t Gc = imax(pz - cge, NEG16);
			const int32_t h = imax(h1, Gc);
			if (!EXT) {
				// bits 9..4 of the direction word: Gc > h1 | C | B | A | D | il > hl - go; bits 3..0: the state
				if (live && rowlive) {
					const uint32_t w = push_neg(push_neg(push_neg(push_neg(push_neg((uint32_t)(h1 - Gc) >> 31, dC), dB), dA), dD), (uint32_t)dI);
					tb[(int64_t)i * ncol + gc] = (uint16_t)((w << 4) | (uint32_t)(15 - (kbest & 15)));
				}
			} else {
				// key of this block's row: best (H + end bonus) among the live columns, ties to the smallest column
				const int32_t hb = h + (gc == t.al - 1 ? c.end_bonus : 0);
				const uint32_t kk = reduce_max_u32(live ? ((uint32_t)(imax(hb, NEG16) + 32768) << 6) | (63u - (uint32_t)lane) : 0u);
				if (lane == 63 && rowlive && tid >= 0 && kk != 0)
					atomicMax(&rowkey[i], (unsigned long long)(kk >> 6) << 32 | (0xffffffffu - (uint32_t)(blk * 64 + 63 - (int32_t)(kk & 63))));
			}
			// boundary record for the block to the right
			if (!last_blk && rowlive && col == G - 1 && tid >= 0)
				bnd[i] = make_int4(py_tot, pz, (int)pack16(h, h1), I1);
			if (!EXT) hfin = rowlive ? h : hfin;
			// the new row into the slot of row i - 3
			const uint32_t hsp = spl(h);
			gs.H[K] = hsp;
			gs.Hs[K] = MB ? shift1<G>(hsp, first_blk ? NEGP : splat16(Hb), lane) : shift1_keep<G>(keep_hs, hsp, NEGP, lane);
			if (K == 2) { if (i == 2 && first_blk && col == 0) gs.Hs[0] = NEGP, gs.Hs[1] = NEGP; }
		};
		for (int32_t i = 2; i < max_nl; i += 3) {
			row(std::integral_constant<int, 2>(), i);
			if (i + 1 < max_nl) row(std::integral_constant<int, 0>(), i + 1);
			if (i + 2 < max_nl) row(std::integral_constant<int, 1>(), i + 2);
		}
		if (!EXT && !rz && tid >= 0 && nl >= 3 && gc == t.al - 1) a.score[tid] = hfin;
	}
	if (!EXT && !rz && tid >= 0 && nl < 3 && col == 0) a.score[tid] = NEG16;
}

// one launch for the narrow traceback shapes (classes 16 / 32 / 64 lanes and the block-major one), see k_ext_narrow
template<bool WIDE>
__global__ __launch_bounds__(64) void k_glob_narrow(GlobArgs a, NarrowMap m)
{
	int b = blockIdx.x;
	extern __shared__ __attribute__((aligned(16))) uint32_t lds_raw[];
	const WavePos wp = whole_block((char*)lds_raw);
	if (b < m.cnt[0]) { glob_narrow<16, false, false, WIDE>(a, a.waves[m.first[0] + b], wp); return; }
	b -= m.cnt[0];
	if (b < m.cnt[1]) { glob_narrow<32, false, false, WIDE>(a, a.waves[m.first[1] + b], wp); return; }
	b -= m.cnt[1];
	if (b < m.cnt[2]) { glob_narrow<64, false, false, WIDE>(a, a.waves[m.first[2] + b], wp); return; }
	b -= m.cnt[2];
	glob_narrow<64, true, false, WIDE>(a, a.waves[m.first[3] + b], wp);
}

// Extension calls wider than 1024 columns: one wave per call, 64-column blocks swept one after the other (each all rows),
// boundary records through HBM, per-row keys combined across blocks with 64-bit atomic maxima.
template<bool WIDE>
__global__ __launch_bounds__(64) void k_ext_huge(GlobArgs a)
{
	extern __shared__ __attribute__((aligned(16))) uint32_t lds_raw[];
	glob_narrow<64, true, true, WIDE>(a, a.waves[blockIdx.x], whole_block((char*)lds_raw));
}

// nasw-sse.c:423-443 replayed over per-row keys ((row max + 32768) << 32 | ~column): best row under the length penalty,
// x-drop stop.  One wave per call.
__global__ __launch_bounds__(64) void k_ext_replay(const DTask *tasks, const int32_t *list, int32_t n_list, const unsigned long long *rowkey64,
                                                   ExtOut *out, DpConst c, PenTable pen)
{
	MPA_SHORT_KERNEL();
	if ((int32_t)blockIdx.x >= n_list) return;
	const int32_t tid = list[blockIdx.x];
	const DTask t = tasks[tid];
	const int lane = threadIdx.x;
	const unsigned long long *key = rowkey64 + t.tb_off;
	const int64_t pen_len = 3 * (int64_t)t.al;
	int32_t G = INT32_MIN, best_i = -1, best_sc = INT32_MIN, best_col = -1;
	bool stopped = false;
	for (int32_t base = 2; base < t.nl && !stopped; base += 64) {
		const int32_t i = base + lane;
		const bool ok = i < t.nl;
		const unsigned long long kv = ok ? key[i] : 0ULL;
		int32_t pv = 0;
		{
			const int64_t x = (int64_t)i - pen_len;
			int k = 0;
			while (k + 1 < MPA_PEN_MAX && x >= (int64_t)pen.x[k + 1]) ++k;
			pv = pen.val[k];
		}
		const int32_t sc = (int32_t)(kv >> 32) - 32768;
		const int32_t v = ok ? sc - pv : INT32_MIN;
		int32_t m = v;
#pragma unroll
		for (int off = 1; off < 64; off <<= 1) { const int32_t o = __shfl_up(m, off); if (lane >= off) m = imax(m, o); }
		m = imax(m, G);
		const int32_t up = __shfl_up(m, 1);
		const bool imp = ok && v > (lane == 0 ? G : up);
		const bool brk = ok && (int64_t)m - (int64_t)v > c.xdrop;
		const uint64_t bm = __ballot(brk);
		const int first_brk = bm ? __ffsll((unsigned long long)bm) - 1 : 64;
		const uint64_t im = __ballot(imp) & (first_brk >= 63 ? ~0ULL : ((2ULL << first_brk) - 1));
		if (im) {
			const int last = 63 - __clzll((long long)im);
			best_i = base + last;
			G = __shfl(v, last);
			best_sc = __shfl(sc, last);
			best_col = (int32_t)(0xffffffffu - (uint32_t)__shfl((int)(uint32_t)kv, last));
		}
		if (bm) stopped = true;
	}
	if (lane == 0) {
		ExtOut o;
		o.nt_len = best_i + 1;
		o.aa_len = best_i < 0 ? 0 : best_col + 1;
		o.score = best_sc;
		o.flags = (best_i >= 0 && best_col >= t.al) ? 1 : 0;
		out[tid] = o;
	}
}

// ------------------------------------------------------------------------------------------------
// K2-wide: the same traceback DP for matrices of 65..64*NW columns with one wave per 64-column block, the
// waves skewed by one row (see k_ext_wide); the boundary record travels through LDS instead of HBM and
// the block-major passes of the narrow kernel collapse into one sweep of nl + NW steps.
// ------------------------------------------------------------------------------------------------
#define GLOB_WIDE_LDS(NW) ((size_t)(NW) * 22 * 64 * 2 + 2 * (NW) * 16 + 64 * 4)   /* bytes per group */
template<int NW, bool WIDE = false>
__device__ __forceinline__ void glob_wide_body(const GlobArgs &a, const GlobWave &wv, const WavePos wp)
{
	constexpr int G = 64;
	int16_t *lds_prof = (int16_t*)wp.lds;                       // [NW][22][64]
	int4 *xB = (int4*)(lds_prof + NW * 22 * 64);               // [2 parities][NW] boundary records
	const int lane = wp.lane, w = wp.w;                         // w: wave index inside the group, scalar
	const DpConst c = a.c;
	const int32_t tid = wv.task[0];
	const DTask t = a.tasks[tid];
	const int32_t nl = t.nl, ncol = t.ncol, slen = ncol >> 3;
	const int32_t go = c.go, ge = c.ge, goe = (int16_t)(c.go + c.ge), io = t.io, fs = c.fs;
	const uint32_t *rec = a.rec + t.rec_off;
	uint16_t *tb = a.tb + t.tb_off;
	const int32_t gc = w * 64 + lane;
	const bool live = gc < ncol;
	{
		const int16_t *src = a.prof + t.prof_off;
		for (int k = lane; k < 22 * 64; k += 64) {
			const int aidx = k >> 6, cc = k & 63, gcc = w * 64 + cc;
			lds_prof[w * 22 * 64 + k] = gcc < t.pw ? src[aidx * t.pw + gcc] : (int16_t)NEG16;
		}
		if (wp.tg < 2 * NW) xB[wp.tg] = make_int4(NEG32, NEG32, (int)NEGP, NEG16);
	}
	// record ring of 64 rows shared by the waves, maintained by wave 0 (see k_ext_wide)
	uint32_t *ring = (uint32_t*)(xB + 2 * NW);
	const bool loader = w == 0 && lane < 16;
	uint32_t pf = 0;
	if (w == 0 && lane < 32) ring[lane] = rec[lane];
	if (loader) pf = rec[32 + lane];
	__syncthreads();
	const int16_t *myprof = lds_prof + w * 22 * 64 + lane;
	const int32_t seg = slen > 0 ? gc / slen : 0;
	const bool seg_start = slen > 0 && gc % slen == 0;
	const int32_t cge = gc * ge, yoff = seg * SEG_BIG + cge;
	const bool first_blk = w == 0;

	GlobState gs;
	glob_state_init(gs, ring[0], ring[1], first_blk && lane == 0, fs);
	const uint32_t go_s = splat16(go), io_s = splat16(io), fs_s = splat16(fs), ge_s = splat16(ge);
	const char *profb = (const char*)myprof;
	uint32_t rcur = ring[2];
	const bool carry_src = seg_start && gc >= slen;              // lanes that feed the carry scan

	for (int k = 0; k < w; ++k) lds_barrier();                   // skew: wave w runs w rows behind wave 0
	auto row = [&](auto kc, const int32_t i) {
		constexpr int K = decltype(kc)::value;
		const int32_t par = (i + w) & 1;
		{
			if (w == 0) {
				if ((i & 15) == 0) { if (i > 2 && loader) pf = rec[i + 32 + lane]; }
				else if ((i & 15) == 12) { if (loader) ring[((i + 20) & 63) + lane] = pf; }
			}
			const uint32_t rnext = ring[(i + 1) & 63];
			const char *sp = profb + (rcur & 0xff) * G;
			const uint32_t S = prof2(sp, sp);
			int4 bin = make_int4(NEG32, NEG32, (int)NEGP, NEG16);
			if (!first_blk) bin = xB[(par ^ 1) * NW + w - 1];
			const int32_t Hb = lo16((uint32_t)bin.z), h1b = hi16((uint32_t)bin.z), I1b = bin.w;
			uint32_t dD, dA, dB, dC;
			const int32_t knon = glob_cands<K, WIDE>(gs, rcur, S, go_s, io_s, fs_s, dD, dA, dB, dC, ge_s);
			const int32_t nonI = knon >> 16;
			const int32_t py = scan_max_i32<G>(nonI + yoff);
			int32_t pex = shift1_i32<G>(py, NEG32, lane);
			pex = imax(pex, bin.x);
			const int32_t py_tot = imax(bin.x, __shfl(py, 63));
			const int32_t I1 = imax(pex - yoff - go, NEG16);
			const int32_t kbest = imax(knon, st_key((uint32_t)I1, 1));
			const int32_t h1 = kbest >> 16;
			const int32_t hl_raw = shift1_i32<G>(h1, first_blk ? NEG16 : h1b, lane);
			const int32_t il_raw = shift1_i32<G>(I1, first_blk ? NEG16 : I1b, lane);
			const int32_t dI = seg_start ? 0 : s_sub(hl_raw, go) - il_raw;   // (at a lane start both are -inf: the difference is 0)
			const int32_t E = imax(s_sub(hl_raw, goe), s_sub(il_raw, ge));
			const int32_t z = carry_src ? E + cge : NEG32;
			int32_t pz = scan_max_i32<G>(z);
			pz = imax(pz, bin.y);
			const int32_t Gc = imax(pz - cge, NEG16);
			const int32_t h = imax(h1, Gc);
			// bits 9..4 of the direction word: Gc > h1 | C | B | A | D | il > hl - go; bits 3..0: the state
			if (live) {
				const uint32_t wd = push_neg(push_neg(push_neg(push_neg(push_neg((uint32_t)(h1 - Gc) >> 31, dC), dB), dA), dD), (uint32_t)dI);
				tb[(int64_t)i * ncol + gc] = (uint16_t)((wd << 4) | (uint32_t)(15 - (kbest & 15)));
			}
			if (lane == 63) xB[par * NW + w] = make_int4(py_tot, pz, (int)pack16(h, h1), I1);
			const uint32_t hsp = spl(h);
			gs.H[K] = hsp;
			gs.Hs[K] = shift1<G>(hsp, first_blk ? NEGP : splat16(Hb), lane);
			if (K == 2) { if (i == 2 && first_blk && lane == 0) gs.Hs[0] = NEGP, gs.Hs[1] = NEGP; }
			rcur = rnext;
		}
		lds_barrier();
	};
	for (int32_t i = 2; i < nl; i += 3) {
		row(std::integral_constant<int, 2>(), i);
		if (i + 1 < nl) row(std::integral_constant<int, 0>(), i + 1);
		if (i + 2 < nl) row(std::integral_constant<int, 1>(), i + 2);
	}
	for (int k = w; k < NW - 1; ++k) lds_barrier();
	if (nl >= 3 && gc == t.al - 1) { const int k = (nl - 1) % 3; a.score[tid] = lo16(k == 0 ? gs.H[0] : k == 1 ? gs.H[1] : gs.H[2]); }
	if (nl < 3 && wp.tg == 0) a.score[tid] = NEG16;
}

template<int NW, bool WIDE>
__global__ __launch_bounds__(NW * 64) void k_glob_wide(GlobArgs a)
{
	extern __shared__ __attribute__((aligned(16))) uint32_t lds_raw[];
	glob_wide_body<NW, WIDE>(a, a.waves[blockIdx.x], whole_block((char*)lds_raw));
}

// ------------------------------------------------------------------------------------------------
// k_dp_round: ALL DP units of a round -- extension calls of every width class and the traceback calls of the first traceback
// chunk -- in ONE launch of 256-thread workgroups.  A workgroup reads its DpUnit and its four waves become four independent
// one-wave units, two two-wave groups, one four-wave group or a quarter of a split 1024-column group.  Units are ordered by
// expected duration, longest first, so the hardware's in-order workgroup dispatcher is the task queue: a workgroup slot that
// frees up takes the longest unit not yet started.  One launch instead of eight to ten concurrent ones means one hardware
// queue per batch in flight, which is what lets a stream of batches keep more than two batches in their DP rounds (each
// concurrently running kernel occupies a queue, and more queues than the hardware schedules at once are time-sliced).
// ------------------------------------------------------------------------------------------------
constexpr size_t dp_round_max(size_t a, size_t b) { return a > b ? a : b; }
constexpr size_t DP_ROUND_LDS = dp_round_max(dp_round_max(dp_round_max(4 * EXT_NARROW_LDS, 2 * EXT_WIDE_LDS(2)), dp_round_max(EXT_WIDE_LDS(4), 4 * GLOB_NARROW_LDS)),
                                             dp_round_max(2 * GLOB_WIDE_LDS(2), GLOB_WIDE_LDS(4)));   // 30 400 bytes: five workgroups per CU
// one DpUnit on the four waves of a 256-thread workgroup (lds: the workgroup's DP_ROUND_LDS bytes)
// WG_ONLY: only the kinds that take a whole workgroup (the worker pool runs the one-wave kinds wave by wave)
template<bool WG_ONLY>
__device__ __forceinline__ void dp_run_unit(const DpUnit &u, const ExtArgs &ea, const ExtWideArgs &wa, const GlobArgs &ga, char *lds, const int tid)
{
	const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
	switch (u.pad_) {                                             // issue priority by expected duration (dp_exec.hip, launch_round)
	case 3: __builtin_amdgcn_s_setprio(3); break;
	case 2: __builtin_amdgcn_s_setprio(2); break;
	case 1: __builtin_amdgcn_s_setprio(1); break;
	default: __builtin_amdgcn_s_setprio(0); break;
	}
	switch (u.kind) {
	case U_EXT16: if (!WG_ONLY && wv < u.count) ext_narrow<16>(ea, ea.waves + u.first + wv, WavePos{ lds + wv * EXT_NARROW_LDS, lane, 0, lane }); break;
	case U_EXT32: if (!WG_ONLY && wv < u.count) ext_narrow<32>(ea, ea.waves + u.first + wv, WavePos{ lds + wv * EXT_NARROW_LDS, lane, 0, lane }); break;
	case U_EXT64: if (!WG_ONLY && wv < u.count) ext_narrow<64>(ea, ea.waves + u.first + wv, WavePos{ lds + wv * EXT_NARROW_LDS, lane, 0, lane }); break;
	case U_EXT_W2: {
		const int g = wv >> 1, w = wv & 1;
		if (!WG_ONLY && g < u.count) ext_wide_body<2, false>(wa, u.first + g, SplitPos{ 0, 0, 1, 0 }, WavePos{ lds + g * EXT_WIDE_LDS(2), lane, w, w * 64 + lane });
	} break;
	case U_EXT_W4: ext_wide_body<4, false>(wa, u.first, SplitPos{ 0, 0, 1, 0 }, WavePos{ lds, lane, wv, tid }); break;
	case U_EXT_SPLIT: ext_wide_body<4, true>(wa, u.first, SplitPos{ u.sgroup, u.blk, u.n_blk, u.xg_first }, WavePos{ lds, lane, wv, tid }); break;
	case U_GLOB16: if (!WG_ONLY && wv < u.count) glob_narrow<16, false>(ga, ga.waves[u.first + wv], WavePos{ lds + wv * GLOB_NARROW_LDS, lane, 0, lane }); break;
	case U_GLOB32: if (!WG_ONLY && wv < u.count) glob_narrow<32, false>(ga, ga.waves[u.first + wv], WavePos{ lds + wv * GLOB_NARROW_LDS, lane, 0, lane }); break;
	case U_GLOB64: if (!WG_ONLY && wv < u.count) glob_narrow<64, false>(ga, ga.waves[u.first + wv], WavePos{ lds + wv * GLOB_NARROW_LDS, lane, 0, lane }); break;
	case U_GLOB_MB: if (!WG_ONLY && wv < u.count) glob_narrow<64, true>(ga, ga.waves[u.first + wv], WavePos{ lds + wv * GLOB_NARROW_LDS, lane, 0, lane }); break;
	case U_GLOB_W2: {
		const int g = wv >> 1, w = wv & 1;
		if (!WG_ONLY && g < u.count) glob_wide_body<2>(ga, ga.waves[u.first + g], WavePos{ lds + g * GLOB_WIDE_LDS(2), lane, w, w * 64 + lane });
	} break;
	case U_GLOB_W4: glob_wide_body<4>(ga, ga.waves[u.first], WavePos{ lds, lane, wv, tid }); break;
	case U_LITE16: if (!WG_ONLY && wv < u.count) ext_narrow<16, true>(ea, ea.waves + u.first + wv, WavePos{ lds + wv * EXT_NARROW_LDS, lane, 0, lane }); break;
	case U_LITE32: if (!WG_ONLY && wv < u.count) ext_narrow<32, true>(ea, ea.waves + u.first + wv, WavePos{ lds + wv * EXT_NARROW_LDS, lane, 0, lane }); break;
	case U_LITE64: if (!WG_ONLY && wv < u.count) ext_narrow<64, true>(ea, ea.waves + u.first + wv, WavePos{ lds + wv * EXT_NARROW_LDS, lane, 0, lane }); break;
	case U_LITE128: if (!WG_ONLY && wv < u.count) ext_narrow<64, true, true>(ea, ea.waves + u.first + wv, WavePos{ lds + wv * EXT_NARROW_LDS, lane, 0, lane }); break;
	case U_EXT128: if (!WG_ONLY && wv < u.count) ext_narrow<64, false, true>(ea, ea.waves + u.first + wv, WavePos{ lds + wv * EXT_NARROW_LDS, lane, 0, lane }); break;
	default: break;
	}
}

// (the default executor: one launch per round, one workgroup per unit, the hardware's in-order dispatcher as the task queue)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void k_dp_round(ExtArgs ea, ExtWideArgs wa, GlobArgs ga, const DpUnit *units)
{
	extern __shared__ __attribute__((aligned(16))) uint32_t lds_raw[];
	const DpUnit u = units[blockIdx.x];
	dp_run_unit<false>(u, ea, wa, ga, (char*)lds_raw, (int)threadIdx.x);
}

// ------------------------------------------------------------------------------------------------
// The DP worker pool (MPA_DP_POOL=1; off by default -- measured level to 10 % behind the launches): the units of ALL batches in their DP rounds on a device are served by one bounded
// population of worker workgroups instead of one launch per round.
//
// Why.  A round's units are a few thousand short ones and a handful that sweep 100 000 - 190 000 rows (25 - 50 ms at one row per
// ~255 ns).  As one launch per round, a round occupies every workgroup slot the LDS allows for its first milliseconds and then
// idles the chip behind its stragglers; several batches in flight take turns at flooding the CUs (round 4: 20 000 waves per
// launch, 114 GCUPS per launch against 924 GCUPS for the same kernel on a full machine; every kernel next to a starting round
// stretched 2-6 x).  With the pool the number of resident DP workgroups is bounded and steady, a straggler of batch k runs
// under the bulk of batches k+1.., and a lane's workers that run out of their own round's units take another lane's.
//
// How.  Every DP lane owns a slot of the device's DpPool: its round's arguments and unit list (longest unit first).  The
// lane arms the slot (k_dp_arm, in stream order behind the round's uploads and prep kernels) and launches workers on a stream
// of its own; a worker takes units from its home slot first, then from the other slots, and EXITS when no slot has a unit
// left -- no workgroup ever waits for the host, so nothing can hang on a host that is slow, blocked in hipFree, or gone.
// Admission: a worker beyond the pool's budget of resident workers leaves at once (the first 64 workgroups of a launch are
// always admitted: every round brings workers of its own, whatever the others do).  A round is complete when its `remaining`
// counter reaches zero -- the lane polls a word of pinned host memory -- not when any particular launch ends.
// Visibility: a worker that switches to another (slot, generation) does an agent-scope acquire before it reads the round's
// inputs (written by kernels and copies that finished before the slot was armed, possibly while this worker was already
// running: its CU's L1 and its XCD's L2 may hold lines of the round before); a unit's results are written back from the L2s by
// k_l2_writeback, which the lane launches behind the completed round, before anything reads them.
// ------------------------------------------------------------------------------------------------
struct DpRoundArgs { ExtArgs ea; ExtWideArgs wa; GlobArgs ga; const DpUnit *units; int32_t n_group, pad_; };
struct DpPool { DpPoolCtl ctl; DpSlotCtl slot[MPA_DP_SLOTS]; DpRoundArgs args[MPA_DP_SLOTS]; };
#define MPA_DP_ALWAYS_ADMITTED 64

__device__ __forceinline__ unsigned long long ld_u64_agent(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// arms slot `s` for a round of n_group workgroup units and n_wave one-wave units (generation gen); single thread, stream-ordered
// behind everything the round reads
__global__ void k_dp_arm(DpPool *pool, int s, int n_group, int n_wave, unsigned int gen, int32_t *done, long long *trace)
{
	DpSlotCtl &c = pool->slot[s];
	__hip_atomic_store(&c.remaining, n_group + n_wave, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	__hip_atomic_store(&c.stamp, atomicAdd(&pool->ctl.arm_seq, 1) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	c.done = done, c.trace = trace;
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
	__hip_atomic_store(&c.lim[0], (unsigned long long)gen << 32 | (unsigned int)n_group, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	__hip_atomic_store(&c.lim[1], (unsigned long long)gen << 32 | (unsigned int)n_wave, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
	__hip_atomic_store(&c.head[0], (unsigned long long)gen << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	__hip_atomic_store(&c.head[1], (unsigned long long)gen << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Writes the dirty lines of every XCD's L2 back to memory: the results of a round's units, wherever their workers ran.  Launched by
// the lane behind a completed round, 16 workgroups per XCD (workgroup b runs on XCD b % 8; any one per XCD would do).
__global__ __launch_bounds__(64) void k_l2_writeback()
{
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
}

// One lane takes the next unit of queue q (0: workgroup units, 1: one-wave units) of the first slot that has one, the home slot
// first.  Returns the slot or -1.
__device__ __forceinline__ int dp_pull(DpPool *pool, const int q, const int home, const int n_slots, int &unit, unsigned int &gen)
{
	for (int k = 0; k < n_slots; ++k) {
		const int s = home + k < n_slots ? home + k : home + k - n_slots;
		DpSlotCtl &c = pool->slot[s];
		for (;;) {
			// look first (plain atomic loads: an empty slot costs its scanners no read-modify-write), then take with ONE fetch-add --
			// never a compare-and-swap loop: with several hundred workgroups at one head word every failed attempt is another
			// round trip to the L2 (measured: a launch of 1 024 workers on 1 024 units took 8.4 ms against 2.3 ms)
			const unsigned long long h = ld_u64_agent(&c.head[q]), l = ld_u64_agent(&c.lim[q]);
			if ((h >> 32) != (l >> 32) || (unsigned int)h >= (unsigned int)l) break;          // being armed, or no unit left
			const unsigned long long h2 = __hip_atomic_fetch_add(&c.head[q], 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			const unsigned long long l2 = ld_u64_agent(&c.lim[q]);
			// (an index past the end is nobody's unit: the head is set afresh when the slot is armed again, and a head of a
			// generation can only be seen once that generation's lim has been written)
			if ((h2 >> 32) == (l2 >> 32) && (unsigned int)h2 < (unsigned int)l2) { unit = (int)(unsigned int)h2, gen = (unsigned int)(h2 >> 32); return s; }
		}
	}
	return -1;
}

// What a worker does before it reads the inputs of a round it has not read before (MPA_DP_ACQUIRE).  2 (default): an agent-scope
// acquire if the round was armed after the last acquire on this CU that this workgroup knows of -- an acquire is not about a slot,
// it drops every stale line the CU's L1 and the XCD's L2 hold at that moment (the arguments, unit lists and task tables of a round
// arrive by host-to-device copies, which no kernel boundary makes visible to a workgroup that is already running:
// profiles/r05_stale_reads_across_kernel_boundaries.txt), and it is not free for the kernels next door; 1: whenever the workgroup
// moves to another (slot, generation); 0: never (measurement only).  The round's arguments are read through the scalar cache
// (constant address space), which the fence does not cover: invalidated by hand.  `seen` = {sequence number, slot, generation} in LDS,
// shared by the waves of the workgroup.
__device__ __forceinline__ void dp_acquire(DpPool *pool, const int mode, const int s, const unsigned int gen, volatile int32_t *seen)
{
	bool fresh = false;
	if (mode == 1) fresh = s != seen[1] || (int32_t)gen != seen[2];
	else if (mode == 2) fresh = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&pool->slot[s].stamp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) - seen[0] > 0;
	if (fresh) {
		const int now_seq = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&pool->ctl.arm_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
		__builtin_amdgcn_s_dcache_inv();
		seen[0] = now_seq, seen[1] = s, seen[2] = (int32_t)gen;
	}
}

// a unit's results are in the L2 (see the comment in the loop below): count it, and tell the lane if it was the round's last
__device__ __forceinline__ void dp_unit_done(DpPool *pool, const int s, const unsigned int gen)
{
	DpSlotCtl &c = pool->slot[s];
	if (__hip_atomic_fetch_add(&c.remaining, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 1)
		__hip_atomic_store(c.done, (int32_t)gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__device__ __forceinline__ DpUnit dp_load_unit(const DpUnit *up)
{
	DpUnit u;
	u.kind = __builtin_amdgcn_readfirstlane(up->kind), u.first = __builtin_amdgcn_readfirstlane(up->first), u.count = __builtin_amdgcn_readfirstlane(up->count);
	u.blk = __builtin_amdgcn_readfirstlane(up->blk), u.n_blk = __builtin_amdgcn_readfirstlane(up->n_blk), u.sgroup = __builtin_amdgcn_readfirstlane(up->sgroup);
	u.xg_first = __builtin_amdgcn_readfirstlane(up->xg_first), u.pad_ = __builtin_amdgcn_readfirstlane(up->pad_);
	return u;
}

// A worker workgroup.  First its four waves work TOGETHER on workgroup units (wide and split extension groups, wide traceback
// groups: bodies that use the workgroup barrier) until no slot has one left; then every wave goes its own way and takes one-wave
// units (up to eight narrow extension calls or four traceback calls each) until no slot has one of those left -- a wave whose
// calls are x-dropped after a few hundred rows is on its next unit while its neighbour sweeps 100 000 rows.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void k_dp_worker(DpPool *pool, const int home, const int n_slots)
{
	extern __shared__ __attribute__((aligned(16))) uint32_t lds_raw[];
	__shared__ int32_t pick[4];                                   // slot, unit, generation (thread 0 -> the workgroup), admission
	__shared__ int32_t seen[4];                                   // dp_acquire; [3]: waves that have left
	char *lds = (char*)lds_raw;
	int tid = (int)threadIdx.x;                                   // (made opaque once per unit, see below)
	if (tid == 0) {
		int ok = 1;
		if (blockIdx.x >= MPA_DP_ALWAYS_ADMITTED) {
			const int a = atomicAdd(&pool->ctl.alive, 1);
			if (a >= __hip_atomic_load(&pool->ctl.budget, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicSub(&pool->ctl.alive, 1), ok = 0;
		} else atomicAdd(&pool->ctl.alive, 1);
		pick[3] = ok;
		seen[0] = 0, seen[1] = -1, seen[2] = 0, seen[3] = 0;
	}
	__syncthreads();
	if (!pick[3]) return;
	const int acquire_mode = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&pool->ctl.acquire_mode, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
	typedef const __attribute__((address_space(4))) DpRoundArgs *ConstArgs;
	// ---- workgroup units
	for (;;) {
		// (the thread index is made opaque per unit: otherwise every lane-dependent constant of every body -- column masks, group
		// indices, profile addresses of twelve instantiations -- is hoisted out of the loop and stays live across it: 204 VGPRs)
		asm volatile("" : "+v"(tid));
		if (tid == 0) {
			int unit = 0;
			unsigned int gen = 0;
			pick[0] = dp_pull(pool, 0, home, n_slots, unit, gen), pick[1] = unit, pick[2] = (int32_t)gen;
		}
		__syncthreads();
		const int s = pick[0], ui = pick[1];
		const unsigned int gen = (unsigned int)pick[2];
		if (s < 0) break;
		if ((tid >> 6) == 0) dp_acquire(pool, acquire_mode, s, gen, seen);
		__syncthreads();
		// The round's arguments are read through the CONSTANT address space: they do not change while the round is live, and scalar
		// loads keep them out of the vector registers (as plain global loads behind the atomics above they cost 110 VGPRs).
		const DpRoundArgs &ra = *(const DpRoundArgs*)(ConstArgs)(uintptr_t)&pool->args[s];
		const DpUnit u = dp_load_unit(ra.units + ui);
		long long *trace = pool->slot[s].trace;
		const long long t0 = trace ? wall_clock64() : 0;               // (uniform: scalar registers)
		dp_run_unit<true>(u, ra.ea, ra.wa, ra.ga, lds, tid);
		__builtin_amdgcn_s_setprio(0);
		// The unit's results stay where the stores left them -- dirty lines of this XCD's L2: the round's consumers are kernels and
		// copies the lane enqueues AFTER it has seen the round complete, behind k_l2_writeback (one L2 write-back per XCD and round).
		// A release fence here, per wave and unit, is 40 000 whole-L2 write-backs per round: measured, it stretched every kernel on
		// the chip.  All that is needed is that the stores have left the CU before the unit counts.
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__syncthreads();
		if (tid == 0) {
			if (trace) trace[2 * ui] = t0, trace[2 * ui + 1] = wall_clock64();
			dp_unit_done(pool, s, gen);
		}
	}
	// ---- one-wave units: no workgroup barrier from here on
	const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
	char *lds_wave = lds + wv * EXT_NARROW_LDS;
	for (;;) {
		asm volatile("" : "+v"(tid));
		const int lane = tid & 63;
		int s = -1, ui = 0;
		unsigned int gen = 0;
		if (lane == 0) s = dp_pull(pool, 1, home, n_slots, ui, gen);
		s = __builtin_amdgcn_readfirstlane(s), ui = __builtin_amdgcn_readfirstlane(ui), gen = (unsigned int)__builtin_amdgcn_readfirstlane((int)gen);
		if (s < 0) break;
		dp_acquire(pool, acquire_mode, s, gen, seen);
		const DpRoundArgs &ra = *(const DpRoundArgs*)(ConstArgs)(uintptr_t)&pool->args[s];
		const int uidx = ra.n_group + ui;                            // (the one-wave units follow the workgroup units in the round's list)
		const DpUnit u = dp_load_unit(ra.units + uidx);
		long long *trace = pool->slot[s].trace;
		const long long t0 = trace ? wall_clock64() : 0;
		switch (u.pad_) {                                             // issue priority by expected duration (dp_exec.hip, launch_round)
		case 3: __builtin_amdgcn_s_setprio(3); break;
		case 2: __builtin_amdgcn_s_setprio(2); break;
		case 1: __builtin_amdgcn_s_setprio(1); break;
		default: break;
		}
		const WavePos wp{ lds_wave, lane, 0, lane };
		switch (u.kind) {
		case U_EXT16: ext_narrow<16>(ra.ea, ra.ea.waves + u.first, wp); break;
		case U_EXT32: ext_narrow<32>(ra.ea, ra.ea.waves + u.first, wp); break;
		case U_EXT64: ext_narrow<64>(ra.ea, ra.ea.waves + u.first, wp); break;
		case U_GLOB16: glob_narrow<16, false>(ra.ga, ra.ga.waves[u.first], wp); break;
		case U_GLOB32: glob_narrow<32, false>(ra.ga, ra.ga.waves[u.first], wp); break;
		case U_GLOB64: glob_narrow<64, false>(ra.ga, ra.ga.waves[u.first], wp); break;
		case U_GLOB_MB: glob_narrow<64, true>(ra.ga, ra.ga.waves[u.first], wp); break;
		case U_LITE16: ext_narrow<16, true>(ra.ea, ra.ea.waves + u.first, wp); break;
		case U_LITE32: ext_narrow<32, true>(ra.ea, ra.ea.waves + u.first, wp); break;
		case U_LITE64: ext_narrow<64, true>(ra.ea, ra.ea.waves + u.first, wp); break;
		case U_LITE128: ext_narrow<64, true, true>(ra.ea, ra.ea.waves + u.first, wp); break;
		case U_EXT128: ext_narrow<64, false, true>(ra.ea, ra.ea.waves + u.first, wp); break;
		default: break;
		}
		__builtin_amdgcn_s_setprio(0);
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		if (lane == 0) {
			if (trace) trace[2 * uidx] = t0, trace[2 * uidx + 1] = wall_clock64();
			dp_unit_done(pool, s, gen);
		}
	}
	if ((tid & 63) == 0 && atomicAdd(&seen[3], 1) == 3) atomicSub(&pool->ctl.alive, 1);   // the last wave to leave takes the workgroup off the count
}

// ------------------------------------------------------------------------------------------------
// traceback walk (ns_backtrack nasw-sse.c:40-89), one WAVE per call.
//
// The reference follows one cell per step.  The path is made of runs that move in a fixed direction while
// a bit of the traceback word stays set (match runs on the diagonal, gap/intron extensions along a column
// or a row), so the 64 lanes fetch the next 64 cells of the current direction at once, a ballot finds where
// the run ends, and the whole run is emitted as one CIGAR operation.  A 50 kb intron costs ~800 dependent
// memory round trips instead of 50 000.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_backtrack(const DTask *tasks, const int32_t *list, int32_t n_list, const uint16_t *tbpool, uint32_t *cigpool, int32_t *n_cigar)
{
	MPA_SHORT_KERNEL();
	if ((int32_t)blockIdx.x >= n_list) return;
	const int32_t tid = list[blockIdx.x];
	const DTask t = tasks[tid];
	const uint16_t *tb = tbpool + t.tb_off;
	uint32_t *cig = cigpool + t.cig_off;
	const int32_t ncol = t.ncol, cap = t.cig_cap, lane = threadIdx.x;
	int32_t i = t.nl - 1, j = t.al - 1, carry = 0, n = 0;
	// the CIGAR operation being accumulated (ns_push_cigar merges equal neighbours except F/G, nasw.h:141-152)
	int32_t cur_op = -1, cur_len = 0;
	auto push = [&](int32_t op, int32_t len) {
		if (cur_op == op && op != 10 && op != 11) { cur_len += len; return; }
		if (cur_op >= 0) { if (lane == 0 && n < cap) cig[n] = (uint32_t)cur_len << 4 | (uint32_t)cur_op; ++n; }
		cur_op = op, cur_len = len;
	};
	auto state_of = [](int32_t w) { return (w >> 9 & 1) ? 1 : (w & 0xf); };   // a cell raised by the cross-lane I counts as I
	while (i >= 2 && j >= 0) {
		int32_t st = carry;
		if (st == 0) {                                           // fresh cell: its own state decides the direction
			const int32_t w0 = tb[(int64_t)i * ncol + j];
			st = state_of(w0);
		}
		if (st == 0) {                                           // run of matches on the diagonal (i-3l, j-l)
			const int32_t ii = i - 3 * lane, jj = j - lane;
			const bool ok = ii >= 2 && jj >= 0;
			const int32_t w = ok ? tb[(int64_t)ii * ncol + jj] : 0xf;
			const uint64_t stop = __ballot(!ok || state_of(w) != 0);
			const int32_t run = stop ? __ffsll((unsigned long long)stop) - 1 : 64;
			push(0, run), i -= 3 * run, j -= run, carry = 0;
		} else if (st <= 5) {                                    // extension runs: I along the row, D/N/U/V along the column
			const int32_t di = st == 1 ? 0 : st == 2 ? 3 : 1, dj = st == 1 ? 1 : 0;
			const int32_t ii = i - di * lane, jj = j - dj * lane;
			const bool ok = ii >= 2 && jj >= 0;
			const int32_t w = ok ? tb[(int64_t)ii * ncol + jj] : 0;
			const bool ext = ok && (w >> (st + 3) & 1);
			const uint64_t inval = __ballot(!ok), noext = __ballot(!ext);
			// cells are consumed up to and including the first one whose extension bit is clear; a cell outside
			// the matrix ends the walk before it is consumed
			const int32_t first_noext = noext ? __ffsll((unsigned long long)noext) - 1 : 64;
			const int32_t first_inval = inval ? __ffsll((unsigned long long)inval) - 1 : 64;
			int32_t run;
			bool closed;                                          // did the run meet a cell with the extension bit clear?
			if (first_noext < first_inval) run = first_noext + 1, closed = true;
			else run = first_inval, closed = false;
			static const int32_t op_of[6] = { 0, 1, 2, 3, 12, 13 };
			push(op_of[st], run);
			i -= di * run, j -= dj * run;
			if (closed && (st == 4 || st == 5)) --j;               // the intron of phase 1/2 ends inside a codon
			carry = closed ? 0 : st;
			if (run == 0) break;                                  // cannot happen (the loop condition holds for lane 0)
		} else {                                                 // frameshifts: one cell
			if (st == 6) push(10, 1), i -= 1;
			else if (st == 7) push(10, 2), i -= 2;
			else if (st == 8) push(11, 1), i -= 1, j -= 1;
			else push(11, 2), i -= 2, j -= 1;
			carry = 0;
		}
	}
	if (j > 0) push(1, j);
	if (i >= 0) {
		const int32_t l = (i + 1) / 3 * 3, r = (i + 1) % 3;
		if (l > 0) push(2, l);
		if (r != 0) push(10, r);
	}
	push(-2, 0);                                                 // flush
	if (n > cap) n = cap;
	__syncthreads();
	for (int32_t x = lane; x < n >> 1; x += 64) { uint32_t tmp = cig[x]; cig[x] = cig[n - 1 - x], cig[n - 1 - x] = tmp; }
	__syncthreads();
	for (int32_t x = lane; x < n; x += 64) {                      // ns_fix_tiny_UV
		const uint32_t op = cig[x] & 0xf;
		if ((op == 12 || op == 13) && cig[x] >> 4 < 3) cig[x] = cig[x] >> 4 << 4 | 11;
	}
	if (lane == 0) n_cigar[tid] = n;
}

// ------------------------------------------------------------------------------------------------
// The walk of the checkpointed traceback (dp_device.h): ns_backtrack (nasw-sse.c:40-89) over a call whose sweep kept only the four
// extension bits per cell and a checkpoint per block of MPA_TB_BLOCK rows.  One wave per call.  Deletion and intron runs (states 2..5)
// are followed on the bits, 64 cells at a time; in any other state the wave first recomputes the traceback words of the block it
// stands in (glob_narrow restarted from the block's checkpoint, words into LDS) and walks those, exactly like k_backtrack -- a run
// that leaves the block is simply continued in the next iteration.  A call is swept twice only where its path is not in a run.
// ------------------------------------------------------------------------------------------------
struct WalkArgs {
	GlobArgs ga;                 // tasks, records, profiles, scoring
	const int32_t *list;         // the calls
	int32_t n_list;
	const uint32_t *lite, *ckpt; // extension-bit words and checkpoints of the packed sweeps
	uint32_t *cig;
	int32_t *n_cigar;
	unsigned long long *n_blocks; // (statistics) blocks recomputed
};
// (per class of the packed sweep -- 16 / 32 / 64 / 128 columns at most: 8 / 11 / 17 / 29 KB; one launch per class, so that the narrow calls'
// walks, most of them, take LDS for their own block of direction words and not for a 128-column one)
#define WALK_LDS(NC) (GLOB_NARROW_LDS + (size_t)MPA_TB_BLOCK * (NC) * 2 + (size_t)(MPA_TB_BLOCK + 2) * 16)
// DUAL: a call of 65..128 columns, swept with column c + 64 in the high half of lane c (ext_narrow<64, true, true>) and recomputed
// by the block-major traceback sweep (two blocks of 64 columns, boundary records in LDS)
template<int G, bool DUAL = false>
__device__ __forceinline__ void walk_call(const WalkArgs &wa, const DTask &t, const int32_t tid, char *lds, const int lane)
{
	constexpr int NG = 64 / G;
	const int slot = DUAL ? 0 : (t.flag >> MPA_LITE_SLOT_SHIFT) & 15, half = slot / NG, lane0 = (slot % NG) * G;
	const uint32_t *lite = wa.lite + t.tb_off + lane0;
	uint16_t *tbs = (uint16_t*)(lds + GLOB_NARROW_LDS);            // [MPA_TB_BLOCK][ncol] words of the block that is materialised
	int4 *bnds = (int4*)(lds + GLOB_NARROW_LDS + (size_t)MPA_TB_BLOCK * (DUAL ? 128 : G) * 2);
	uint32_t *cig = wa.cig + t.cig_off;
	const int32_t ncol = t.ncol, cap = t.cig_cap;
	int32_t blk = -1, blk_lo = 0, blk_hi = 0;                      // the materialised block and its rows [blk_lo, blk_hi)
	auto ensure = [&](const int32_t row) {
		if (blk >= 0 && row >= blk_lo && row < blk_hi) return;
		const int32_t b = (row - 2) / MPA_TB_BLOCK;
		GlobResume rz;
		rz.row_off = b * MPA_TB_BLOCK, rz.n_rows = t.nl - 2 - rz.row_off < MPA_TB_BLOCK ? t.nl - 2 - rz.row_off : MPA_TB_BLOCK;
		rz.ck = b ? wa.ckpt + t.bnd_off + (int64_t)(b - 1) * 9 * 64 : nullptr, rz.ck_lane0 = lane0, rz.half = half, rz.tb = tbs, rz.bnd = bnds;
		GlobWave gw;
		gw.task[0] = tid, gw.task[1] = gw.task[2] = gw.task[3] = -1, gw.max_nl = rz.n_rows + 2;
		wave_sync();
		glob_narrow<G, DUAL, false, false>(wa.ga, gw, WavePos{ lds, lane, 0, lane }, &rz);
		wave_sync();
		blk = b, blk_lo = 2 + rz.row_off, blk_hi = blk_lo + rz.n_rows;
		if (lane == 0) atomicAdd(wa.n_blocks, 1ULL);
	};
	auto full = [&](const int32_t ii, const int32_t jj) -> int32_t { return tbs[(ii - blk_lo) * ncol + jj]; };
	auto nibble = [&](const int32_t ii, const int32_t jj) -> int32_t {
		const uint32_t r = (uint32_t)(ii - 2);
		return (int32_t)(lite[(int64_t)(r / 3) * 64 + (DUAL ? jj & 63 : jj)] >> (16 * (DUAL ? jj >> 6 : half) + 4 * (2 - r % 3))) & 0xf;
	};
	int32_t i = t.nl - 1, j = t.al - 1, carry = 0, n = 0;
	int32_t cur_op = -1, cur_len = 0;                              // (ns_push_cigar merges equal neighbours except F/G, nasw.h:141-152)
	auto push = [&](int32_t op, int32_t len) {
		if (cur_op == op && op != 10 && op != 11) { cur_len += len; return; }
		if (cur_op >= 0) { if (lane == 0 && n < cap) cig[n] = (uint32_t)cur_len << 4 | (uint32_t)cur_op; ++n; }
		cur_op = op, cur_len = len;
	};
	auto state_of = [](int32_t w) { return (w >> 9 & 1) ? 1 : (w & 0xf); };
	while (i >= 2 && j >= 0) {
		int32_t st = carry;
		if (st == 0) { ensure(i); st = state_of(full(i, j)); }          // fresh cell: its own state decides the direction
		if (st == 0) {                                               // run of matches on the diagonal (i-3l, j-l), inside the block
			const int32_t ii = i - 3 * lane, jj = j - lane;
			const bool ok = ii >= blk_lo && jj >= 0;
			const int32_t w = ok ? full(ii, jj) : 0xf;
			const uint64_t stop = __ballot(!ok || state_of(w) != 0);
			const int32_t run = stop ? __ffsll((unsigned long long)stop) - 1 : 64;
			push(0, run), i -= 3 * run, j -= run, carry = 0;
		} else if (st <= 5) {                                        // extension runs: I along the row (words), D/N/U/V along the column (bits)
			const int32_t di = st == 1 ? 0 : st == 2 ? 3 : 1, dj = st == 1 ? 1 : 0;
			if (st == 1) ensure(i);
			const int32_t ii = i - di * lane, jj = j - dj * lane;
			const bool ok = ii >= 2 && jj >= 0;
			bool ext = false;
			if (ok) ext = st == 1 ? (full(ii, jj) >> 4 & 1) != 0 : (nibble(ii, jj) >> (st - 2) & 1) != 0;
			const uint64_t inval = __ballot(!ok), noext = __ballot(!ext);
			const int32_t first_noext = noext ? __ffsll((unsigned long long)noext) - 1 : 64;
			const int32_t first_inval = inval ? __ffsll((unsigned long long)inval) - 1 : 64;
			int32_t run;
			bool closed;
			if (first_noext < first_inval) run = first_noext + 1, closed = true;
			else run = first_inval, closed = false;
			static const int32_t op_of[6] = { 0, 1, 2, 3, 12, 13 };
			push(op_of[st], run);
			i -= di * run, j -= dj * run;
			if (closed && (st == 4 || st == 5)) --j;
			carry = closed ? 0 : st;
			if (run == 0) break;
		} else {                                                     // frameshifts: one cell
			if (st == 6) push(10, 1), i -= 1;
			else if (st == 7) push(10, 2), i -= 2;
			else if (st == 8) push(11, 1), i -= 1, j -= 1;
			else push(11, 2), i -= 2, j -= 1;
			carry = 0;
		}
	}
	if (j > 0) push(1, j);
	if (i >= 0) {
		const int32_t l = (i + 1) / 3 * 3, r = (i + 1) % 3;
		if (l > 0) push(2, l);
		if (r != 0) push(10, r);
	}
	push(-2, 0);                                                 // flush
	if (n > cap) n = cap;
	__syncthreads();
	for (int32_t x = lane; x < n >> 1; x += 64) { uint32_t tmp = cig[x]; cig[x] = cig[n - 1 - x], cig[n - 1 - x] = tmp; }
	__syncthreads();
	for (int32_t x = lane; x < n; x += 64) {                      // ns_fix_tiny_UV
		const uint32_t op = cig[x] & 0xf;
		if ((op == 12 || op == 13) && cig[x] >> 4 < 3) cig[x] = cig[x] >> 4 << 4 | 11;
	}
	if (lane == 0) wa.n_cigar[tid] = n;
}
__global__ __launch_bounds__(64) void k_walk(WalkArgs wa)
{
	extern __shared__ __attribute__((aligned(16))) uint32_t lds_raw[];
	if ((int32_t)blockIdx.x >= wa.n_list) return;
	const int32_t tid = wa.list[blockIdx.x];
	const DTask t = wa.ga.tasks[tid];
	const int lane = (int)threadIdx.x;
	switch (t.pad_) {                                            // (class of the call's packed sweep: 8 + {16, 32, 64 lanes per call})
	case 8: walk_call<16>(wa, t, tid, (char*)lds_raw, lane); break;
	case 9: walk_call<32>(wa, t, tid, (char*)lds_raw, lane); break;
	case 10: walk_call<64>(wa, t, tid, (char*)lds_raw, lane); break;
	default: walk_call<64, true>(wa, t, tid, (char*)lds_raw, lane); break;   // 11: 65..128 columns
	}
}

// dense copy of the CIGARs: every call's slot was sized for the worst case, only n_cigar words are real
__global__ __launch_bounds__(64) void k_cigar_gather(const DTask *tasks, const int32_t *list, const int64_t *dst_off, int32_t n_list,
                                                     const int32_t *n_cigar, const uint32_t *src, uint32_t *dst)
{
	MPA_SHORT_KERNEL();
	if ((int32_t)blockIdx.x >= n_list) return;
	const int32_t tid = list[blockIdx.x], n = n_cigar[tid];
	const uint32_t *from = src + tasks[tid].cig_off;
	uint32_t *to = dst + dst_off[blockIdx.x];
	for (int32_t k = threadIdx.x; k < n; k += 64) to[k] = from[k];
}

} // namespace mpa
