// dp_device.h -- device-side data layout shared by the DP kernels (dp_kernels.hip) and the executor.
#pragma once
#include <cstdint>
#ifndef __HIPCC__
#define __host__
#define __device__
#endif

// The short kernels around the DP rounds (row records, profiles, backtrack, seeding, scans) share the GPU with round kernels of
// other batches whose waves are long-lived and VALU-bound; instruction issue on a SIMD goes by priority, then by age, so a young
// wave at the default priority gets only the slots the old DP waves leave.  They raise their priority once, at entry: they are
// short, latency-critical (a DP lane or a pipeline stage waits for them) and cost the rounds next to nothing.
#ifdef __HIPCC__
#define MPA_SHORT_KERNEL() __builtin_amdgcn_s_setprio(2)
#endif

namespace mpa {

// One DP call as the kernels see it (built by the executor from mpa_dp_task_t).
struct DTask {
	int64_t nt_off;      // strand-local start of the window on vid
	int64_t q_off;       // offset of aa[0] of this call in the concatenated query buffer
	int64_t rec_off;     // first per-row record (uint32 units) in the record pool
	int64_t prof_off;    // query profile (int16 units) in the profile pool, layout [22][pw]
	int64_t tb_off;      // traceback matrix (uint16 units), layout [nl][ncol]         (global mode only)
	int64_t cig_off;     // CIGAR slot (uint32 units) in the cigar pool               (global mode only)
	int64_t bnd_off;     // column-block boundary scratch (16 B units), nl entries     (global mode, ncol>64)
	int32_t vid, nl, al, flag, io;
	int32_t ncol;        // 8*ceil(al/8): the columns the striped reference computes (padded columns are live)
	int32_t pw;          // profile row width (columns, multiple of the lane-group width)
	int32_t cig_cap;
	int32_t out_idx;     // index into the caller's task array
	int32_t pad_;
};

// Per-row record produced by the prep kernel, one uint32 per window row i:
//   byte0 2*nas[i]      nas = amino acid (aa20 code) of the codon ending at row i, 21 when undefined; stored
//                       times 2 because it is used directly as a byte offset into an int16 LDS profile column
//   byte1 donor[i+1]    (int8)  penalty of an intron whose first base is row i+2
//   byte2 gei[i]        (uint8) gap-extension of D at this row: fs when nas[i] is a stop, else ge
//   byte3 acceptor[i]   (int8)  penalty of an intron whose last base is row i
// The byte positions are chosen so that v_perm_b32 can sign-extend byte1/byte3 into packed int16.
__host__ __device__ static inline uint32_t make_rec(uint32_t nas, int32_t donor_next, uint32_t gei, int32_t acceptor)
{
	return (nas & 0xff) | ((uint32_t)(donor_next & 0xff) << 8) | ((gei & 0xff) << 16) | ((uint32_t)(acceptor & 0xff) << 24);
}

// rows of one call that one workgroup of k_prep_rows turns into records (256 threads, four rows each)
#define MPA_PREP_CHUNK_ROWS 1024

// A wave of the extension kernel works on up to 8 calls at once: lanes are split in 64/G groups of G
// columns, and each 32-bit lane register carries two independent calls in its int16 halves.
struct ExtWave {
	int32_t task[8];     // slot = half * (64/G) + group ; -1 = empty
	int32_t max_nl;      // rows this wave iterates
	int32_t pad_[3];
	int64_t rec_base;    // smallest rec_off of the wave's tasks (record reads are base + 32-bit offset)
	int64_t lite_off;    // checkpointed traceback only (below): the wave's extension-bit words / checkpoints in their pools (dwords)
	int64_t ck_off;
};

// Checkpointed traceback (round 6).  A traceback call of up to 64 columns and many rows is swept by the PACKED row sweep of the
// extension kernel (two calls per lane, ~27 VALU instructions per call and row instead of ~105), which keeps, per cell, only the
// four "extension" bits that the walk follows through deletion and intron runs (bits 5..8 of the reference's traceback word:
// D, A, B, C carried rather than opened, nasw-sse.c:455-485), and the DP state at the top of every block of MPA_TB_BLOCK rows.
// The walk (k_walk) follows those runs on the bits alone and recomputes the full traceback words of a block -- with the exact
// traceback sweep (glob_narrow), restarted from the block's checkpoint -- only when it stands in a block in any other state.
//   bits:        one dword per lane and three rows at lite_off + ((i - 2) / 3) * 64 + lane: row i's nibble at bit 4 * (2 - (i - 2) % 3)
//                of the call's half (call slot s of the wave: half s / (64/G), lanes (s % (64/G)) * G + column); nibble = D | A << 1 | B << 2 | C << 3
//   checkpoints: block k >= 1 (first row 2 + k * MPA_TB_BLOCK) at ck_off + (k - 1) * 9 * 64: nine dwords per lane, packed like the
//                sweep's registers: H of rows i-1, i-2, i-3; D of rows i-1, i-2, i-3; A; B; C
#define MPA_TB_BLOCK 96
#define MPA_LITE_SLOT_SHIFT 8        /* DTask::flag bits 8..11: the call's slot in its wave */

struct ExtOut { int32_t nt_len, aa_len, score, flags; };

// penalty on the extension length (nasw-sse.c:426) as a step function of x = i - 3*al:
// pen(x) = val[k] for x in [x[k], x[k+1]).  x[0] = INT32_MIN, last x = INT32_MAX.
#define MPA_PEN_MAX 64
struct PenTable { int32_t n; int32_t x[MPA_PEN_MAX]; int32_t val[MPA_PEN_MAX]; };

struct GlobWave {
	int32_t task[4];     // one call per group of G lanes; -1 = empty
	int32_t max_nl;
	int32_t pad_[3];
};

// One 256-thread workgroup of k_dp_round: which kind of DP unit(s) its four waves work on.
enum DpUnitKind : int32_t {
	U_EXT16 = 0, U_EXT32, U_EXT64,      // up to four independent extension waves (16 / 32 / 64 lanes per call), descriptors first .. first+count
	U_EXT_W2,                           // up to two two-wave extension groups
	U_EXT_W4,                           // one four-wave extension group
	U_EXT_SPLIT,                        // a quarter (or half) of a 1024- (512-) column pair of calls: blk of n_blk
	U_GLOB16, U_GLOB32, U_GLOB64,       // up to four independent traceback waves
	U_GLOB_MB,                          // ... block-major ones (more than 1024 columns)
	U_GLOB_W2, U_GLOB_W4,               // two two-wave / one four-wave traceback group(s)
	U_LITE16, U_LITE32, U_LITE64,       // up to four independent waves of the checkpointed traceback's packed sweep (ExtWave descriptors)
	U_LITE128,                          // ... one call of 65..128 columns per wave (column c + 64 in the high half of lane c)
	U_EXT128                            // up to four independent extension waves of one 65..128-column call each (same layout)
};
struct DpUnit { int32_t kind, first, count, blk, n_blk, sgroup, xg_first, pad_; };

// The device-resident unit queues of the DP worker pool (k_dp_worker, dp_kernels.hip): one slot per DP lane of the stream
// pipeline (a batch in its DP round), all slots of a device in ONE block of device memory that every lane's workers see.
// A round is armed by writing its arguments and unit list, then `lim` = {generation, number of units}, then `head` =
// {generation, 0}; a workgroup takes unit i of a slot with one fetch-add on head -- its unit if i < n and the generations of
// head and lim agree (an index past the end is nobody's; the head is set afresh at the next arming).  The last unit to finish (remaining reaches 0) stores the generation into the lane's word
// of pinned host memory.  Workers are ordinary workgroups that run units until no slot has any left and then exit: nothing on
// the device ever waits for the host.
#define MPA_DP_SLOTS 16
struct DpSlotCtl {               // two queues: [0] units that take a whole workgroup, [1] one-wave units (taken by single waves)
	unsigned long long head[2];  // {generation << 32 | next unit}
	unsigned long long lim[2];   // {generation << 32 | number of units}
	int32_t remaining;           // units of this round (both queues) not yet finished
	int32_t stamp;               // the pool's arm sequence number when this round was armed
	int32_t *done;               // pinned host word of the lane: receives the generation when the round is complete
	long long *trace;            // (MPA_DP_TRACE) per unit {start, end} in 100-MHz ticks, or nullptr
	char pad2_[8];               // 64 bytes: slots do not share a cache line
};
struct DpPoolCtl {
	int32_t alive;               // admitted workers that have not exited
	int32_t budget;              // admission limit (the first few workgroups of every launch are admitted regardless)
	int32_t arm_seq;             // rounds armed so far on this device
	int32_t acquire_mode;        // what a worker does before it reads a round's inputs (MPA_DP_ACQUIRE, see k_dp_worker)
	int32_t pad_[12];
};

static_assert(sizeof(DpSlotCtl) == 64 && sizeof(DpPoolCtl) == 64, "pool control records are one cache line each");

struct DpConst {         // scalar scoring parameters, passed by value
	int32_t go, ge, fs, xdrop, end_bonus;
	int32_t sp[6];
	int32_t sp_null_bonus;   // splice-score track only
	int32_t wide_ge;         // ge or fs above 255: the records carry a stop flag instead of the row's gap extension (glob_cands<K, true>)
};

struct DevGenome {
	const uint8_t *seq;      // 4-bit packed genome
	const int64_t *ctg_off;  // [n_ctg]
	const int64_t *ctg_len;  // [n_ctg]
	const uint8_t *spsc;     // splice-score track [2][l_seq] or nullptr (--spsc)
	int64_t l_seq;
};

} // namespace mpa
