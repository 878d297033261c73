// host_core.h -- host-side building blocks of the mapping pipeline (sorting, sketching, chaining,
// region bookkeeping).  These are the stages the GPU does not own yet (SURVEY.md section 8 rows a1-a11,
// a16-a19); they are written from scratch but must reproduce the reference's observable behaviour
// exactly, including the tie order of its unstable radix sort.
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>
#include "mpa_internal.h"
#include "chain_core.h"               // Pair64, ChainParams, ChainView + the extraction / pair sort shared with the device

namespace mpa {

// ---- sorting (ksort.h:109-162 semantics) ---------------------------------------------------------
void sort_u64(uint64_t *beg, uint64_t *end);          // radix_sort_mp64: ties are identical values
void sort_anchors_by_block(std::vector<uint64_t> &a);  // a[] ascending in its low word: stable sort on the high word
void sort_pairs_by_x(Pair64 *beg, Pair64 *end);        // radix_sort_mp128x: tie order is algorithm-defined

// ---- sketching (sketch.c) ---------------------------------------------------------------------------
uint32_t hash32_mask(uint32_t key, uint32_t mask);
void sketch_protein(const char *seq, int32_t len, int32_t kmer, int32_t mod_bit, std::vector<uint64_t> &out);
// sorted=false leaves the k-mers unsorted and with duplicates (for callers that sort a merged list anyway)
void sketch_nt4(const uint8_t *seq, int64_t len, int32_t min_aa_len, int32_t kmer, int32_t mod_bit, int32_t bbit,
                int64_t boff, std::vector<uint64_t> &out, bool sorted = true);

// refinement seeding (map.c:97-131): all (window position<<32 | query position) pairs of equal k-mers, sorted
void refine_seed_pairs(const uint8_t *nt, int64_t len, int32_t min_aa_len, int32_t kmer, const std::vector<uint64_t> &qk_sorted,
                       const uint64_t *filter, int32_t max_ava, std::vector<uint64_t> &hits_scratch, std::vector<uint64_t> &a);

void refine_pairs_from_hits(std::vector<uint64_t> &hits, const std::vector<uint64_t> &qk_sorted, int32_t max_ava, std::vector<uint64_t> &a);

// ---- chaining (chain.c) -----------------------------------------------------------------------------
// a: sorted anchors, rewritten with the anchors of the kept chains (chain by chain, ascending);
// u: one entry per chain, score<<32 | n_anchors
void chain_anchors(const ChainParams &p, std::vector<uint64_t> &a, std::vector<uint64_t> &u, int32_t *f_copy = nullptr, int32_t *pred_copy = nullptr, bool set_only = false);
// a: rewritten with the anchors of the kept chains as an ascending set (the pre-chain: its chains are dissolved right away)
void chain_anchors_set(const ChainParams &p, std::vector<uint64_t> &a);
void chain_extract(const ChainParams &p, const ChainView &v, std::vector<uint64_t> &a_out, std::vector<uint64_t> &u);
// the anchors of the kept chains as an ascending set, without the chain list (what the pre-chain's caller needs)
void chain_extract_set(const ChainParams &p, const ChainView &v, std::vector<uint64_t> &a_out);
// chain_anchors() when the forward pass was done elsewhere (the device, dev_chain_forward): extraction only
void chain_anchors_after_forward(const ChainParams &p, const int32_t *f, const int32_t *pred, std::vector<uint64_t> &a, std::vector<uint64_t> &u);
// the forward pass alone (reference values for the operator test of k_chain_fwd)
void chain_forward(const ChainParams &p, const uint64_t *a, int64_t n, int32_t *f, int32_t *pred);

// ---- regions (miniprot.h:119-145, hit.c) --------------------------------------------------------------
struct Feat {
	int64_t vs = 0, ve = 0;
	int32_t qs = 0, qe = 0;
	int16_t type = 0, phase = 0;
	int32_t n_fs = 0, n_stop = 0, score = 0, n_iden = 0, blen = 0;
	char donor[2] = { 0, 0 }, acceptor[2] = { 0, 0 };
};

struct Region {
	int32_t off = 0, cnt = 0, id = 0, parent = 0, n_sub = 0, subsc = 0;
	int32_t n_exon = 0, chn_sc = 0, chn_sc_ungap = 0;
	uint32_t hash = 0, vid = 0;
	int32_t qs = 0, qe = 0;
	int64_t vs = 0, ve = 0;
	std::vector<uint64_t> a;          // this region's anchors after refinement (window-relative x)
	// alignment (mp_extra_t)
	bool aligned = false;
	int32_t dp_score = 0, dp_max = 0, dp_max2 = 0, blen = 0, n_fs = 0, n_stop = 0, dist_stop = 0, dist_start = 0, n_iden = 0, n_plus = 0;
	std::vector<uint32_t> cigar;
	std::vector<Feat> feat;
};

int32_t chain_score_ungapped(const uint64_t *a, int32_t n, int32_t kmer);                  // hit.c:18-30
void regions_from_chains(const mpa_idx_s *mi, const std::vector<uint64_t> &u, const std::vector<uint64_t> &a, std::vector<Region> &out); // hit.c:32-76
void sort_regions(std::vector<Region> &r);                                                  // hit.c:97-126
void assign_parents(float mask_level, int mask_len, std::vector<Region> &r, int sub_diff);  // hit.c:128-187
void select_secondary(float pri_ratio, int min_diff, int best_n, std::vector<Region> &r);   // hit.c:189-236
void prefer_multi_exon(std::vector<Region> &r, int32_t single_penalty);                     // hit.c:238-250
// hit.c:252-287: per region (left<<32 | right) extension limits; anchors==nullptr means "use genome coordinates"
void extension_limits(const mpa_idx_s *mi_or_null, const std::vector<Region> &r, const std::vector<uint64_t> *first_anchor_x,
                      int32_t min_ext, int32_t max_ext, std::vector<uint64_t> &ext);

} // namespace mpa
