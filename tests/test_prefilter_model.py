"""The arithmetic behind k_seed_prefilter (seed_exec.hip), restated in numpy and checked against plain sorting -- runs without a
GPU.  Claim 1: an anchor can be in a pre-chain run of two or more (consecutive sorted blocks differing by at most one) iff its
block was hit twice or a neighbouring block was hit.  Claim 2: a dropped anchor is alone in its block, so the position of a
kept anchor in the query's FULL stably-sorted list is its rank among the kept anchors plus the number of dropped blocks below
its block -- per range of blocks: dropped blocks of the lower ranges plus a prefix count inside its own range."""
import numpy as np
import pytest

RANGE = 131008                      # PF_PART_BLOCKS of seed_exec.hip


def census_keep(blocks):
    """keep flag per anchor from the two-bit census, evaluated range by range with one halo block per side"""
    keep = np.zeros(len(blocks), bool)
    below = np.zeros(len(blocks), np.int64)          # dropped blocks below the anchor's block, all ranges
    n_part = int(blocks.max()) // RANGE + 1 if len(blocks) else 0
    dropped_per_range = np.zeros(n_part, np.int64)
    local = np.zeros(len(blocks), np.int64)
    for part in range(n_part):
        base = part * RANGE
        inside = (blocks >= base - 1) & (blocks <= base + RANGE)
        cnt = np.bincount(blocks[inside] - (base - 1), minlength=RANGE + 2)
        hit, twice = cnt > 0, cnt > 1
        left = np.concatenate(([False], hit[:-1]))
        right = np.concatenate((hit[1:], [False]))
        k = hit & (twice | left | right)
        dropped = hit & ~k
        dropped[0] = dropped[-1] = False              # halo fields belong to the neighbouring ranges
        dropped_per_range[part] = dropped.sum()
        prefix = np.concatenate(([0], np.cumsum(dropped)[:-1]))
        own = (blocks >= base) & (blocks < base + RANGE)
        f = blocks[own] - (base - 1)
        keep[own] = k[f]
        local[own] = prefix[f]
    before = np.concatenate(([0], np.cumsum(dropped_per_range)[:-1])) if n_part else np.zeros(0, np.int64)
    below = before[blocks // RANGE] + local
    return keep, below


@pytest.mark.parametrize("seed", range(6))
def test_census_rule_and_restored_positions(seed):
    rng = np.random.default_rng(seed)
    n_block = int(rng.choice([5000, RANGE - 3, RANGE + 5, 3 * RANGE + 17]))
    # isolated hits plus clusters, some of them right at range boundaries
    parts = [rng.integers(0, n_block, 4000)]
    for c in list(rng.integers(0, n_block, 60)) + [RANGE - 1, RANGE, 2 * RANGE - 1]:
        if c < n_block:
            parts.append(np.clip(c + rng.integers(-2, 3, int(rng.integers(2, 7))), 0, n_block - 1))
    blocks = np.concatenate(parts).astype(np.int64)
    rng.shuffle(blocks)                                # gather order
    order = np.argsort(blocks, kind="stable")          # the device's stable sort on (query, block)
    sb = blocks[order]
    # reference: runs of consecutive sorted blocks that differ by at most one; anchors in runs of >= 2 can chain
    new_run = np.concatenate(([True], np.diff(sb) > 1))
    run_id = np.cumsum(new_run) - 1
    run_len = np.bincount(run_id)[run_id]
    in_run = np.zeros(len(blocks), bool)
    in_run[order] = run_len >= 2
    keep, below = census_keep(blocks)
    assert np.array_equal(keep, in_run)
    # positions in the full list, restored from the filtered list
    full_pos = np.empty(len(blocks), np.int64)
    full_pos[order] = np.arange(len(blocks))
    kept_idx = np.flatnonzero(keep)
    kept_order = kept_idx[np.argsort(blocks[kept_idx], kind="stable")]
    filtered_rank = np.empty(len(blocks), np.int64)
    filtered_rank[kept_order] = np.arange(len(kept_order))
    assert np.array_equal(filtered_rank[kept_idx] + below[kept_idx], full_pos[kept_idx])
