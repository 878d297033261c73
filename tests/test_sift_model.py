"""CPU model of k_seed_sift (miniprot_amd/csrc/seed_exec.hip): the range walk with one cursor per occurrence list, the carry of
a range's last block, the bucket sort of a range's keys, the halving of ranges that do not fit the buffer, the rank base and the neighbour tests at segment
boundaries -- step for step as the kernel does them, checked against the definition: sort all anchors of the query by
(block, seed), keep those that have another anchor in the same or an adjacent block (the only ones the pre-chain of
map.c:186-192 can link), rank = index in the full sorted list.  The HIP kernel itself is checked on the GPU against the host
seeding stage (tests/test_seed_gpu.py); this model pins the algorithm both follow."""
import numpy as np
import pytest


class Overflow(Exception):
    pass


def bucket_sort(keys, key_max, nbk_log2=9, bucket_max=512):
    """The kernel's in-LDS sort: buckets by the leading bits of the key range (arrival order inside a bucket is arbitrary: here
    reversed), then each key's place = its bucket's start + the smaller keys of the bucket; the bitonic network (here: sorted)
    takes ranges with a bucket of more than bucket_max keys."""
    sh = max(int(key_max).bit_length() - nbk_log2, 0)
    buckets = [[] for _ in range(1 << nbk_log2)]
    for k in reversed(keys):
        buckets[k >> sh].append(k)
    if max(len(b) for b in buckets) > bucket_max:
        return sorted(keys)
    out = [None] * len(keys)
    start = 0
    for b in buckets:
        for k in b:
            r = start + sum(1 for x in b if x < k)
            assert out[r] is None                                  # keys are unique ((block, list): a list holds a block once)
            out[r] = k
        start += len(b)
    assert out == sorted(keys)
    return out


def sift_segment(lists, lo0, hi0, n_block, cap=2048, carry_max=1024, target=1280):
    """One workgroup of k_seed_sift: blocks [lo0, hi0) of a query whose occurrence lists (ascending block ids) are `lists`.
    Returns (rank base, [(block, list, rank)] of the kept anchors in output order)."""
    nl = len(lists)
    LB = 1
    while (1 << LB) < nl:
        LB += 1
    lmask = (1 << LB) - 1
    w_max = (1 << (32 - LB)) - 2
    n_anchor = sum(len(x) for x in lists)
    cur = [int(np.searchsorted(x, lo0, "left")) if lo0 > 0 else 0 for x in lists]
    pos_base = sum(cur)
    left_nb = any(lo0 > 0 and c > 0 and x[c - 1] == lo0 - 1 for x, c in zip(lists, cur))
    lo, carry, emitted, out = lo0, [], pos_base, []
    prev_blk = lo0 - 1 if left_nb else None
    width = min(max(n_block * target // n_anchor if n_anchor else n_block, 1), w_max)
    while lo < hi0:
        while True:
            hi = lo + width if hi0 - lo > width else hi0
            cur2 = [int(np.searchsorted(x, hi, "left")) for x in lists]
            total = sum(c1 - c0 for c0, c1 in zip(cur, cur2))
            if len(carry) + total <= cap:
                break
            if hi - lo <= 1:
                raise Overflow()
            width = (hi - lo) >> 1
        keys = list(carry)
        for l, (x, c0, c1) in enumerate(zip(lists, cur, cur2)):
            keys += [((int(b) - lo + 1) << LB) | l for b in x[c0:c1]]
        cur = cur2
        n = len(keys)
        keys = bucket_sort(keys, ((hi - lo + 1) << LB) - 1)
        assert all(k < 0xFFFFFFFF for k in keys)
        last = hi == hi0
        right_nb = last and hi0 < n_block and any(c < len(x) and x[c] == hi0 for x, c in zip(lists, cur))
        n_emit = n
        if not last:
            bound = (hi - lo) << LB
            n_emit = int(np.searchsorted(np.array(keys, np.int64), bound, "left"))
        if n - n_emit > carry_max:
            raise Overflow()
        left0 = -(1 << 30) if prev_blk is None else prev_blk - lo + 1
        right_end = (hi0 - lo) + 1 if right_nb else (1 << 40)
        for i in range(n_emit):
            b = keys[i] >> LB
            lf = keys[i - 1] >> LB if i > 0 else left0
            rt = keys[i + 1] >> LB if i + 1 < n else right_end
            if b - lf <= 1 or rt - b <= 1:
                out.append((lo + b - 1, keys[i] & lmask, emitted + i))
        if n_emit > 0:
            prev_blk = lo + (keys[n_emit - 1] >> LB) - 1
        carry = [k & lmask for k in keys[n_emit:]]
        emitted += n_emit
        lo = hi
        if total < target // 2:
            width = w_max if width > w_max // 2 else width * 2
    assert not carry
    return pos_base, out


def brute(lists):
    a = sorted((int(b), l) for l, x in enumerate(lists) for b in x)
    blk = [b for b, _ in a]
    out = []
    for i, (b, l) in enumerate(a):
        if (i > 0 and b - blk[i - 1] <= 1) or (i + 1 < len(a) and blk[i + 1] - b <= 1):
            out.append((b, l, i))
    return out


def random_lists(rng, n_block, nl, kind):
    lists = []
    for l in range(nl):
        n = int(rng.integers(0, 60))
        if kind == "uniform":
            x = rng.integers(0, n_block, n)
        elif kind == "clustered":                              # many lists hit the same few neighbourhoods (a gene family)
            c = rng.choice(rng.integers(0, n_block, 4), n)
            x = np.clip(c + rng.integers(-3, 4, n), 0, n_block - 1)
        else:                                                   # edges of the block space and of the segments
            x = rng.choice(np.r_[0, 1, n_block - 1, n_block - 2, n_block // 2 + np.arange(-2, 3), n_block // 3, n_block // 3 - 1], min(n, 9), replace=False)
        lists.append(np.unique(x).astype(np.int64))             # (an occurrence list holds a block once, index.c:86-90)
    return lists


@pytest.mark.parametrize("kind", ["uniform", "clustered", "edges"])
def test_model_equals_definition(kind):
    rng = np.random.default_rng({"uniform": 1, "clustered": 2, "edges": 3}[kind])
    for it in range(60):
        n_block = int(rng.choice([50, 1000, 40000]))
        nl = int(rng.choice([1, 2, 7, 40, 130]))
        lists = random_lists(rng, n_block, nl, kind)
        if not any(len(x) for x in lists):
            continue
        want = brute(lists)
        for n_seg in (1, 2, 3, 7):
            cap = int(rng.choice([64, 256, 2048]))
            got = []
            below = 0
            for k in range(n_seg):
                lo, hi = n_block * k // n_seg, n_block * (k + 1) // n_seg
                if hi <= lo:
                    continue
                try:
                    base, part = sift_segment(lists, lo, hi, n_block, cap=cap, carry_max=cap // 2, target=max(cap * 5 // 8, 1))
                except Overflow:
                    got = None
                    break
                assert base == sum(int(np.searchsorted(x, lo, "left")) for x in lists)
                # the staging rule: a segment's kept anchors start at its rank base and never reach the next segment's base
                assert base >= below
                below = base + len(part)
                got += part
            if got is None:                                      # only legitimate when one block (+ carry) really exceeds the buffer
                worst = max(np.bincount(np.concatenate(lists)).max(), 1)
                assert 2 * worst > cap // 2, (kind, it, n_seg, cap)
                continue
            assert got == want, (kind, it, n_seg, cap)


def test_model_flags_a_block_that_cannot_fit():
    lists = [np.array([7, 90], np.int64) for _ in range(200)]    # 200 seeds in the same block: more than cap 64 holds
    with pytest.raises(Overflow):
        sift_segment(lists, 0, 100, 100, cap=64, carry_max=32, target=40)
    # with the real sizes it fits
    base, out = sift_segment(lists, 0, 100, 100)
    assert base == 0 and out == brute(lists)
