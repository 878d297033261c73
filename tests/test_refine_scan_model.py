"""CPU model of k_refine_scan_map's hit collection (miniprot_amd/csrc/seed_exec.hip), written BEFORE the kernel was changed (the
round-5 attempt at aggregating the hit list's atomics went to the GPU without one and produced a difference nobody could explain).

What the kernel computes (mp_refine_reg, map.c:53-79, with mp_sketch_nt4 / mp_sketch_clean_orf, sketch.c:40-100, as its source of
window k-mers): for every window of a batch, the positions whose k-mer (reduced alphabet, no stop / N in its codons) is one of the
query's k-mers and lies in an open reading frame of at least min_aa_len codons -- as a LIST of (window, position, group) in no
particular order, plus wcnt[window][group] = how many positions of the window carry the group's k-mer (n1 of map.c:66).

Round 6 changes how the list is filled: the threads of a workgroup (a few consecutive chunks of a window) append to a buffer in LDS with a local
counter; behind a barrier the workgroup reserves its hits' places in the global list with ONE fetch-add and copies them; a hit that
finds the LDS buffer full takes a global place by itself, as before.  The model replays that under random thread orders, small
buffers (so that the overflow path runs) and a global capacity that is exceeded, and checks it against the definition:
  * the list holds every hit of the definition exactly once while the capacity suffices;
  * the counter always ends at the true number of hits, also when the list overflows (the host then refuses the batch);
  * no place of the list is written twice; wcnt equals the definition's counts."""
import numpy as np
import pytest

STOP = 0xFF


def codon_table(rng):
    """a reduced alphabet of 13 letters over 64 codons with three stops, like ns_tab_codon13"""
    t = rng.integers(0, 13, 64).astype(np.int64)
    t[rng.choice(64, 3, replace=False)] = STOP
    return t


def window_hits(bases, tab, kmer, min_aa_len, words):
    """the definition, position by position: k-mer ending at pos (codons ending at pos - 3c), all codons valid, word in the query's
    set, the reading frame around it open for at least min_aa_len codons (the walk stops at the window's ends)"""
    n = len(bases)
    cod = np.full(n, STOP, dtype=np.int64)
    for e in range(2, n):
        b = bases[e - 2:e + 1]
        if (b <= 3).all():
            cod[e] = tab[b[0] << 4 | b[1] << 2 | b[2]]
    out = []
    mask = (1 << (4 * kmer)) - 1
    for pos in range(n):
        if pos - 3 * (kmer - 1) < 2:
            continue
        cs = [cod[pos - 3 * c] for c in range(kmer - 1, -1, -1)]
        if any(c == STOP for c in cs):
            continue
        w = 0
        for c in cs:
            w = (w << 4 | int(c))
        w &= mask
        if w not in words:
            continue
        k = kmer
        q = pos - 3 * kmer
        while k < min_aa_len and q >= 2 and cod[q] != STOP:
            k += 1
            q -= 3
        q = pos + 3
        while k < min_aa_len and q < n and cod[q] != STOP:
            k += 1
            q += 3
        if k >= min_aa_len:
            out.append((pos, words[w]))
    return out


class GlobalList:
    def __init__(self, cap):
        self.cap, self.n, self.slots = cap, 0, {}

    def fetch_add(self, k):
        at = self.n
        self.n += k
        return at

    def put(self, at, hit):
        if at < self.cap:
            assert at not in self.slots, "a place of the list written twice"
            self.slots[at] = hit


def workgroup(glist, wcnt, win, wg_start, chunk, n_super, win_len, hits_of_window, rng, lcap, n_threads=256):
    """one workgroup of the kernel: up to n_super consecutive chunks of one window; threads in a random order (any interleaving of
    the LDS atomics); the buffer is emptied behind a chunk that leaves it at least half full and behind the last chunk"""
    l_hit, l_n = [None] * lcap, 0
    cstart = wg_start
    while cstart < win_len and cstart < wg_start + n_super * chunk:
        mine = [h for h in hits_of_window if cstart <= h[0] < cstart + chunk]
        # who finds what (round 6, the rolling k-mer word): thread 3q + r takes the positions 3 PER q + r, + 3, ... of the chunk, one
        # reading frame of a stretch of 3 PER bases, in PER steps; thread 255 the chunk's last PER positions
        per = chunk // n_threads
        by_step = {}
        covered = set()
        for pos, grp in mine:
            o = pos - cstart
            if o < 3 * per * 85:
                th, step = 3 * (o // (3 * per)) + o % 3, (o % (3 * per)) // 3
            else:
                th, step = 255, o - 3 * per * 85
            assert 0 <= th < n_threads and 0 <= step < per and (th, step) not in covered
            covered.add((th, step))
            by_step.setdefault(step, []).append((th, pos, grp))
        for step in range(per):                                         # the threads run their step-th position in any order
            todo = by_step.get(step, [])
            for k in rng.permutation(len(todo)):
                th, pos, grp = todo[int(k)]
                li = l_n
                l_n += 1                                                # atomicAdd on the LDS counter
                if li < lcap:
                    l_hit[li] = (win, pos, grp)
                else:                                                   # the buffer is full: a place of its own, as in round 5
                    glist.put(glist.fetch_add(1), (win, pos, grp))
                wcnt[(win, grp)] = wcnt.get((win, grp), 0) + 1
        last = cstart + chunk >= win_len or cstart + chunk >= wg_start + n_super * chunk
        ln = min(l_n, lcap) if (l_n >= lcap // 2 or last) else 0       # behind the barrier
        if ln:
            base = glist.fetch_add(ln)                                  # thread 0
            for k in range(ln):
                glist.put(base + k, l_hit[k])
            l_n = 0
        cstart += chunk
    assert l_n == 0


@pytest.mark.parametrize("seed,lcap,cap_frac,n_super", [(1, 384, 2.0, 4), (2, 4, 2.0, 4), (3, 1, 2.0, 1), (4, 8, 0.5, 3), (5, 192, 0.9, 4), (6, 16, 2.0, 8)])
def test_workgroup_hit_collection_equals_the_definition(seed, lcap, cap_frac, n_super):
    rng = np.random.default_rng(seed)
    tab = codon_table(rng)
    kmer, min_aa_len, chunk = 5, 12, 512
    # a query's k-mer groups: a few hundred distinct words, some of them planted densely in the windows
    words = {}
    while len(words) < 300:
        w = 0
        for _ in range(kmer):
            w = w << 4 | int(rng.integers(0, 13))
        words.setdefault(w, len(words))
    windows = []
    for _ in range(6):
        n = int(rng.integers(700, 2600))
        b = rng.integers(0, 4, n).astype(np.int64)
        b[rng.random(n) < 0.003] = 4                                   # N
        # plant stretches that spell the query's words in one frame (exon-like: many hits close together)
        rev = {}
        for c in range(64):
            if tab[c] != STOP:
                rev.setdefault(int(tab[c]), []).append(c)
        for _ in range(4):
            at = int(rng.integers(10, n - 200))
            for w in rng.choice(list(words), 8):
                for c in range(kmer - 1, -1, -1):
                    letter = (int(w) >> (4 * c)) & 15
                    if letter in rev and at + 3 <= n:
                        cc = rev[letter][0]
                        b[at:at + 3] = [cc >> 4, cc >> 2 & 3, cc & 3]
                        at += 3
        windows.append(b)
    want, want_cnt = [], {}
    per_window = []
    for wi, b in enumerate(windows):
        hs = window_hits(b, tab, kmer, min_aa_len, words)
        per_window.append(hs)
        for pos, grp in hs:
            want.append((wi, pos, grp))
            want_cnt[(wi, grp)] = want_cnt.get((wi, grp), 0) + 1
    assert len(want) > 50
    glist, wcnt = GlobalList(int(len(want) * cap_frac)), {}
    chunks = [(wi, st) for wi, b in enumerate(windows) for st in range(0, len(b), chunk * n_super)]
    for k in rng.permutation(len(chunks)):                              # workgroups run in any order
        wi, st = chunks[int(k)]
        workgroup(glist, wcnt, wi, st, chunk, n_super, len(windows[wi]), per_window[wi], rng, lcap)
    assert glist.n == len(want)                                         # the counter is the true number of hits, whatever fitted
    assert wcnt == want_cnt
    if glist.cap >= len(want):
        assert sorted(glist.slots.values()) == sorted(want) and sorted(glist.slots) == list(range(len(want)))
    else:
        assert len(glist.slots) == glist.cap and set(glist.slots.values()) <= set(want)
