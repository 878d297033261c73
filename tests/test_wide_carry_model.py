"""CPU model of how a wave of ext_wide_body (miniprot_amd/csrc/dp_kernels.hip, MPA_ROW_TAIL_W) takes over from the wave to its left.

A wave owns 64 columns of a call that is wider than that.  For its row it needs, from the wave that owns the 64 columns before:
  * carry = the running maximum of (h + j * ge) over ALL columns to the left of its first (the gap chain of nasw-sse.c:408-421 as a
    max-plus scan), which every one of its lanes must see, and which it has to extend for the wave to its right;
  * left_h = H of the column before its first, which only its first lane needs (the diagonal / frameshift terms of later rows).
The asm row lets both enter through LANE 0 only:
  * x'[0] = max(x[0], carry), then the wave's ordinary INCLUSIVE prefix maximum y' = scan(x');
  * the exclusive value of lane j >= 1 is y'[j - 1] (a wave_shr:1 move), of lane 0 the carry itself (the move leaves lane 0 of its
    destination untouched, and the destination was preset with the carry);
  * lane 63 of y' is the carry for the next wave;
  * H shifted by one lane is a wave_shr:1 move into a register preset with left_h.
The model checks that against the definition over saturating int16 values, both int16 halves of a lane at once, including carries and
inputs at the -32768 the kernels use for "no value", and chains several waves the way a 1 024-column group does."""
import numpy as np
import pytest

NEG = -32768


def definition(x_all, h_all):
    """over the whole row (all waves' columns side by side): exclusive running maximum of x, and H shifted by one column"""
    n = len(x_all)
    ex = np.full(n, NEG, dtype=np.int64)
    run = NEG
    for j in range(n):
        ex[j] = run
        run = max(run, int(x_all[j]))
    hs = np.concatenate(([NEG], h_all[:-1]))
    return ex, hs


def wave(x, h, carry, left_h):
    """one wave of 64 lanes as the asm row does it"""
    xp = x.copy()
    xp[0] = max(int(xp[0]), carry)                            # v_pk_max_i16 x, x, cx   (cx: lane 0 = carry, the others -inf)
    y = np.maximum.accumulate(xp)                             # the DPP scan (row_shr 1, 2, 4, 8, row_bcast 15, 31)
    ke = np.empty(64, dtype=np.int64)
    ke[:] = carry                                             # v_mov_b32 ke, cx: lane 0 keeps it ...
    ke[1:] = y[:-1]                                           # ... v_mov_b32_dpp ke, x wave_shr:1 writes lanes 1..63
    hs = np.empty(64, dtype=np.int64)
    hs[:] = left_h                                            # v_mov_b32 hs3, ch
    hs[1:] = h[:-1]                                           # v_mov_b32_dpp hs3, h wave_shr:1
    return ke, hs, int(y[63]), int(h[63])                     # lane 63 writes {scan, H} for the wave to the right


@pytest.mark.parametrize("seed,n_waves", [(1, 2), (2, 4), (3, 8), (4, 16), (5, 16)])
def test_carry_through_lane_zero_equals_the_row_wide_scan(seed, n_waves):
    rng = np.random.default_rng(seed)
    for _ in range(200):
        n = 64 * n_waves
        x = rng.integers(-3000, 3000, n).astype(np.int64)
        h = rng.integers(-3000, 3000, n).astype(np.int64)
        # stretches of "no value", a whole wave of it, the extremes
        for _k in range(3):
            a = int(rng.integers(0, n))
            x[a:a + int(rng.integers(1, 100))] = NEG
        if seed >= 4:
            w = int(rng.integers(0, n_waves))
            x[64 * w:64 * w + 64] = NEG
            x[int(rng.integers(0, n))] = 32767
        want_ex, want_hs = definition(x, h)
        carry, left_h = NEG, NEG                              # the first wave of a call: nothing to its left
        for w in range(n_waves):
            ke, hs, carry, left_h = wave(x[64 * w:64 * w + 64], h[64 * w:64 * w + 64], carry, left_h)
            assert (ke == want_ex[64 * w:64 * w + 64]).all()
            assert (hs == want_hs[64 * w:64 * w + 64]).all()
        assert carry == max(NEG, int(x.max()))


def test_exchange_slot_addresses_never_collide():
    """the LDS exchange of a workgroup: slot (parity, wave, row of the step) = parity * 16 + 3 * wave + row, 8 bytes each; wave w reads, in
    a step of parity p, what wave w - 1 wrote in its previous step, which had parity p ^ 1 for it too (wave w runs one step behind):
    no slot is written while it can still be read"""
    NW = 4
    for step in range(2, 40):                                 # global step number; wave w is at its own step (step - w)
        writes = {}
        for w in range(NW):
            own = step - w
            if own < 0:
                continue
            par = (own + w) & 1                               # = step & 1: "step parity = (step + w) & 1" counts the wave's own steps
            for k in range(3):
                writes[(par * 16 + 3 * w + k)] = (w, own)
        for w in range(1, NW):
            own = step - w
            if own < 0:
                continue
            par = (own + w) & 1
            for k in range(3):
                slot = (par ^ 1) * 16 + 3 * (w - 1) + k       # what the left neighbour wrote one global step ago
                assert slot not in writes, "a slot is rewritten in the step in which its reader reads it"
