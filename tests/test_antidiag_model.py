"""The CPU model of the anti-diagonal extension sweep (tools/antidiag_model.py mirrors dp_antidiag.hip step for step: delay lines,
prologue mask, running row maximum, per-lane history, slow path) against the oracle: what the prototype kernel must compute, checked
without a GPU.  (The kernel itself is checked against the oracle under -m gpu: tests/test_dp_gpu.py.)"""
import os
import sys
import numpy as np
import refbind
from dpgen import make_task

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import antidiag_model  # noqa: E402


def test_antidiagonal_model_equals_the_oracle(oracle_built):
    rng = np.random.default_rng(7)
    n = 0
    for (xd, eb, coef) in [(100, 5, 0.5), (15, 5, 0.5), (50, 0, 1.0)]:
        P = refbind.DpParams(refbind.mapping_matrix(23), xdrop=xd, end_bonus=eb, ie_coef=coef)
        for it in range(4):
            nt, aa = make_task(rng, al=int(rng.integers(17, 33)), flank=int(rng.choice([1, 40, 200])), p_indel=0.0)
            for flag in (2, 4):
                rec, aas = refbind.ora_prep_rec(nt, aa, P, flag)
                assert tuple(antidiag_model.antidiag(rec, aas, len(aa), P)) == tuple(refbind.ora_nasw(nt, aa, P, flag)[:3]), (xd, it, flag)
                n += 1
    assert n == 24
