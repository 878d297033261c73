"""The queue protocol of the DP worker pool (dp_kernels.hip: k_dp_arm, dp_pull, dp_unit_done) as a step-by-step model: several
workers and one lane per slot take turns at ATOMIC steps in random order -- a worker's look (two loads), its fetch-add on
{generation, index}, its second load of the limit, the unit itself, the decrement of `remaining`; the lane's arming in its four
stores (arguments, limit, head) -- and every interleaving must hand out every unit of every round exactly once, with the arguments
of its own round, never a unit of a round that is being armed, and complete a round only when all its units have finished.
(What the model cannot show -- memory visibility -- is covered on the device: tools/ubench/stale.hip, tests/test_dp_pool_gpu.py.)"""
import random

import pytest


class Slot:
    def __init__(self):
        self.head = (0, 0)           # {generation, next unit}
        self.lim = (0, 0)            # {generation, number of units}
        self.remaining = 0
        self.args = None             # which round the arguments in the slot belong to
        self.done = 0                # the lane's pinned word


def worker(slots, home, log, rng):
    """one worker workgroup: units from the home slot first, then the others; leaves when no slot has one (generator: one atomic step per yield)"""
    while True:
        got = None
        for k in range(len(slots)):
            s = (home + k) % len(slots)
            c = slots[s]
            while True:
                h = c.head
                yield
                lim = c.lim
                yield
                if h[0] != lim[0] or h[1] >= lim[1]:
                    break                                          # being armed, or no unit left
                h2 = c.head
                c.head = (h2[0], h2[1] + 1)                        # the fetch-add: returns the old word
                yield
                l2 = c.lim
                yield
                if h2[0] == l2[0] and h2[1] < l2[1]:
                    got = (s, h2[0], h2[1])
                    break
            if got:
                break
        if not got:
            return
        s, gen, unit = got
        c = slots[s]
        args = c.args                                               # (the device reads them behind an acquire)
        for _ in range(rng.randint(0, 3)):                          # the unit runs for a while
            yield
        log.append((s, gen, unit, args))
        c.remaining -= 1                                            # fetch-add(-1): the last one tells the lane
        if c.remaining == 0:
            c.done = gen
        yield


def lane(slots, s, rounds, rng, spawn):
    """a DP lane: arms its slot for one round after the other (a round's units must all have finished before the next is armed)"""
    c = slots[s]
    for gen in range(1, rounds + 1):
        n = rng.randint(0, 12)
        if n == 0:
            continue                                                # (a round without units launches nothing)
        c.remaining = n
        c.args = (s, gen)
        yield
        c.lim = (gen, n)
        yield
        c.head = (gen, 0)
        yield
        spawn(s, min(n, 4))                                          # the round's own workers, behind the arming in stream order
        c.expected = getattr(c, "expected", [])
        c.expected.append((gen, n))
        while c.done != gen:
            yield


@pytest.mark.parametrize("seed", range(40))
def test_every_unit_of_every_round_runs_exactly_once(seed):
    rng = random.Random(seed)
    n_slots = rng.randint(1, 4)
    slots = [Slot() for _ in range(n_slots)]
    log = []
    actors = []

    def spawn(home, count):
        for _ in range(count):
            actors.append(worker(slots, home, log, rng))

    for s in range(n_slots):
        actors.append(lane(slots, s, rng.randint(1, 6), rng, spawn))
    steps = 0
    while actors:
        steps += 1
        assert steps < 2_000_000, "the model does not terminate: a round never completes"
        a = rng.choice(actors)
        try:
            next(a)
        except StopIteration:
            actors.remove(a)
    want = set()
    for s, c in enumerate(slots):
        for gen, n in getattr(c, "expected", []):
            want |= {(s, gen, u) for u in range(n)}
    got = [(s, gen, u) for s, gen, u, _ in log]
    assert len(got) == len(set(got)), "a unit ran twice"
    assert set(got) == want, "a unit was lost or invented"
    assert all(args == (s, gen) for s, gen, _, args in log), "a unit ran with another round's arguments"
