"""A host model of the dynamic tile schedule of the split-role Jacobian kernels (csrc/kernels.hpp jac_split_loop): workgroups with random speeds draw their
fourth and later tiles from counters, three iterations ahead -- one counter over the plain order, or (XR) one counter per XCD range with stealing from
the following ranges.  Whatever the interleaving, every tile must be processed exactly once, a workgroup must stop at its first tile number >= ntiles,
and the counters must be back at zero when the last workgroup has left.  (The kernels themselves are checked on the GPU: big batch == its halves.)"""
import random

import pytest

DYN_STATIC = 3


class Launch:
    def __init__(self, ntiles, nwg, xr):
        self.ntiles, self.nwg, self.xr = ntiles, nwg, xr
        self.next = 0
        self.done = 0
        self.range_ctr = [0] * 8
        self.ntx = (ntiles + 7) // 8
        self.nxw = nwg // 8

    def leave(self):
        self.done += 1
        if self.done == self.nwg:                       # the last workgroup to leave resets every word
            self.next = self.done = 0
            self.range_ctr = [0] * 8


class Workgroup:
    """One workgroup as a generator: yields once per iteration (so that a scheduler can interleave workgroups arbitrarily)."""

    def __init__(self, L, wg):
        self.L, self.wg = L, wg
        self.processed = []
        self.own = wg & 7
        self.victim = self.own

    def rend(self, v):
        return min((v + 1) * self.L.ntx, self.L.ntiles)

    def resolve(self, a):
        L = self.L
        while True:
            t = self.victim * L.ntx + DYN_STATIC * L.nxw + a
            if t < self.rend(self.victim):
                return t
            self.victim = (self.victim + 1) & 7
            if self.victim == self.own:
                self.victim = -1
                return L.ntiles
            a = L.range_ctr[self.victim]; L.range_ctr[self.victim] += 1

    def run(self):
        L = self.L
        if L.xr:
            tile, tstep = self.own * L.ntx + (self.wg >> 3), L.nxw
        else:
            tile, tstep = self.wg, L.nwg
        if tile >= L.ntiles:
            L.leave(); return
        ring = {1: tile + tstep}
        pend, raw, pending = tile + 2 * tstep, 0, False
        it = 0
        while tile < L.ntiles:
            self.processed.append(tile)
            yield                                        # ... the tile's columns; other workgroups run meanwhile ...
            # window between barriers A and B, thread 0
            if L.xr:
                if pending:
                    pend = self.resolve(raw)
                ring[(it + 2) & 3] = pend
                if self.victim >= 0:
                    raw = L.range_ctr[self.victim]; L.range_ctr[self.victim] += 1; pending = True
                else:
                    pend, pending = L.ntiles, False
            else:
                ring[(it + 2) & 3] = pend
                pend = DYN_STATIC * L.nwg + L.next; L.next += 1
            tile = ring[(it + 1) & 3]
            it += 1
        L.leave()


@pytest.mark.parametrize("xr", [False, True])
@pytest.mark.parametrize("ntiles,nwg", [(2048, 256), (2049, 256), (4097, 256), (32768 + 5, 256), (64 * 8 + 3, 8), (1000, 16)])
def test_every_tile_once_under_random_interleavings(ntiles, nwg, xr):
    if xr and ntiles < 24 * nwg:                         # (the kernels use the per-range counters from 128 tiles per workgroup on; the model needs ranges longer than the static head)
        pytest.skip("range shorter than its static head")
    for seed in range(5):
        rnd = random.Random(1000 * seed + ntiles + nwg)
        L = Launch(ntiles, nwg, xr)
        wgs = [Workgroup(L, w) for w in range(nwg)]
        gens = [w.run() for w in wgs]
        speed = [rnd.choice((1, 1, 2, 3, 7)) for _ in range(nwg)]      # slow workgroups get scheduled less often
        live = list(range(nwg))
        while live:
            i = rnd.choice(live)
            if rnd.randrange(speed[i]) != 0:
                continue
            try:
                next(gens[i])
            except StopIteration:
                live.remove(i)
        seen = sorted(t for w in wgs for t in w.processed)
        assert seen == list(range(ntiles)), (ntiles, nwg, xr, seed)
        assert (L.next, L.done, L.range_ctr) == (0, 0, [0] * 8)
        counts = [len(w.processed) for w in wgs]
        assert min(counts) >= min(DYN_STATIC, ntiles // nwg)
