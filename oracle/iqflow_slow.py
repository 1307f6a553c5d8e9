"""Second witness for mpboot_amd/host/iqflow.cpp: the steps of IQTree::doTreeSearch between two climbs, restated in plain Python
from the reference's text.  TEST INFRASTRUCTURE -- only tests/ may import it.  Parity unpinned (the reference's C++ layer cannot
be built here): two independent restatements of the same lines agree.

  random_nnis       IQTree::doRandomNNIs (iqtree.cpp:1083-1106) + PhyloTree::doOneRandomNNI (phylotree.cpp:3665-3711) +
                    MTree::getInternalBranches (mtree.cpp:797-815)
  perturb_weights   Alignment::createPerturbAlignment (alignment.cpp:1915-1969)
  random_double     SPRNG lcg64 (sprng/lcg64.c:199-268), random_int (tools.cpp:3351-3355)
"""
import math
import sys

M64 = (1 << 64) - 1


class Stream:
    def __init__(self, state):
        self.state = state & M64

    def random_double(self):
        self.state = (self.state * 0x27BB2EE687B0B0FD + 3037000493) & M64
        return float(self.state) * 5.4210108624275222e-20

    def random_int(self, n):
        return int(math.floor(self.random_double() * n))


def _nx(r):
    v, s = divmod(r, 3)
    return 3 * v + (s + 1) % 3


def internal_branches(back, n):
    """getInternalBranches from the root leaf (tip 1): for every neighbour that is not a leaf, first its own subtree, then the
    branch to it -- unless the node itself is a leaf (mtree.cpp:801-813).  A branch = the record on the near side."""
    out = []
    sys.setrecursionlimit(max(sys.getrecursionlimit(), 4 * n + 100))

    def walk(r):                       # r: record of an inner node that faces where we come from
        for c in (_nx(r), _nx(_nx(r))):
            if back[c] // 3 > n:       # the neighbour is an inner node
                walk(back[c])
                out.append(c)

    if back[3] // 3 > n:
        walk(back[3])
    return out


def random_nnis(back, num_nni, state):
    back = [int(x) for x in back]
    n = (len(back) // 3 + 1) // 2
    g = Stream(state)
    branches = internal_branches(back, n)
    assert len(branches) == n - 3
    used = set()
    relists = 0

    def one_nni(ru):
        rv = back[ru]
        g.random_int(1)                # node1's neighbour: random_int(1) == 0 -> the first one (phylotree.cpp:3677-3687)
        g.random_int(1)                # node2's
        sa, sc = _nx(ru), _nx(rv)
        a, c = back[sa], back[sc]
        back[sa], back[c] = c, sa
        back[sc], back[a] = a, sc
        used.add(ru // 3)
        used.add(rv // 3)

    for _ in range(num_nni):
        index = g.random_int(len(branches))
        ru = branches[index]
        if ru // 3 not in used and back[ru] // 3 not in used:
            one_nni(ru)
        else:
            used.clear()
            branches = internal_branches(back, n)
            relists += 1
            one_nni(branches[index])
    return back, g.state, relists


def perturb_weights(weights, informative, percent, add, state):
    g = Stream(state)
    site_pattern = [p for p, w in enumerate(weights) for _ in range(int(w))]
    nsite = len(site_pattern)
    n_informative_sites = sum(int(w) for p, w in enumerate(weights) if informative[p])
    ratchet_nsite = n_informative_sites * percent // 100
    out = [int(w) for w in weights]
    selected = [False] * nsite
    for _ in range(ratchet_nsite):
        while True:
            site_id = g.random_int(nsite)
            if informative[site_pattern[site_id]] and not selected[site_id]:
                break
        selected[site_id] = True
        out[site_pattern[site_id]] += add
    return out, g.state
