"""Lower-bound helpers of the reference, restated in plain Python -- TEST INFRASTRUCTURE ONLY.

pllCalcMinParsScorePattern (sprparsimony.cpp:2513-2547), ParsTree::findMstScore (parstree.cpp:606-680),
IQTree::doSegmenting (iqtree.cpp:3793-3820), the remain bounds (iqtree.cpp:3842-3853 / sprparsimony.cpp:2813-2819).
Parity status: unpinned (C++ layer of the reference not buildable from its sources alone); written from the reference
text independently of mpboot_amd/host/bounds.cpp and cross-checked in tests/test_bounds.py against properties
(valid lower bounds on actual tree scores, brute-force spanning trees).
"""
import numpy as np

UINT_MAX = 0xFFFFFFFF


def is_unambiguous(code, datatype):
    # :2500-2507 -- the second test repeats PLL_DNA_DATA, so non-DNA data always lands in `return true`
    if datatype == 0:
        return code in (1, 2, 4, 8)
    return True


def calc_min_pars_score_pattern(codes, datatype, site):
    undetermined = 15 if datatype == 0 else 22
    check = [0] * 256
    for j in range(codes.shape[0]):
        check[int(codes[j, site])] = 1
    counter = 0
    for j in range(undetermined):
        if check[j] > 0 and is_unambiguous(j, datatype):
            counter += 1
    return counter - 1


def find_mst_score(states, cost, ptn):
    S = cost.shape[0]
    site_states = [UINT_MAX] * S
    for j in range(states.shape[0]):
        if 0 <= states[j, ptn] < S:
            site_states[int(states[j, ptn])] = 0
    if sum(1 for v in site_states if v == 0) <= 1:
        return 0
    labelled = [UINT_MAX] * S
    added = [False] * S
    count = 0
    while True:
        if count == 0:
            for c in range(S):
                if not added[c] and site_states[c] == 0:
                    labelled[c] = 0
                    break
        min_label, add_node = UINT_MAX, -1
        for c in range(S):
            if not added[c] and site_states[c] == 0 and labelled[c] < min_label:
                min_label, add_node = labelled[c], c
        if add_node >= 0:
            added[add_node] = True
            count += 1
        else:
            break
        for c in range(S):
            if site_states[c] == 0 and not added[c] and labelled[c] > cost[add_node, c]:
                labelled[c] = int(cost[add_node, c])
        if not count < S:
            break
    return sum(labelled[i] for i in range(S) if site_states[i] == 0)


def do_segmenting(ras_pars_score, frequency, n_informative, vcsize=16):
    upper, seg_sum = [], 0
    for i in range(len(ras_pars_score)):
        seg_sum += int(ras_pars_score[i]) * int(frequency[i])
        if (i + 1) % vcsize == 0 and seg_sum > 65535 // 16:
            upper.append(i + 1)
            seg_sum = 0
    if seg_sum:
        upper.append(n_informative)
    return upper


def remain_bounds(segment_upper, min_unit_pars, weight):
    n = len(min_unit_pars)
    return [sum(int(min_unit_pars[pos]) * int(weight[pos]) for pos in range(segment_upper[s], n))
            for s in range(len(segment_upper) - 1)]
