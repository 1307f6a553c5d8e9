// iqflow.cpp -- the steps of IQTree::doTreeSearch that sit BETWEEN two climbs (iqtree.cpp:1631-1965), device-free.
//
// The reference's search repeats, ≥ 1000 times at a thousand taxa (stop rule, iqtree.cpp:129-130): draw one of the best
// candidate trees (candidateset.cpp:35-45), perturb it with floor(0.5 (n - 3)) random NNIs (iqtree.cpp:1739-1747,
// doRandomNNIs :1083-1106) -- or, every second iteration, re-weight the alignment instead (createPerturbAlignment,
// alignment.cpp:1915-1969; tools.cpp:778-780) --, then climb (pllOptimizeSprParsimony, the engine's part).  All of these steps
// draw from the SAME random_double() stream as the climb's tie rules (random_int = floor(random_double() * n),
// tools.cpp:3351-3355), so a host that hands the stream over by state (mpf_set_tie_state) makes them between two hand-overs.
//
// These helpers are that host logic on the C-ABI's own topology format (`back` records), so that a host without IQ-TREE's tree
// classes -- bench.py, the tests, a C++ driver -- runs the flow the reference runs.  mpboot itself keeps its own code for them;
// nothing here is called from the engine.  What is OURS, because the reference's comes out of Newick round trips through two
// tree libraries that cannot be built here (C++ layer): the ORDER in which the inner branches are listed and which of
// the two NNIs around a branch "the first neighbour at each end" denotes.  The rule set is the reference's: the number of NNIs,
// one index draw per NNI, two (always-zero) neighbour draws per NNI, the used-node rule with its re-listing.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mpfitch.h"
#include "rng.hpp"
#include "ufb_books.hpp"

namespace mpf { void set_error(const std::string &s); }

namespace {

inline int nx(int r) { const int b = (r / 3) * 3, s = r - b; return b + (s == 2 ? 0 : s + 1); }

inline int random_int(mpf::TieRng &g, int n) { return (int)std::floor(g.next() * n); }

// getInternalBranches(root leaf) (mtree.cpp:797-815): depth first from tip 1, the branch to an inner child is listed behind that
// child's own subtree; a branch is named by the record on the parent's side.
void inner_branches(int n, const int32_t *back, std::vector<int32_t> &out, std::vector<int32_t> &work)
{
  out.clear();
  work.clear();
  const int tipmax = 3 * n + 2;
  const int top = back[3];
  if (top <= tipmax) return;
  // work holds (record, state): state 0 = enter node behind `record` (record faces the parent), 1 = emit branch `record`
  work.push_back(top); work.push_back(0);
  while (!work.empty()) {
    const int st = work.back(); work.pop_back();
    const int r = work.back(); work.pop_back();
    if (st == 1) { out.push_back(r); continue; }
    const int c1 = nx(r), c2 = nx(c1);
    // second child is pushed first so that the first child's subtree and branch come out first
    if (back[c2] > tipmax) { work.push_back(c2); work.push_back(1); work.push_back(back[c2]); work.push_back(0); }
    if (back[c1] > tipmax) { work.push_back(c1); work.push_back(1); work.push_back(back[c1]); work.push_back(0); }
  }
}

bool check_tree(int n, const int32_t *back)
{
  if (n < 4 || !back) return false;
  const int nrec = 3 * (2 * n - 1);
  for (int v = 1; v <= 2 * n - 2; v++)
    for (int s = 0; s < (v <= n ? 1 : 3); s++) {
      const int r = 3 * v + s, b = back[r];
      if (b < 3 || b >= nrec || back[b] != r) return false;
    }
  return true;
}

}  // namespace

extern "C" {

// IQTree::doRandomNNIs(numNNI) (iqtree.cpp:1083-1106) with doOneRandomNNI (phylotree.cpp:3665-3711) on `back`, in place.
// *tie_state: the host's random_double() stream (SPRNG lcg64), advanced by the draws the reference makes: one random_int(n - 3)
// per NNI for the branch, two random_int(1) per NNI for the neighbours (random_int(1) is 0 whatever is drawn: the reference always
// takes the first neighbour at either end, :3677-3699).  An NNI whose branch touches a node an earlier NNI of the current list
// used re-lists the branches of the tree as it stands and takes the branch at the SAME index of the new list (:1096-1103).
// n_relists (may be null): how often that happened.
int mpf_iq_random_nnis(int32_t n_taxa, int32_t *back, int32_t num_nni, uint64_t *tie_state, int32_t *n_relists)
{
  if (!check_tree(n_taxa, back) || !tie_state || num_nni < 0) { mpf::set_error("mpf_iq_random_nnis: bad argument"); return MPF_E_INVALID; }
  mpf::TieRng g;
  g.state = *tie_state;
  std::vector<int32_t> list, work;
  std::vector<uint8_t> used((size_t)2 * n_taxa, 0);
  inner_branches(n_taxa, back, list, work);
  const int nb = (int)list.size();
  if (nb != n_taxa - 3) { mpf::set_error("mpf_iq_random_nnis: not a binary tree"); return MPF_E_INVALID; }
  int relists = 0;
  for (int i = 0; i < num_nni; i++) {
    const int idx = random_int(g, nb);
    int ru = list[(size_t)idx], rv = back[ru];
    if (used[(size_t)(ru / 3)] || used[(size_t)(rv / 3)]) {
      std::fill(used.begin(), used.end(), 0);
      inner_branches(n_taxa, back, list, work);
      relists++;
      ru = list[(size_t)idx];
      rv = back[ru];
    }
    (void)random_int(g, 1);                         // node1's neighbour: always the first (phylotree.cpp:3677-3687)
    (void)random_int(g, 1);                         // node2's
    const int sa = nx(ru), sc = nx(rv);
    const int a = back[sa], c = back[sc];
    back[sa] = c; back[c] = sa;
    back[sc] = a; back[a] = sc;
    used[(size_t)(ru / 3)] = used[(size_t)(rv / 3)] = 1;
  }
  *tie_state = g.state;
  if (n_relists) *n_relists = relists;
  return MPF_OK;
}

// Alignment::createPerturbAlignment(aln, percent, add, sort) (alignment.cpp:1915-1969) as pattern weights: of the sites of
// informative patterns, n_informative_sites * percent / 100 DISTINCT ones are drawn (random_int(n_sites), redrawn while the site's
// pattern is uninformative or the site was taken, :1947-1951) and each adds `add` copies of its pattern.  Sites are taken in
// pattern order (the reference's site_pattern is in the alignment's column order: another labelling of the same urn).
// informative[p] != 0: the reference tests ras_pars_score != 0; the engine's own filter (mpf_get_informative) is the same set on
// data without ambiguity-only variation (SURVEY hazard 3).
int mpf_iq_perturb_weights(int32_t n_patterns, const int32_t *weights, const uint8_t *informative, int32_t percent, int32_t add,
                           uint64_t *tie_state, int32_t *out)
{
  if (n_patterns < 1 || !weights || !informative || !tie_state || !out || percent < 0 || percent > 100 || add < 0) {
    mpf::set_error("mpf_iq_perturb_weights: bad argument");
    return MPF_E_INVALID;
  }
  std::vector<int32_t> site_ptn;
  int64_t n_inf_sites = 0;
  for (int p = 0; p < n_patterns; p++) {
    if (weights[p] < 0) { mpf::set_error("mpf_iq_perturb_weights: negative weight"); return MPF_E_INVALID; }
    for (int k = 0; k < weights[p]; k++) site_ptn.push_back(p);
    if (informative[p]) n_inf_sites += weights[p];
  }
  const int nsite = (int)site_ptn.size();
  const int64_t want = n_inf_sites * percent / 100;
  std::memcpy(out, weights, (size_t)n_patterns * sizeof(int32_t));
  if (nsite == 0 || want == 0) return MPF_OK;
  mpf::TieRng g;
  g.state = *tie_state;
  std::vector<uint8_t> taken((size_t)nsite, 0);
  for (int64_t s = 0; s < want; s++) {
    int site;
    do site = random_int(g, nsite);
    while (!informative[(size_t)site_ptn[(size_t)site]] || taken[(size_t)site]);
    taken[(size_t)site] = 1;
    out[(size_t)site_ptn[(size_t)site]] += add;
  }
  *tie_state = g.state;
  return MPF_OK;
}

// 128-bit digest of the canonical form of an unrooted topology (the key of CandidateSet::topologies, candidateset.cpp:110-150,
// where the reference keeps the sorted Newick string): two trees get the same key iff they are the same topology (up to hash
// collisions of two independent 64-bit mixes over ≥ 2n 16-bit words).
int mpf_iq_topology_key(int32_t n_taxa, const int32_t *back, uint64_t key[2])
{
  if (!check_tree(n_taxa, back) || !key) { mpf::set_error("mpf_iq_topology_key: bad argument"); return MPF_E_INVALID; }
  std::vector<int32_t> bk(back, back + 3 * (2 * (size_t)n_taxa - 1));
  std::string canon;
  mpf::books::CanonScratch sc;
  mpf::books::canonical_topology(n_taxa, bk, canon, sc);
  uint64_t h1 = 0xcbf29ce484222325ULL, h2 = 0x9e3779b97f4a7c15ULL;
  for (unsigned char ch : canon) {
    h1 = (h1 ^ ch) * 0x100000001b3ULL;
    h2 = (h2 + ch + 1) * 0xff51afd7ed558ccdULL;
    h2 ^= h2 >> 29;
  }
  key[0] = h1;
  key[1] = h2;
  return MPF_OK;
}

}  // extern "C"
