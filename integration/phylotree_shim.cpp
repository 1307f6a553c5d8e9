// phylotree_shim.cpp -- the IQ-TREE-level drop-in: PhyloTree::computeParsimony() (reference phylotree.cpp:1049-1061,
// virtual, phylotree.h:432) and its ParsTree override (parstree.cpp:101-116, parstree.h:50) on libmpfitch.so.
//
// Both classes stay opaque: the two members are DEFINED here under their Itanium-mangled names
//     _ZN9PhyloTree16computeParsimonyEv        int PhyloTree::computeParsimony()
//     _ZN8ParsTree16computeParsimonyEv         int ParsTree::computeParsimony()
// (an extern "C" function whose symbol IS the member's; `this` arrives as the first argument), so a mpboot build that
// drops the two bodies from phylotree.cpp / parstree.cpp and links this file + -lmpfitch gets every call site
// (iqtree.cpp:1772, :2143, :2558-2958; phyloanalysis.cpp:1309, :1737, :1865, :2790, :2811; phylotree.cpp:1704, :3258)
// and every virtual dispatch served by the engine.  Members are reached through integration/phylotree_hooks.h.
//
// Contract kept: returns the int score; side effect: _pattern_pars[0 .. nptn) = per-pattern
// lengths of the tree (phylotree.cpp:986-987), the VCSIZE_USHORT = 16 entries behind them zero (:956-957).  Patterns
// are NOT filtered: the IQ-TREE kernel scores every pattern x frequency, uninformative ones included (the engine is
// created with keep_all_sites = 1).  ParsTree with a cost matrix runs the weighted engine (mpf_engine_create_sankoff).
//
// Built and exercised without the reference: oracle/Makefile `shims` links it with oracle/phylotree_shim_driver.cpp (a
// stand-in class hierarchy that only DECLARES the two members) -> oracle/_build/phylotree_shim_driver, run on the GPU
// by tests/test_gpu_dropin.py.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "phylotree_hooks.h"

namespace {

constexpr int kVcsizeUshort = 16;          // VCSIZE_USHORT (phyloanalysis.h / vectorclass: 16 unsigned shorts per AVX vector)

mpf_phylotree_hooks g_h;
bool g_installed = false;

struct Cached {
  mpf_engine *eng = nullptr;
  const void *aln = nullptr;
  unsigned long long stamp = 0;            // alignment_stamp at the last content check (when the hook exists)
  unsigned long long hash = 0;             // of the pattern states the engine was built from
  int n = 0, P = 0, protein = 0;
  std::vector<int32_t> freq;               // pattern frequencies in force in the engine
  std::vector<uint32_t> cost;              // empty = Fitch
};
Cached g_fitch, g_cost;                    // PhyloTree / unit-cost ParsTree, and ParsTree with a matrix

[[noreturn]] void die(const char *what)
{
  std::fprintf(stderr, "mpfitch phylotree shim: %s: %s\n", what, mpf_last_error());
  std::exit(EXIT_FAILURE);                 // the reference's outError() (tools.cpp) also exits
}

void drop(Cached &c)
{
  if (c.eng) mpf_engine_destroy(c.eng);
  c = Cached();
}

mpf_engine *engine_for(PhyloTree *t, const unsigned int *cost)
{
  if (!g_installed) { std::fprintf(stderr, "mpfitch phylotree shim: mpfitch_phylotree_install() was not called\n"); std::exit(EXIT_FAILURE); }
  const int n = g_h.n_taxa(t), P = g_h.n_patterns(t), protein = g_h.is_protein(t) ? 1 : 0;
  const int S = protein ? 20 : 4;
  Cached &c = cost ? g_cost : g_fitch;
  const bool same_cost = !cost || (c.cost.size() == (size_t)S * (size_t)S && std::memcmp(c.cost.data(), cost, c.cost.size() * sizeof(uint32_t)) == 0);
  const bool same_shape = c.eng && c.n == n && c.P == P && c.protein == protein && same_cost;
  // the host vouches for the content: same object, same stamp
  if (same_shape && g_h.alignment_stamp && c.aln == g_h.alignment_id(t) && c.stamp == g_h.alignment_stamp(t)) return c.eng;
  // otherwise the content decides (a pointer is no identity: `delete aln; new Alignment` returns the same address):
  // the alignment as the IQ-TREE side holds it (patterns of convertState codes), frequencies, and a hash of the states
  std::vector<signed char> col((size_t)n);
  std::vector<int8_t> states((size_t)n * (size_t)P);
  std::vector<int32_t> freq((size_t)P);
  unsigned long long hash = 1469598103934665603ull;       // FNV-1a over the patterns, taxon-minor
  for (int p = 0; p < P; p++) {
    int f = 0;
    g_h.pattern(t, p, col.data(), &f);
    freq[(size_t)p] = f;
    for (int i = 0; i < n; i++) {
      states[(size_t)i * (size_t)P + (size_t)p] = (int8_t)col[(size_t)i];
      hash = (hash ^ (unsigned char)col[(size_t)i]) * 1099511628211ull;
    }
  }
  if (same_shape && hash == c.hash) {
    // the same characters: at most the frequencies differ (a re-weighted copy of the alignment)
    if (freq != c.freq) {
      if (mpf_set_weights(c.eng, freq.data())) die("mpf_set_weights");
      c.freq = freq;
    }
    c.aln = g_h.alignment_id(t);
    c.stamp = g_h.alignment_stamp ? g_h.alignment_stamp(t) : 0;
    return c.eng;
  }
  drop(c);
  std::vector<uint8_t> codes(states.size());
  if (mpf_encode_iqtree_states(protein ? MPF_AA : MPF_DNA, states.data(), (int64_t)states.size(), codes.data())) die("mpf_encode_iqtree_states");
  mpf_config cfg;
  std::memset(&cfg, 0, sizeof cfg);
  cfg.n_taxa = n;
  cfg.n_patterns = P;
  cfg.datatype = protein ? MPF_AA : MPF_DNA;
  cfg.keep_all_sites = 1;                  // every pattern counts (phylotree.cpp:758, :985)
  if (cost) {
    c.cost.assign(cost, cost + (size_t)S * (size_t)S);
    if (mpf_engine_create_sankoff(&c.eng, &cfg, codes.data(), freq.data(), c.cost.data())) die("mpf_engine_create_sankoff");
  } else if (mpf_engine_create(&c.eng, &cfg, codes.data(), freq.data())) die("mpf_engine_create");
  c.aln = g_h.alignment_id(t);
  c.stamp = g_h.alignment_stamp ? g_h.alignment_stamp(t) : 0;
  c.hash = hash;
  c.freq = freq;
  c.n = n;
  c.P = P;
  c.protein = protein;
  return c.eng;
}

// IQ-TREE nodes (ids: leaves = taxon ids 0..n-1, inner n..2n-3) -> the engine's record links: node number = id + 1,
// slot = position in neighbors[]
void marshal_tree(PhyloTree *t, int n, std::vector<int32_t> &back)
{
  back.assign(3 * (size_t)(2 * n - 1), -1);
  std::vector<int> nei((size_t)(2 * n - 2) * 3, -1);
  for (int id = 0; id < 2 * n - 2; id++) g_h.neighbors(t, id, &nei[(size_t)id * 3]);
  auto slot_of = [&](int id, int other) {
    const int k = id < n ? 1 : 3;
    for (int s = 0; s < k; s++)
      if (nei[(size_t)id * 3 + (size_t)s] == other) return s;
    std::fprintf(stderr, "mpfitch phylotree shim: nodes %d and %d are not mutual neighbours\n", id, other);
    std::exit(EXIT_FAILURE);
  };
  for (int id = 0; id < 2 * n - 2; id++) {
    const int k = id < n ? 1 : 3;
    for (int s = 0; s < k; s++) {
      const int o = nei[(size_t)id * 3 + (size_t)s];
      if (o < 0 || o >= 2 * n - 2) { std::fprintf(stderr, "mpfitch phylotree shim: node %d has no neighbour %d (multifurcating or rooted tree?)\n", id, s); std::exit(EXIT_FAILURE); }
      back[(size_t)(3 * (id + 1) + s)] = 3 * (o + 1) + slot_of(o, id);
    }
  }
}

int compute(PhyloTree *t, const unsigned int *cost)
{
  mpf_engine *e = engine_for(t, cost);
  const int n = g_h.n_taxa(t), P = g_h.n_patterns(t);
  std::vector<int32_t> back;
  marshal_tree(t, n, back);
  unsigned short *pp = g_h.pattern_pars(t, P + kVcsizeUshort);           // phylotree.cpp:1056-1057
  std::memset(pp, 0, sizeof(unsigned short) * (size_t)(P + kVcsizeUshort));   // :957
  uint32_t score = 0;
  // ParsTree::computeParsimony() = computeParsimonyBranch(root->neighbors[0], root) (parstree.cpp:101-116): evaluated at the
  // root leaf's edge, the rest of the tree as the parent side -- the engine's evaluation at that leaf (node number = id + 1)
  const int root = (cost && g_h.root_id) ? g_h.root_id(t) : 0;
  if (root < 0 || root >= n) { std::fprintf(stderr, "mpfitch phylotree shim: root %d is not a leaf id\n", root); std::exit(EXIT_FAILURE); }
  if (mpf_compute_parsimony_at(e, back.data(), root + 1, &score, pp)) die("mpf_compute_parsimony_at");
  return (int)score;
}

}  // namespace

void mpfitch_phylotree_install(const mpf_phylotree_hooks *hooks)
{
  g_h = *hooks;
  g_installed = hooks->n_taxa && hooks->n_patterns && hooks->is_protein && hooks->pattern && hooks->neighbors && hooks->pattern_pars &&
                hooks->alignment_id;
}

void mpfitch_phylotree_release(void)
{
  drop(g_fitch);
  drop(g_cost);
}

extern "C" {

// int PhyloTree::computeParsimony()        (phylotree.cpp:1049-1061)
int _ZN9PhyloTree16computeParsimonyEv(PhyloTree *self) { return compute(self, nullptr); }

// int ParsTree::computeParsimony()         (parstree.cpp:101-116): the Sankoff twin; "-cost fitch" / "-cost e" load unit
// costs (parstree.cpp:42-49), for which the weighted length IS the Fitch length -- those run on the Fitch engine
int _ZN8ParsTree16computeParsimonyEv(PhyloTree *self)
{
  const unsigned int *cost = g_h.cost_matrix ? g_h.cost_matrix(self) : nullptr;
  if (cost) {
    const int S = g_h.is_protein(self) ? 20 : 4;
    bool unit = true;
    for (int i = 0; i < S && unit; i++)
      for (int j = 0; j < S; j++)
        if (cost[i * S + j] != (i == j ? 0u : 1u)) { unit = false; break; }
    if (unit) cost = nullptr;
    // (an asymmetric matrix makes the length depend on the rooting: compute() evaluates where ParsTree does, hook root_id)
  }
  return compute(self, cost);
}

}  // extern "C"
