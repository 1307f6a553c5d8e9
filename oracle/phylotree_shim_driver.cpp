/*
 * phylotree_shim_driver.cpp -- TEST INFRASTRUCTURE ONLY.
 * Plays the part of mpboot's tree classes for integration/phylotree_shim.cpp: a stand-in hierarchy
 *     class PhyloTree { virtual ~PhyloTree(); virtual int computeParsimony(); ... };  class ParsTree : public PhyloTree
 * that only DECLARES computeParsimony() -- the definitions (and therefore the vtable slots) come from the shim, under the
 * members' mangled names, exactly as they would in a mpboot build that drops the bodies at phylotree.cpp:1049-1061 and
 * parstree.cpp:101-116.  The program reads an alignment in IQ-TREE state codes, a tree as neighbour lists and optionally
 * a cost matrix, calls tree->computeParsimony() through the base-class pointer (virtual dispatch) and prints the score and
 * _pattern_pars including its 16-entry tail.
 *
 * input (text, stdin): n P protein(0|1) ; P frequencies ; n rows of P state codes ; 2n-2 rows "k id0 [id1 id2]" ;
 *                      then "cost 0" or "cost 1" + S*S entries ; then "trees T" and T-1 further neighbour tables ;
 *                      then optionally "mutate M" and M times either "freq" + P frequencies or "states" + n rows of P codes:
 *                      the alignment object is changed IN PLACE (same address, same pattern count -- what `delete aln; new
 *                      Alignment` of the reference's bootstrap loop looks like to a pointer-keyed cache) and the last tree scored again
 */
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../integration/phylotree_hooks.h"

class PhyloTree {
 public:
  virtual ~PhyloTree();                       // key function: the vtable lives in this file
  virtual int computeParsimony();             // defined by integration/phylotree_shim.cpp
  int n = 0, P = 0, protein = 0;
  std::vector<int> freq;
  std::vector<signed char> states;            // [n][P]
  std::vector<int> nei;                       // [2n-2][3]
  unsigned short *_pattern_pars = nullptr;
  std::vector<unsigned int> cost_matrix;
};
PhyloTree::~PhyloTree() { std::free(_pattern_pars); }

class ParsTree : public PhyloTree {
 public:
  ~ParsTree() override;
  int computeParsimony() override;            // defined by integration/phylotree_shim.cpp
};
ParsTree::~ParsTree() {}

static int hk_ntaxa(const PhyloTree *t) { return t->n; }
static int hk_nptn(const PhyloTree *t) { return t->P; }
static int hk_prot(const PhyloTree *t) { return t->protein; }
static void hk_pattern(const PhyloTree *t, int p, signed char *st, int *f)
{
  for (int i = 0; i < t->n; i++) st[i] = t->states[(size_t)i * (size_t)t->P + (size_t)p];
  *f = t->freq[(size_t)p];
}
static void hk_nei(const PhyloTree *t, int id, int out[3]) { for (int k = 0; k < 3; k++) out[k] = t->nei[(size_t)id * 3 + (size_t)k]; }
static unsigned short *hk_ptnpars(PhyloTree *t, int len)
{
  if (!t->_pattern_pars) {
    t->_pattern_pars = (unsigned short *)std::malloc(sizeof(unsigned short) * (size_t)len);
    for (int i = 0; i < len; i++) t->_pattern_pars[i] = 0xBEEF;          // the shim must clear all of it
  }
  return t->_pattern_pars;
}
static const unsigned int *hk_cost(const PhyloTree *t) { return t->cost_matrix.empty() ? nullptr : t->cost_matrix.data(); }
static const void *hk_alnid(const PhyloTree *t) { return t->states.data(); }
static int g_root = 0;
static int hk_root(const PhyloTree *) { return g_root; }

static void read_tree(PhyloTree *t)
{
  t->nei.assign((size_t)(2 * t->n - 2) * 3, -1);
  for (int id = 0; id < 2 * t->n - 2; id++) {
    int k = 0;
    if (std::scanf("%d", &k) != 1) std::exit(2);
    for (int j = 0; j < k; j++)
      if (std::scanf("%d", &t->nei[(size_t)id * 3 + (size_t)j]) != 1) std::exit(2);
  }
}

int main()
{
  int n, P, protein;
  if (std::scanf("%d %d %d", &n, &P, &protein) != 3) return 2;
  std::vector<int> freq((size_t)P);
  for (int &f : freq) if (std::scanf("%d", &f) != 1) return 2;
  std::vector<signed char> states((size_t)n * (size_t)P);
  for (auto &s : states) { int v; if (std::scanf("%d", &v) != 1) return 2; s = (signed char)v; }
  std::vector<int> first_nei;
  PhyloTree probe;
  probe.n = n;
  read_tree(&probe);
  char word[16];
  int have_cost = 0;
  if (std::scanf("%15s %d", word, &have_cost) != 2) return 2;
  const int S = protein ? 20 : 4;
  std::vector<unsigned int> cost;
  if (have_cost) { cost.resize((size_t)S * (size_t)S); for (auto &c : cost) if (std::scanf("%u", &c) != 1) return 2; }
  int T = 1;
  if (std::scanf("%15s %d", word, &T) != 2) return 2;
  if (word[0] == 'r') {                                  // "root <leaf id>" in front of "trees T": IQ-TREE's root leaf
    g_root = T;
    if (std::scanf("%15s %d", word, &T) != 2) return 2;
  }

  mpf_phylotree_hooks h{};
  h.n_taxa = hk_ntaxa; h.n_patterns = hk_nptn; h.is_protein = hk_prot; h.pattern = hk_pattern; h.neighbors = hk_nei;
  h.pattern_pars = hk_ptnpars; h.cost_matrix = hk_cost; h.alignment_id = hk_alnid;
  h.root_id = hk_root;
  mpfitch_phylotree_install(&h);

  // mpboot holds its tree as IQTree (a PhyloTree) or, under -cost, as ParsTree; every caller goes through the base pointer
  PhyloTree *tree = have_cost ? new ParsTree : new PhyloTree;
  tree->n = n; tree->P = P; tree->protein = protein; tree->freq = freq; tree->states = states; tree->nei = probe.nei;
  tree->cost_matrix = cost;
  for (int k = 0; k < T; k++) {
    if (k) read_tree(tree);
    const int score = tree->computeParsimony();          // virtual call, resolved to the shim's definition
    std::printf("score %d\npattern_pars", score);
    for (int p = 0; p < P + 16; p++) std::printf(" %u", (unsigned)tree->_pattern_pars[p]);
    std::printf("\n");
  }
  int M = 0;
  if (std::scanf("%15s %d", word, &M) == 2) {
    for (int m = 0; m < M; m++) {
      if (std::scanf("%15s", word) != 1) return 2;
      if (word[0] == 'f') {
        for (int &f : tree->freq) if (std::scanf("%d", &f) != 1) return 2;
      } else {
        for (auto &s : tree->states) { int v; if (std::scanf("%d", &v) != 1) return 2; s = (signed char)v; }
      }
      const int score = tree->computeParsimony();
      std::printf("score %d\npattern_pars", score);
      for (int p = 0; p < P + 16; p++) std::printf(" %u", (unsigned)tree->_pattern_pars[p]);
      std::printf("\n");
    }
  }
  delete tree;
  mpfitch_phylotree_release();
  return 0;
}
