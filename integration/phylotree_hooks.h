// phylotree_hooks.h -- what integration/phylotree_shim.cpp needs from mpboot's PhyloTree / ParsTree objects.
//
// The shim defines PhyloTree::computeParsimony() and ParsTree::computeParsimony() (phylotree.h:432, parstree.h:50) on
// libmpfitch.so without seeing the class definitions (phylotree.h drags in tools.h -> <iqtree_config.h>, a
// CMake-generated header), so every member it needs is reached through this table.  A maintainer fills it once, e.g. in
// phylotree.cpp:
//
//     static int  hk_ntaxa(const PhyloTree *t)  { return t->aln->getNSeq(); }
//     static int  hk_nptn(const PhyloTree *t)   { return t->aln->size(); }
//     static int  hk_prot(const PhyloTree *t)   { return t->aln->seq_type == SEQ_PROTEIN; }
//     static void hk_pattern(const PhyloTree *t, int p, signed char *st, int *f)
//         { const Pattern &pt = t->aln->at(p); for (size_t i = 0; i < pt.size(); i++) st[i] = pt[i]; *f = pt.frequency; }
//     static int  hk_root(const PhyloTree *t)   { return t->root->id; }
//     static void hk_nei(const PhyloTree *t, int id, int out[3])    // ids of the neighbours of node `id` (leaf: one)
//         { ... NodeVector from getAllNodes / a cached id -> Node* table; out[k] = node->neighbors[k]->node->id ... }
//     static unsigned short *hk_ptnpars(PhyloTree *t, int len)
//         { if (!t->_pattern_pars) t->_pattern_pars = aligned_alloc<BootValTypePars>(len); return t->_pattern_pars; }
//     static const unsigned int *hk_cost(const PhyloTree *t)        // ParsTree only: its cost_matrix, NULL for Fitch
//         { const ParsTree *p = dynamic_cast<const ParsTree *>(t); return p ? p->cost_matrix : NULL; }
//     static const void *hk_alnid(const PhyloTree *t) { return t->aln; }
//     ... mpfitch_phylotree_install(&hooks);
#pragma once
#include "../include/mpfitch.h"

class PhyloTree;

struct mpf_phylotree_hooks {
  int (*n_taxa)(const PhyloTree *);                       // aln->getNSeq()
  int (*n_patterns)(const PhyloTree *);                   // aln->size()
  int (*is_protein)(const PhyloTree *);                   // aln->seq_type == SEQ_PROTEIN (20 states), else DNA
  // aln->at(ptn): Alignment::convertState codes of every taxon (alignment.cpp:839-916) and the pattern's frequency
  void (*pattern)(const PhyloTree *, int ptn, signed char *states /* [n_taxa] */, int *frequency);
  // topology by node id: leaves carry the taxon id 0..n-1, inner nodes n..2n-3 (Node::id, node.h); out = ids of the
  // neighbours in neighbors[] order (a leaf fills out[0] only)
  void (*neighbors)(const PhyloTree *, int node_id, int out[3]);
  // _pattern_pars (phylotree.h:1368), allocated with room for `len` = nptn + VCSIZE_USHORT entries if still NULL
  // (phylotree.cpp:956, :1057)
  unsigned short *(*pattern_pars)(PhyloTree *, int len);
  // ParsTree: cost_matrix[i * nstates + j] (parstree.h; loaded and triangle-repaired by loadCostMatrixFile,
  // parstree.cpp:31-95); NULL = unit costs (PhyloTree, and ParsTree with "-cost fitch|e")
  const unsigned int *(*cost_matrix)(const PhyloTree *);
  // optional (may be NULL = taxon 0): t->root->id, the leaf ParsTree::computeParsimony() roots the tree at (parstree.cpp:101-116).
  // Only an ASYMMETRIC cost matrix makes the length depend on it.
  int (*root_id)(const PhyloTree *);
  // identity of the alignment the tree currently holds (t->aln).  NOT sufficient as a cache key: optimizeBootTrees does
  // `bootstrap_aln = new Alignment; ...; delete aln;` once per sample (iqtree.cpp:2519/2863, :2940/2977) and the allocator
  // hands the same address back, often with the same pattern count -- so the shim compares CONTENT: every call re-reads the
  // patterns, hashes the states and compares the frequencies (states changed -> engine rebuilt; only frequencies changed ->
  // mpf_set_weights).  That is P hook calls and n x P bytes per computeParsimony().
  const void *(*alignment_id)(const PhyloTree *);
  // optional (may be NULL): a number the host changes whenever the alignment's CONTENT changes (e.g. a counter bumped in
  // Alignment's constructors / setAlignment()).  When present and unchanged, with the same alignment_id, the shim skips
  // the re-read above.
  unsigned long long (*alignment_stamp)(const PhyloTree *);
};

void mpfitch_phylotree_install(const mpf_phylotree_hooks *hooks);
void mpfitch_phylotree_release(void);                     // frees the engines (end of the run / new alignment set)
