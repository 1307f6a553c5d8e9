// mpboot_hooks.h -- what integration/sprparsimony_shim.cpp needs from mpboot's C++ side.
//
// The reference's sprparsimony.cpp reads a handful of IQTree / Params members directly (cited per field).  The shim
// reaches them through this table instead, so that it compiles against pll.h alone (iqtree.h drags in tools.h ->
// <iqtree_config.h>, a CMake-generated header).  A maintainer fills the table once in iqtree.cpp, e.g.
//
//     static double hk_random(void)                { return random_double(); }
//     static int    hk_ratchet(IQTree *t)          { return globalParam->ratchet_iter >= 0 && (t->on_ratchet_hclimb1 || t->on_ratchet_hclimb2); }
//     static int    hk_opt_btree(IQTree *t)        { return t && t->on_opt_btree; }
//     static int    hk_freq(IQTree *t, int ptn)    { return t->aln->at(ptn).frequency; }
//     static double hk_score(IQTree *t)            { return t->curScore; }
//     static const unsigned short *hk_boot(IQTree *t, int b) { return t->boot_samples_pars[b]; }
//     static double hk_cutoff(IQTree *t)           { return t->logl_cutoff; }
//     ...  mpfitch_shim_install(&hooks);           // before the first parsimony call (IQTree::initializePLL)
#pragma once
#include "../include/mpfitch.h"

class IQTree;

struct mpf_mpboot_hooks {
  double (*random_double)(void);                  // tools.cpp:3363; tie-breaks consume mpboot's own SPRNG stream
  // Hand-over of that stream (optional, both or neither).  With these two hooks the shim passes the generator's 64-bit state
  // to the engine before every call and writes it back afterwards (mpf_set_tie_state / mpf_get_tie_state) instead of
  // installing random_double as a per-draw call-back -- the whole sweep loop of pllOptimizeSprParsimony can then run on the
  // device (k_climb).  rng_get_state returns 1 and fills the three words if randstream is SPRNG's lcg64 (RAN_TYPE ==
  // RAN_SPRNG, tools.cpp:3320-3331), 0 otherwise (the shim then falls back to the call-back).  mpboot side, with SPRNG's public
  // pack_sprng / unpack_sprng / free_sprng (sprng/sprng.h:61-62; layout lcg64.c:486-497: gentype string, seven 4-byte big-endian
  // integers -- the last one the prime addend --, then state and multiplier as 8-byte big-endian words):
  //     static int hk_rng_get(uint64_t *st, uint64_t *mul, uint64_t *add) {
  //       char *b; if (pack_sprng(randstream, &b) <= 0) return 0;
  //       const unsigned char *p = (const unsigned char *)b + strlen(b) + 1;
  //       *add = be(p + 24, 4); *st = be(p + 28, 8); *mul = be(p + 36, 8); free(b); return 1; }
  //     static void hk_rng_set(uint64_t st) {
  //       char *b; pack_sprng(randstream, &b); put_be((unsigned char *)b + strlen(b) + 1 + 28, st, 8);
  //       free_sprng(randstream); randstream = unpack_sprng(b); free(b); }
  // (oracle/spr_shim_driver.cpp has exactly this code running against the reference's own SPRNG objects.)
  int (*rng_get_state)(uint64_t *state, uint64_t *multiplier, uint64_t *addend);
  void (*rng_set_state)(uint64_t state);
  int (*ratchet_climb)(IQTree *);                 // sprparsimony.cpp:3249 (re-weighted climb: refresh tr->aliaswgt)
  int (*on_opt_btree)(IQTree *);                  // sprparsimony.cpp:3253
  int (*pattern_frequency)(IQTree *, int ptn);    // sprparsimony.cpp:3026 (_updateInternalPllOnRatchet)
  double (*cur_score)(IQTree *);                  // sprparsimony.cpp:3279 (start-score check); NULL = no check
  int sort_alignment;                             // globalParam->sort_alignment (sprparsimony.cpp:2462, :3383)
  int gbo_replicates;                             // globalParam->gbo_replicates (sprparsimony.cpp:3245); 0 = no -bb
  // -bb only (unused when gbo_replicates == 0)
  const unsigned short *(*boot_sample)(IQTree *, int b);    // iqtree->boot_samples_pars[b], nptn entries (iqtree.cpp:213-313)
  double ufboot_epsilon;                          // globalParam->ufboot_epsilon (iqtree.cpp:3594)
  double (*logl_cutoff)(IQTree *);                // iqtree->logl_cutoff (iqtree.cpp:3343)
  // globalParam->no_hclimb1_bb (tools.cpp:795, iqtree.cpp:3280): 1 = ratchet climbs run without saveCurrentTree; 0 (mpboot's
  // default) = they are booked too, with the cur_logl of iqtree.cpp:3283-3295 (mpf_ufboot_set_ratchet_booking)
  int no_hclimb1_bb;
  // globalParam->multiple_hits (-mulhits, iqtree.cpp:3498-3540): 1 = every tree that reaches a sample's best REPS joins its
  // boot_trees_parsimony set (mpf_ufboot_set_mulhits; read back with mpf_ufboot_get_sample_trees in ufboot_sync).  The
  int multiple_hits;
  int store_candidate_trees;                      // globalParam->store_candidate_trees (-storetrees, iqtree.cpp:3302-3346): mpf_ufboot_set_store_trees
  int distinct_iter_top_boot;                     // globalParam->distinct_iter_top_boot (iqtree.cpp:3587-3680; without -mulhits)
  int (*cur_iteration)(IQTree *);                 // iqtree->curIt, read before every pllOptimizeSprParsimony (needed by that rule only)
  int store_top_boot_trees;                       // globalParam->store_top_boot_trees (-topboot N, with -mulhits): mpf_ufboot_set_topboot
  // globalParam->cutoff_from_btrees (tools.cpp:2442): boot_tree_orig_logl is read back in ufboot_sync (mpf_ufboot_get_orig_logl) --
  // the main loop then takes its minimum as logl_cutoff itself (iqtree.cpp:1657-1660), or asks mpf_ufboot_next_cutoff
  int cutoff_from_btrees;
  // called at the end of every pllOptimizeSprParsimony with the engine that holds the saveCurrentTree bookkeeping of
  // this climb: copy treels_logl / boot_logl / boot_counts / boot_trees back with mpf_ufboot_* 
  void (*ufboot_sync)(IQTree *, mpf_engine *);
};

void mpfitch_shim_install(const mpf_mpboot_hooks *hooks);
mpf_engine *mpfitch_shim_engine(void);            // the engine behind the shim (NULL before the first call)
