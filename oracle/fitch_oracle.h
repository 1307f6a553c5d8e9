/*
 * fitch_oracle.h -- CPU restatement of the reference's Fitch parsimony path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product (libmpfitch.so) never
 * links, loads or calls it.
 *
 * It follows the reference's own algorithm (one vector per node, lazy
 * re-orientation through per-record xPars flags, traversal descriptors, one
 * insertion test at a time) on a flat-array data model of our own, each function
 * citing the reference lines it restates.  Parity status: PINNED -- checked
 * against the reference's PLL parsimony path compiled from its sources
 * (oracle/_ref/pll_ref_driver, see oracle/Makefile) on the fixtures under
 * tests/golden/, see tests/test_oracle_golden.py.
 */
#ifndef FITCH_ORACLE_H
#define FITCH_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_DNA = 0, ORC_AA = 1, ORC_BIN = 2 /* PLL_BINARY_DATA, 2 states */, ORC_GENERIC = 3 /* PLL_GENERIC_32 */ };
enum { ORC_TIE_FIRST = 0,   /* PLL original: strict '<' (pllrepo/src/fastDNAparsimony.c:1224, :1803, :1925) */
       ORC_TIE_RANDOM = 1 };/* mpboot: uniform random among ties (sprparsimony.cpp:2168-2176, :3001-3008, :3306-3311) */

typedef struct orc orc;

orc *orc_create(int n_taxa, int n_patterns, int datatype, const unsigned char *codes /* [n][P] PLL tip codes */,
                const int *weights /* [P] */, int keep_all_sites);
/* weighted (Sankoff) parsimony: same search code, cost-matrix arithmetic (sprparsimony.cpp:477-551, :880-961) */
orc *orc_create_sankoff(int n_taxa, int n_patterns, int datatype, const unsigned char *codes, const int *weights,
                        int keep_all_sites, const unsigned *cost /* [S*S], cost[i*S+j] = i -> j */);
void orc_destroy(orc *o);
int orc_words(const orc *o);                 /* parsimonyLength W */
int orc_states(const orc *o);
int orc_num_informative(const orc *o);
const int *orc_informative(const orc *o);    /* [P] 0/1 */
const uint32_t *orc_node_vector(const orc *o, int node); /* S*W words */
void orc_set_weights(orc *o, const int *weights);        /* re-pack tips (ratchet / bootstrap re-weighting) */
void orc_enable_persite(orc *o, int on);

void orc_set_tree(orc *o, const int *back);  /* [3*(2n-1)] */
void orc_get_tree(const orc *o, int *back);
void orc_reset_nodep(orc *o);
void orc_get_nodep(const orc *o, int *nodep /* [2n] */);
void orc_node_rectifier(orc *o);
unsigned orc_evaluate(orc *o, int rec, int full);
unsigned orc_score_tree(orc *o);             /* nodeRectifier + evaluate(start, full) */
int orc_pattern_scores(orc *o, unsigned short *ptn /* [P] */);  /* returns sum(ptn*weight) */
int orc_site_scores(orc *o, int *site_pars, int nsite);         /* pllComputeSiteParsimony; returns the sum */

void orc_seed_ties(orc *o, int tie_mode, int seed);
/* the 64-bit state of the restated lcg64 tie stream (rng.h), for hand-over tests */
void orc_set_tie_state(orc *o, unsigned long long state);
unsigned long long orc_get_tie_state(const orc *o);
void orc_set_rand_callback(orc *o, double (*fn)(void *), void *arg);
/* the evaluateParsimony(p) at the top of rearrangeParsimony (sprparsimony.cpp:2285; absent from the PLL original):
   -1 = as the tie mode's variant has it, 0 = off, 1 = on */
void orc_set_pre_evaluate(orc *o, int mode);
void orc_set_max_visits(orc *o, long k);   /* test aid: orc_optimize_spr stops behind k prune-node visits (0 = no limit) */

/* trace of the insertion tests performed by the calls below: (q rec, mp), -1/-2 separators */
void orc_trace(orc *o, int on);
int orc_trace_len(const orc *o);
void orc_trace_get(const orc *o, int *q, unsigned *mp);
/* trace of accepted moves: (remove rec, insert rec, score) */
int orc_moves_len(const orc *o);
void orc_moves_get(const orc *o, int *rem, int *ins, unsigned *score);

int orc_rearrange(orc *o, int rec, int mintrav, int maxtrav);   /* one prune node; updates best/insert/remove */
void orc_set_best(orc *o, unsigned best);
unsigned orc_get_best(const orc *o, int *remove_rec, int *insert_rec);
unsigned orc_optimize_spr(orc *o, int mintrav, int maxtrav);     /* pllOptimizeSprParsimony */
unsigned orc_make_tree(orc *o, long seed, int spr_dist, int *perm_out /* [n+1] or NULL */);
/* stepwise addition only; per added taxon: best score and insertion record */
unsigned orc_stepwise(orc *o, long seed, unsigned *best_per_step, int *insert_per_step);

/* UFBoot-MP online bookkeeping: IQTree::saveCurrentTree (iqtree.cpp:3271-3785, default options) as called from
   testInsertParsimony (sprparsimony.cpp:2163-2166) during orc_optimize_spr.  Parity status of this part: UNPINNED
   (the C++ layer of the reference is not buildable from its sources alone, oracle/Makefile); restated line by line,
   and its per-pattern vectors are cross-checked against an independent from-scratch recomputation in the tests. */
void orc_ufboot_attach(orc *o, int B, const unsigned short *samples /* [B][P] boot_samples_pars */, double epsilon);
void orc_ufboot_detach(orc *o);
void orc_ufboot_set_cutoff(orc *o, double logl_cutoff);       /* 0 = none (iqtree.cpp:3343) */
void orc_ufboot_set_ratchet_booking(orc *o, int on);          /* 0 = -no_hclimb1_bb (iqtree.cpp:3280); default 1 */
void orc_ufboot_set_store_trees(orc *o, int on);              /* 1 = -storetrees (iqtree.cpp:3302-3346): a topology met again is not booked twice */
int orc_ufboot_duplicates(const orc *o);                      /* duplication_counter */
void orc_ufboot_set_mulhits(orc *o, int on);                  /* 1 = -mulhits update rule (iqtree.cpp:3498-3540); right after attach */
void orc_ufboot_set_topboot(orc *o, int n_top);               /* -mulhits -topboot N (iqtree.cpp:3542-3585); after set_mulhits */
void orc_ufboot_set_distinct_iter(orc *o, int k);             /* -distinct_iter_top_boot k (iqtree.cpp:3587-3680); without -mulhits */
void orc_ufboot_set_iteration(orc *o, int cur_it);            /* IQTree::curIt */
int orc_ufboot_sample_iters(const orc *o, int sample, int *iters);   /* boot_trees_parsimony_top_iter[sample] */
int orc_ufboot_sample_top(const orc *o, int sample, int *idx, int *rell, int *threshold);   /* boot_trees_parsimony_top[sample]; returns its size */
int orc_ufboot_sample_trees(const orc *o, int sample, int *out, int cap);   /* boot_trees_parsimony[sample] (insertion order); returns its size */
int orc_ufboot_ntrees(const orc *o);                          /* treels_logl.size() */
int orc_ufboot_bad(const orc *o);                             /* # candidates whose pattern-score sum != mp (:3366) */
unsigned long long orc_ufboot_draws(const orc *o);
void orc_ufboot_tree_logl(const orc *o, double *out);
void orc_ufboot_state(const orc *o, double *boot_logl, int *boot_counts, int *boot_trees);
int orc_ufboot_tree(const orc *o, int tree_index, int *back); /* 1 if that tree was accepted by some sample */
/* books of another search chain of the same run: strictly better (sample, tree) offers are taken (iqtree.cpp:3686, :3710-3720) */
int orc_ufboot_adopt(orc *o, int n_upd, const int *sample, const unsigned *score, const int *tree_of, int n_trees, const int *backs,
                     const unsigned *lengths);
double orc_ufboot_next_cutoff(const orc *o, int percent);     /* iqtree.cpp:1662-1676; with -cutoff_from_btrees :1657-1660 */
void orc_ufboot_set_cutoff_from_btrees(orc *o, int on);       /* params->cutoff_from_btrees (tools.cpp:2442): boot_tree_orig_logl kept at every acceptance */
void orc_ufboot_orig_logl(const orc *o, int *out);            /* boot_tree_orig_logl [B] */

void orc_counters(const orc *o, unsigned long long *newviews, unsigned long long *evaluates, unsigned long long *tests);

#ifdef __cplusplus
}
#endif
#endif
