/*
 * mpfitch.h -- C-ABI of libmpfitch.so, the MI355X-native Fitch parsimony engine
 * that sits behind the parsimony entry points of diepthihoang/mpboot.
 *
 * Plain pointers and sizes only; no C++/torch types cross this boundary.  All
 * state lives in an opaque engine handle that owns its HBM buffers.  Every
 * call returns MPF_OK (0) or a negative MPF_E_* code; mpf_last_error() gives the
 * text.  The library needs a HIP device: mpf_engine_create fails with
 * MPF_E_NO_DEVICE when none is present -- there is no CPU fallback.
 *
 * Conventions shared with the reference (all file:line under the reference tree)
 * ---------------------------------------------------------------------------
 * tip codes   : PLL yVector bytes (pllrepo/src/utils.c:98-137): DNA = 4-bit state
 *               masks 1..15 (15 = gap/N), protein = 0..22 (20 = B, 21 = Z, 22 = gap/X).
 * weights     : tr->aliaswgt, one int per pattern (sprparsimony.cpp:2922-2943).
 * topology    : `back` = int32[3*(2n-1)], record rec = 3*number + slot; number 1..n are
 *               tips (slot 0), n+1..2n-2 inner nodes whose three slots form PLL's
 *               `next` ring (slot -> (slot+1)%3); back[rec] is PLL's `->back`
 *               (pllrepo/src/pll.h:622-701), -1 where unused.
 * tie rule    : MPF_TIE_RANDOM = mpboot (sprparsimony.cpp:2168-2176, :3001-3008,
 *               :3306-3311) drawing random_double(); MPF_TIE_FIRST = strict '<' as in
 *               pllrepo/src/fastDNAparsimony.c:1224, :1803, :1925, on exactly scored candidates -- i.e. the PLL
 *               original's rule with mpboot's evaluate before each scan (sprparsimony.cpp:2285); the original, lacking
 *               that evaluate, can score insertions on not-yet-refreshed vectors and then take another path.
 *
 * Each entry point names the reference interface it replaces.
 */
#ifndef MPFITCH_H
#define MPFITCH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPF_ABI_VERSION 8   /* 2: mpf_stats grew (plan_kernel_ms_total, plan_launches), mpf_get_option; 3: mpf_stats grew (climb_*);
                               4: mpf_set_tie_state / mpf_get_tie_state; 5: mpf_ufboot_refine_sweep; 6: mpf_compute_parsimony_at;
                               8: mpf_iq_* (the search loop's own steps between two climbs), mpf_ufboot_adopt, mpf_optimize_spr_many */

enum {
  MPF_OK = 0,
  MPF_E_NO_DEVICE = -1,   /* no HIP device / HIP runtime failure at start-up            */
  MPF_E_INVALID = -2,     /* bad argument (reference: assert / exit(EXIT_FAILURE))      */
  MPF_E_HIP = -3,         /* a HIP call failed                                          */
  MPF_E_NOMEM = -4,
  MPF_E_STATE = -5,       /* call out of order (e.g. no tree set)                       */
  MPF_E_UNSUPPORTED = -6  /* e.g. states outside {4,20}; reference sprparsimony.cpp:575 */
};

/* data types (PLL: pll.h:238-244; partition types "DNA", protein models, "BIN", "MOR" -- iqtree.cpp:515-530):
     MPF_DNA      4 states, tip codes = 4-bit sets 1..15 (15 = undetermined)
     MPF_AA      20 states, codes 0..19, B = 20, Z = 21, 22 = undetermined
     MPF_BIN      2 states (PLL_BINARY_DATA), codes 1, 2, 3 = undetermined            (reference Fitch case 2, sprparsimony.cpp:679-721)
     MPF_GENERIC 32 states (PLL_GENERIC_32), codes 0..31, 32 = undetermined           (reference `default` case, :824-869)
   The binary alphabet runs on the 4-state kernels and multistate data on the 20-state kernels with the unused state rows
   empty -- Fitch sets never acquire a state no tip has, so lengths, vectors and trajectories are the reference's.  A
   multistate alignment that uses symbols beyond the 20th (K..V) is refused with MPF_E_UNSUPPORTED. */
enum { MPF_DNA = 0, MPF_AA = 1, MPF_BIN = 2, MPF_GENERIC = 3 };
enum { MPF_TIE_FIRST = 0, MPF_TIE_RANDOM = 1 };

typedef struct mpf_engine mpf_engine;

typedef struct mpf_config {
  int32_t device;          /* HIP device ordinal                                         */
  int32_t n_taxa;          /* tr->mxtips                                                 */
  int32_t n_patterns;      /* tr->originalCrunchedLength                                 */
  int32_t datatype;        /* MPF_DNA | MPF_AA                                           */
  int32_t keep_all_sites;  /* 1 = !globalParam->sort_alignment (sprparsimony.cpp:2462)   */
  int32_t reserved[3];
} mpf_config;

typedef struct mpf_stats {
  uint64_t insertion_tests;   /* candidates scored (testInsertParsimony equivalents)     */
  uint64_t newview_ops;       /* directional vectors recomputed                          */
  uint64_t scan_launches;
  uint64_t view_launches;
  uint64_t moves_applied;
  uint64_t algorithmic_bytes; /* 6*S*W*4 per insertion test + 3*S*W*4 per newview        */
  double   last_scan_kernel_ms; /* HIP-event time of the most recent scan launch         */
  double   scan_kernel_ms_total;
  double   view_kernel_ms_total;
  double   host_plan_ms_total;   /* wall time spent building scan programs                  */
  double   host_views_ms_total;  /* wall time of update_views() incl. launches and sync     */
  double   host_scan_ms_total;   /* wall time of run_scans() incl. copies and sync         */
  double   host_sweep_ms_total;  /* wall time inside mpf_spr_sweep_scan                      */
  double   plan_kernel_ms_total; /* HIP-event time of the scan-program planner (k_walk_plan)  */
  uint64_t plan_launches;        /* scan launches that ran as planned programs (k_scan_prog)  */
  uint64_t climb_launches;       /* k_climb launches (device-resident sweep segments)         */
  uint64_t climb_steps;          /* steps (speculative batches of prune nodes) inside them    */
  uint64_t climb_nodes;          /* prune nodes they visited                                  */
  uint64_t climb_moves;          /* moves they accepted                                       */
  double   climb_ms_total;       /* wall time of those launches incl. hand-over               */
} mpf_stats;

const char *mpf_last_error(void);
int mpf_abi_version(void);

/* _allocateParsimonyDataStructures + compressDNA (sprparsimony.cpp:3032-3060, :2828-2973):
   upload codes/weights, drop uninformative sites, bit-pack the tips in HBM. */
int mpf_engine_create(mpf_engine **out, const mpf_config *cfg, const uint8_t *codes /* [n][P] */,
                      const int32_t *weights /* [P] */);
/* Weighted (Sankoff) parsimony, the reference's `-cost <file|e>` mode: same entry points, scores are sums of
   cost-matrix entries.  cost[i*S+j] = cost of i -> j (pllCostMatrix, sprparsimony.cpp:190; loaded and closed under
   the triangle inequality as ParsTree::loadCostMatrixFile does, parstree.cpp:31-95; initializeCostMatrix
   sprparsimony.cpp:159-188).  Kernels: newviewSankoffParsimonyIterativeFastSIMD / evaluateSankoff... (:477-551,
   :880-961), tips as compressSankoffDNA (:2636-2825).  Arithmetic is exact 32-bit (the reference's -short_off
   mode).  A matrix that is not symmetric (the loader accepts any, parstree.cpp:31-95) makes the length of a tree depend
   on where it is rooted: every evaluation is then rooted as the reference roots it -- mpf_score_tree at the start tip's
   edge with the tip as the child, an SPR insertion test at the new node's edge towards the near side of the tested branch
   (testInsertParsimony evaluates p->next->next, :2158), a stepwise-addition test at the new tip's edge (:2993-2997) --, so
   the numbers are the reference's own; mpf_ufboot_attach refuses such an engine.  Serves ParsTree::computeParsimony
   (parstree.cpp:101-116) through mpf_compute_parsimony (symmetric matrices only: IQ-TREE's own Sankoff kernel roots elsewhere). */
int mpf_engine_create_sankoff(mpf_engine **out, const mpf_config *cfg, const uint8_t *codes, const int32_t *weights,
                              const uint32_t *cost /* [S*S] */);
/* _pllFreeParsimonyDataStructures (sprparsimony.cpp:3062-3104) */
void mpf_engine_destroy(mpf_engine *e);

/* _updateInternalPllOnRatchet + re-allocate (sprparsimony.cpp:3022-3029, :3249-3252):
   new pattern weights (ratchet / bootstrap replicate), tips re-packed on the device. */
int mpf_set_weights(mpf_engine *e, const int32_t *weights);

/* packing geometry and test hooks (tips as the reference lays them out, rows of W words) */
int mpf_get_geometry(const mpf_engine *e, int32_t *states, int32_t *words_per_row /* W, multiple of 8 as the
                     reference's parsimonyLength */, int32_t *n_informative, int32_t *words_padded);
int mpf_get_informative(const mpf_engine *e, int32_t *flags /* [P] */);
int mpf_get_tip_vector(mpf_engine *e, int32_t tip /* 1..n */, uint32_t *out /* [S][W] */);

/* pllTreeInitTopologyNewick's result, handed over as record links instead of a Newick string
   (iqtree.cpp:2127-2129). */
int mpf_set_tree(mpf_engine *e, const int32_t *back);
int mpf_get_tree(const mpf_engine *e, int32_t *back);
/* tr->nodep[] as left by earlier calls matters to the reference's visiting order
   (sprparsimony.cpp:2046-2101); this resets it to the state of a fresh PLL instance. */
int mpf_reset_node_order(mpf_engine *e);

/* evaluateParsimony(tr, pr, tr->start, PLL_TRUE) (sprparsimony.cpp:1889-1917, :3277):
   Fitch length of the current tree. */
int mpf_score_tree(mpf_engine *e, uint32_t *score);
/* the same for many topologies in one call (replicates / candidate trees) */
int mpf_score_trees(mpf_engine *e, int32_t n_trees, const int32_t *backs, uint32_t *scores);

/* pllComputePatternParsimony (sprparsimony.cpp:3363-3392): per-pattern Fitch lengths of the
   current tree, ptn_pars[P] (0 for dropped patterns); *total = sum(ptn * weight). */
int mpf_pattern_scores(mpf_engine *e, uint16_t *ptn_pars, int32_t *total);
/* pllComputeSiteParsimony (sprparsimony.cpp:3403-3450): the same lengths per EXPANDED site, i.e. in the packed order
   of the kept patterns with a pattern of weight w repeated w times (the reference's perSitePartialPars row of
   tr->start); entries from the number of expanded sites up to n_sites are 0; *total = their sum. */
int mpf_site_scores(mpf_engine *e, int32_t *site_pars, int32_t n_sites, int32_t *total);

/* int PhyloTree::computeParsimony() (phylotree.cpp:1049-1061; callers precede it with
   initializeAllPartialPars(); clearAllPartialLH(), e.g. iqtree.cpp:2141-2143): Fitch length of the given
   tree from scratch plus the per-pattern lengths the reference leaves in _pattern_pars
   (phylotree.cpp:956-957, :986-987).  `back` may be NULL to use the current tree. */
int mpf_compute_parsimony(mpf_engine *e, const int32_t *back, uint32_t *score, uint16_t *pattern_pars /* [P] or NULL */);
/* The same, evaluated at the edge of leaf `root_taxon` (1-based; 0 = the engine's own start leaf, taxon 1).  Only matters for
   the weighted engine with an ASYMMETRIC cost matrix, where the length of a tree depends on the edge it is rooted at:
   ParsTree::computeParsimony() (parstree.cpp:101-116) roots at IQ-TREE's `root` leaf -- computeParsimonyBranch(root->neighbors[0],
   root), :439-541: min_i( rest[i] + min_j( leaf[j] + cost[i][j] ) ), the rest of the tree as the parent side (rows of the matrix)
   -- which is the orientation of this evaluation (and of evaluateSankoffParsimonyIterativeFastSIMD, sprparsimony.cpp:880-961).
   (ABI 6) */
int mpf_compute_parsimony_at(mpf_engine *e, const int32_t *back, int32_t root_taxon, uint32_t *score, uint16_t *pattern_pars);

/* The IQ-TREE side of the reference stores Alignment::convertState codes (alignment.cpp:839-916):
   DNA 0..3, ambiguity = 4-bit mask + 3, STATE_UNKNOWN 18; protein 0..19, B 20, Z 21, STATE_UNKNOWN 22.
   This maps them to the PLL tip codes the engine takes (the reference does it through a PHYLIP text
   round trip, iqtree.cpp:557-564).  No device needed. */
int mpf_encode_iqtree_states(int32_t datatype, const int8_t *states, int64_t count, uint8_t *codes);

/* random_double() source for MPF_TIE_RANDOM.  Default: our restatement of the SPRNG lcg64
   stream the reference creates in init_random(seed) (tools.cpp:3320-3331).  A host that
   wants to share ITS stream passes a callback (drop-in inside mpboot: random_double). */
int mpf_seed_ties(mpf_engine *e, int32_t tie_mode, int32_t seed);
int mpf_set_rand_callback(mpf_engine *e, double (*fn)(void *), void *arg);
/* Hand-over of the tie stream itself (ABI 4).  random_double() is SPRNG's 64-bit LCG with prime addend
   (sprng/lcg64.c:220, :268: state = state * multiplier + prime, value = state * 2^-64); mpboot creates stream 0 of 1 with
   the default parameter (tools.cpp:3326), i.e. MPF_LCG64_MULTIPLIER / MPF_LCG64_ADDEND (lcg64.c:63, :197;
   primes-lcg64.c:64-68).  A host whose stream has exactly these two constants passes the generator's 64-bit state in
   before a call and takes it back afterwards: the engine then consumes the host's stream draw for draw (ties of
   testInsertParsimony sprparsimony.cpp:2171-2172, of the sweep :3309-3310, of saveCurrentTree iqtree.cpp:3594) WITHOUT a
   call-back per draw -- which is what lets the whole sweep loop run on the device (a call-back keeps it on the host).
   mpf_set_tie_state also removes an installed call-back. */
#define MPF_LCG64_MULTIPLIER 0x27bb2ee687b0b0fdULL
#define MPF_LCG64_ADDEND 3037000493ULL
int mpf_set_tie_state(mpf_engine *e, uint64_t state);
int mpf_get_tie_state(const mpf_engine *e, uint64_t *state);
/* the state of that stream n_draws random_double() calls on, in O(log n_draws) (no engine, no device: pure arithmetic) -- what the
   batched refinement uses to pass over prune-node visits in which nothing but the visit's own accept draw happens */
uint64_t mpf_tie_state_after(uint64_t state, uint64_t n_draws);

/* rearrangeParsimony(tr, pr, p, mintrav, maxtrav) candidates (sprparsimony.cpp:2259-2376):
   every insertion test of prune record `rec`, in the reference's DFS order (p side, then q
   side); q_recs[i] = record q of testInsertParsimony(p, q), mp[i] = tree length after the move.
   *n_p = number of p-side candidates.  Does not change the tree. */
int mpf_spr_scan(mpf_engine *e, int32_t rec, int32_t mintrav, int32_t maxtrav, int32_t cap,
                 int32_t *q_recs, uint32_t *mp, int32_t *n_p, int32_t *n_total);

/* one full sweep of scans over every prune node of the current tree WITHOUT applying moves:
   the throughput primitive bench.py times.  Returns the number of insertion tests and the
   minimum mp seen. */
int mpf_spr_sweep_scan(mpf_engine *e, int32_t mintrav, int32_t maxtrav, uint64_t *n_tests, uint32_t *min_mp);
/* the same sweep, handing back every insertion test's tree length: prune nodes in the order pllOptimizeSprParsimony
   visits them (nodep[1 .. 2n-2] after nodeRectifierPars, sprparsimony.cpp:3298), each node's candidates in the
   reference's order (as mpf_spr_scan).  offsets[i] .. offsets[i+1] = candidates of the i-th prune node (offsets has
   2n-1 entries; may be NULL).  *n_tests is always set; mp is filled only if cap >= *n_tests. */
int mpf_spr_sweep_costs(mpf_engine *e, int32_t mintrav, int32_t maxtrav, uint64_t cap, uint32_t *mp, uint64_t *offsets,
                        uint64_t *n_tests);
/* tr->nodep[1 .. 2n-2] as nodeRectifierPars leaves it (sprparsimony.cpp:2046-2101): the prune records of a sweep, in order */
int mpf_get_node_order(mpf_engine *e, int32_t *recs /* [2n-2] */);

/* int pllOptimizeSprParsimony(tr, pr, mintrav, maxtrav, iqtree) (sprparsimony.cpp:3244-3319):
   SPR hill climb on the current tree until no sweep improves; the tree is modified in place
   (read it back with mpf_get_tree).  *score = final length (tr->bestParsimony); the function's
   own return value in the reference (startMP) equals it. */
int mpf_optimize_spr(mpf_engine *e, int32_t mintrav, int32_t maxtrav, uint32_t *score);
/* pllOptimizeSprParsimony on n INDEPENDENT engines at once (the 100 start trees of a run, phyloanalysis.cpp:1270-1317; the bootstrap
   samples' refinement climbs, iqtree.cpp:2797-2862): each engine with its tree set, its weights, its tie rule and stream, as for
   mpf_optimize_spr -- and each climb makes exactly the moves its own mpf_optimize_spr call would make.  Every climb is one resident
   workgroup of ONE launch (k_climb_many: all its sweeps inside, nodeRectifierPars on the device), fed by the calling thread; a climb
   with more moves than the launch's list holds goes on in a second launch.  Engines the batch cannot take (another
   alignment shape, a tracker attached, the weighted engine, a host random_double() call-back) run their climb alone inside the call.
   final_scores[n_engines]. */
int mpf_optimize_spr_many(mpf_engine **engines, int32_t n_engines, int32_t mintrav, int32_t maxtrav, uint32_t *final_scores);
/* ... one LAUNCH of it (every active climb to its optimum, or to a full move list), for callers with more climbs than engines:
   state[k] in: 0 = engine k takes no part, 1 = a climb STARTS on it now, 2 = its climb goes on; out: 2 = goes on, 0 = done
   (final_scores[k] valid).  A finished engine gets its next tree (and weights, stream) and state 1 before the next round. */
int mpf_optimize_spr_many_round(mpf_engine **engines, int32_t n_engines, int32_t mintrav, int32_t maxtrav, uint8_t *state, uint32_t *final_scores);

/* _pllComputeRandomizedStepwiseAdditionParsimonyTree(tr, pr, sprDist, iqtree)
   (sprparsimony.cpp:3224-3235, :3107-3209): random addition order from PLL randum(seed),
   stepwise addition, then SPR sweeps with radius spr_dist. */
int mpf_make_parsimony_tree(mpf_engine *e, int64_t seed, int32_t spr_dist, uint32_t *score);
/* stepwise addition only, with the per-taxon checkpoints (best length, insertion record) */
int mpf_stepwise_addition(mpf_engine *e, int64_t seed, uint32_t *best_per_step /* [n+1] */,
                          int32_t *insert_per_step /* [n+1] */, uint32_t *score);

/* accepted moves of the last mpf_optimize_spr / mpf_make_parsimony_tree: (remove rec, insert rec, length) */
int mpf_get_moves(const mpf_engine *e, int32_t cap, int32_t *remove_rec, int32_t *insert_rec, uint32_t *score,
                  int32_t *n_moves);

/* Lower-bound helpers of the reference's parsimony path (host arithmetic, no device needed).  The reference uses them
   to leave its CPU loops early (REPS skip iqtree.cpp:3435-3445, Sankoff evaluate sprparsimony.cpp:946-955); the engine
   computes exact full sums, so nothing in this library depends on them.
     mpf_min_pars_score_patterns : pllCalcMinParsScorePattern (sprparsimony.cpp:2513-2547) for every pattern
     mpf_mst_scores              : ParsTree::findMstScore (parstree.cpp:606-680) for every pattern
     mpf_segment_patterns        : IQTree::doSegmenting (iqtree.cpp:3793-3820)
     mpf_remain_bounds           : IQTree::pllComputeRellRemainBound (iqtree.cpp:3842-3853) / pllRemainderLowerBounds
                                   (sprparsimony.cpp:2813-2819) for one weight vector */
/* ParsTree::loadCostMatrixFile (parstree.cpp:31-95): `file_or_keyword` = "fitch" | "e" (unit costs for
   n_states_alignment states) or the path of a text file "<nstates>  nstates x nstates entries"; the matrix is then closed
   under the triangle inequality by the reference's own k-i-j loop (:74-80; *changed = 1 if that altered an entry).
   cost has room for cap_states x cap_states entries.  The result is what mpf_engine_create_sankoff takes (the reference
   copies it into pllCostMatrix, iqtree.cpp:601-615). */
int mpf_cost_matrix_load(const char *file_or_keyword, int32_t n_states_alignment, int32_t cap_states, uint32_t *cost /* [cap*cap] */,
                         int32_t *n_states, int32_t *changed);
int mpf_cost_matrix_triangle_fix(int32_t n_states, uint32_t *cost /* [S*S], in place */, int32_t *changed);
int mpf_min_pars_score_patterns(int32_t datatype, int32_t n_taxa, int32_t n_patterns, const uint8_t *codes /* [n][P] PLL tip codes */,
                                int32_t *min_score /* [P] */);
int mpf_mst_scores(int32_t n_states, const uint32_t *cost /* [S*S] */, int32_t n_taxa, int32_t n_patterns,
                   const int8_t *states /* [n][P] IQ-TREE state codes */, uint32_t *mst /* [P] */);
int mpf_segment_patterns(int32_t n_patterns, int32_t n_informative, int32_t vcsize, const int32_t *ras_pars_score,
                         const int32_t *frequency, int32_t *segment_upper /* [P] */, int32_t *n_segments);
int mpf_remain_bounds(int32_t n_units, int32_t n_segments, const int32_t *segment_upper, const int32_t *min_unit_pars,
                      const uint16_t *weight, int32_t *remain /* [n_segments - 1] */);

/* Online UFBoot-MP bookkeeping -- what IQTree::saveCurrentTree (iqtree.cpp:3271-3785, default options) does when
   testInsertParsimony calls it after EVERY insertion test of pllOptimizeSprParsimony (sprparsimony.cpp:2163-2166,
   perSiteScores = gbo_replicates > 0, :3245).  With a tracker attached, mpf_optimize_spr additionally
     - applies the logl_cutoff filter (:3343) and appends the candidate's score to treels_logl (:3345-3348),
     - computes its REPS against all n_samples weight vectors on the device (ufboot.hip),
     - applies the per-sample update rule (:3684-3731) in the reference's order, drawing its tie-breaks from the same
       random stream as the SPR tie-breaks (mpf_seed_ties / mpf_set_rand_callback).
   samples = boot_samples_pars (iqtree.cpp:213-313), [n_samples][n_patterns] uint16.  epsilon = params->ufboot_epsilon
   (0.5, tools.cpp:725); any value in (0, 1) is equivalent for integer scores, others are MPF_E_UNSUPPORTED.
   Both engines: on the weighted (Sankoff, -cost) engine the per-pattern lengths are those pllComputeSankoffPatternParsimony
   reads (sprparsimony.cpp:3341-3355) -- with a matrix that is not symmetric the current tree is booked at every prune node's
   visit with the length and the per-pattern lengths it has at that node's edge (:2285) --, sample sharding (below) included.  Climbs under other weights than the attach-time
   ones (ratchet iterations) are booked as the reference books them (iqtree.cpp:3283-3295) unless
   mpf_ufboot_set_ratchet_booking(e, 0) (-no_hclimb1_bb, :3280); weights that take an attach-time pattern out of the
   alignment altogether rest the tracker until the attach-time weights are back. */
int mpf_ufboot_attach(mpf_engine *e, int32_t n_samples, const uint16_t *samples, double epsilon);
/* Batched bootstrap refinement -- IQTree::optimizeBootTrees' default branch (iqtree.cpp:2797-2862: per sample modifyPatternFreq
   :2520, the parsimony structures rebuilt, ONE pllOptimizeSprParsimony from the sample's tree :2837) for every attached sample
   whose tree is the engine's CURRENT tree, at once.  Fitch state sets do not depend on pattern weights, only the counts do: the
   first sweep of all those climbs is one masked scan + one mask x weight product on the matrix cores; every sample's sweep
   (testInsertParsimony's tie rule sprparsimony.cpp:2168-2176, the sweep's accept rule :3306-3311) is then replayed from the few
   (insertion test, sample) pairs that reach the sample's running best, with the sample's own tie stream (seeded like
   mpf_seed_ties(e, MPF_TIE_RANDOM, tie_seeds[b]); NULL: seed b).
     scores[b]  = length of the current tree under sample b's weights (what the climb starts from)
     stable[b]  = 1: that sweep accepts no move -- mpf_set_weights(sample b) + mpf_optimize_spr from this tree would return
                  scores[b] and leave the tree as it is; 0: it accepts one (first_move_visit[b] = 1-based position in the sweep's
                  visiting order, mpf_get_node_order) -- the caller runs that sample's climb alone
   All three arrays have n_samples entries of the attach call (a sample-sharded tracker fills the entries of its own samples);
   any may be NULL.  Needs the attach-time weights in force, the random tie rule (any maxtrav, both engines).  The
   tracker's saveCurrentTree bookkeeping is not touched.  (ABI 5) */
int mpf_ufboot_refine_sweep(mpf_engine *e, int32_t maxtrav, const int32_t *tie_seeds, uint32_t *scores, uint8_t *stable,
                            int32_t *first_move_visit);
/* Multi-GPU online phase: the samples are sharded over the GPUs, the search chain is not.  Every rank runs the same
   mpf_optimize_spr calls (same tree, same tie seed) on its own engine, which holds only n_local of the n_samples weight
   vectors (sample_ids[c] = run-wide index of local vector c) and so does 1/n_gpus of the REPS work.  After each scan batch
   the engine hands its (candidate, sample, score) events to `exchange`, which must return the events of ALL ranks (any
   order; an all-gather -- RCCL on the GPU box); every rank then replays the same merged list, so tie draws, accepted
   moves and all bookkeeping arrays are identical on every rank and identical to the unsharded run. */
typedef struct { uint32_t idx, sample, score; } mpf_ufb_event;
/* tag: position of the call in the run (batch counter; 0xFFFFFFFF closes a climb).  The ranks advance in lock step, so
   an exchange that sees different tags must fail (return non-zero): the engine then stops with MPF_E_STATE instead of
   waiting forever.  Returns 0 on success; *all stays valid until the next call. */
typedef int (*mpf_ufb_exchange_fn)(void *arg, uint32_t tag, const mpf_ufb_event *local, uint32_t n_local,
                                   const mpf_ufb_event **all, uint32_t *n_all);
int mpf_ufboot_attach_sharded(mpf_engine *e, int32_t n_samples, int32_t n_local, const int32_t *sample_ids,
                              const uint16_t *samples_local /* [n_local][n_patterns] */, double epsilon,
                              mpf_ufb_exchange_fn exchange, void *arg);
int mpf_ufboot_detach(mpf_engine *e);
/* The exchanges of the multi-GPU path, native (ABI 7; mpboot_amd/host/rccl_exchange.cpp): RCCL over xGMI from inside the library,
   librccl opened at run time.  One communicator per process / GPU: rank 0 makes the id (mpf_rccl_unique_id, 128 bytes) and hands
   it to the others by whatever the host has; everybody calls mpf_rccl_create.  mpf_rccl_exchange IS an mpf_ufb_exchange_fn --
   pass it with the communicator as `arg` to mpf_ufboot_attach_sharded: one all-gather of fixed-size event blocks per scan
   batch on a stream of its own, a second one only when some rank has more than 4096 events.  mpf_rccl_allreduce_min: the
   "single all-reduce of best scores per round" of independent units (start trees, replicates).  The reference has nothing
   distributed: these replace what a multi-GPU mpboot host would otherwise write with MPI. */
typedef struct mpf_rccl mpf_rccl;
int mpf_rccl_available(void);
int mpf_rccl_unique_id(uint8_t *out /* [128] */);
int mpf_rccl_create(mpf_rccl **out, const uint8_t *id /* [128] */, int32_t rank, int32_t world, int32_t device);
void mpf_rccl_destroy(mpf_rccl *c);
int mpf_rccl_exchange(void *arg /* mpf_rccl* */, uint32_t tag, const mpf_ufb_event *local, uint32_t n_local,
                      const mpf_ufb_event **all, uint32_t *n_all);
int mpf_rccl_allreduce_min(mpf_rccl *c, uint32_t *vals, int32_t n);
int mpf_rccl_counters(const mpf_rccl *c, uint64_t *exchanges, uint64_t *overflows);
int mpf_ufboot_set_cutoff(mpf_engine *e, double logl_cutoff);            /* IQTree::logl_cutoff; 0 = none */
/* Climbs under other pattern weights than the attach-time ones (ratchet iterations: mpf_set_weights between attach and
   mpf_optimize_spr) are booked as the reference's default books them (iqtree.cpp:3283-3295): the length a candidate is
   filtered and recorded with is the ORIGINAL-alignment length of the tree booked last (of the climb's start tree for
   the first candidate) -- saveCurrentTree recomputes cur_logl from _pattern_pars before refreshing that array --, the
   per-sample REPS are the candidate's own.  on = 0 is params->no_hclimb1_bb (iqtree.cpp:3280): such climbs run without
   saveCurrentTree.  Takes effect at the next mpf_set_weights.  Weights that give an attach-time pattern weight 0 suspend
   the bookkeeping in either case (mpboot's ratchet only adds copies of sites). */
int mpf_ufboot_set_ratchet_booking(mpf_engine *e, int32_t on);
/* the main loop's per-iteration cut-off update, "top percent %" rule (iqtree.cpp:1662-1676, cutoff_percent = 10) */
/* params->multiple_hits (-mulhits): the update rule of iqtree.cpp:3498-3540 replaces the default one (:3684-3731) -- every
   tree whose REPS reaches a sample's best joins that sample's set (boot_trees_parsimony), a better one clears the set first;
   trees of one topology share the index of the first of them that hit (the reference's treels string map, :3500-3514); no
   random draws, boot_counts / boot_trees stay untouched.  Call right after the attach, before any tree is booked.
   (-topboot and -distinct_iter_top_boot: below.) */
int mpf_ufboot_set_mulhits(mpf_engine *e, int32_t on);
/* params->store_candidate_trees (-storetrees; off by default, tools.cpp:736): iqtree.cpp:3302-3346 -- every tree that reaches
   saveCurrentTree is looked up by topology (printTree(WT_TAXON_ID | WT_SORT_TAXA) as the key) BEFORE the cut-off test.  One
   met before counts as a duplicate (duplication_counter, mpf_ufboot_get_duplicates) and is skipped, unless its length
   improved on the recorded treels_logl entry (this happens on ratchet climbs, whose lengths come from _pattern_pars as it
   stands): then the entry is updated and the tree goes through the update rule under its OLD index, without the cut-off
   test.  Combines with every update rule.  Costs one canonical form per insertion test on the host (the reference prints,
   re-reads and prints the tree per test).  Call right after the attach, before any tree is booked. */
int mpf_ufboot_set_store_trees(mpf_engine *e, int32_t on);
int mpf_ufboot_get_duplicates(const mpf_engine *e, uint64_t *n);
/* params->store_top_boot_trees (-topboot N, together with -mulhits): the rule of iqtree.cpp:3542-3585 -- per sample the N best
   NEW trees (a tree whose topology was booked before is never added), best first, with boot_threshold behaving as in the
   reference (-INT_MAX until the first replacement in a full list).  After mpf_ufboot_set_mulhits(e, 1), before any tree is
   booked; n_top = 0 switches back to plain -mulhits.  mpf_ufboot_get_sample_top: boot_trees_parsimony_top[sample] as
   (tree index, rell) pairs, *n = its length, *threshold = boot_threshold[sample]. */
int mpf_ufboot_set_topboot(mpf_engine *e, int32_t n_top);
int mpf_ufboot_get_sample_top(const mpf_engine *e, int32_t sample, int64_t *trees, int32_t *rell, int32_t cap, int32_t *n, int32_t *threshold);
/* params->distinct_iter_top_boot (-distinct_iter_top_boot k, without -mulhits): the rule of iqtree.cpp:3587-3680 -- per sample at
   most k trees, one representative per search iteration, accepted against boot_threshold (the list's worst score) with a
   k / boot_counts tie draw from the shared random stream; boot_trees / boot_logl / boot_counts are maintained as that rule
   maintains them (mpf_ufboot_get_state), the list is read with mpf_ufboot_get_sample_top and the iteration each entry stands
   for with mpf_ufboot_get_sample_iters.  Before any tree is booked.  mpf_ufboot_set_iteration: IQTree::curIt, before each
   mpf_optimize_spr of a new search iteration. */
int mpf_ufboot_set_distinct_iter(mpf_engine *e, int32_t k);
int mpf_ufboot_set_iteration(mpf_engine *e, int32_t cur_it);
int mpf_ufboot_get_sample_iters(const mpf_engine *e, int32_t sample, int32_t *iters, int32_t cap, int32_t *n);
/* boot_trees_parsimony[sample] in increasing order: *n = its size, the first min(*n, cap) entries written to out (may be NULL) */
int mpf_ufboot_get_sample_trees(const mpf_engine *e, int32_t sample, int64_t *out, int32_t cap, int32_t *n);
int mpf_ufboot_next_cutoff(const mpf_engine *e, int32_t percent, double *logl_cutoff);
/* -cutoff_from_btrees (params->cutoff_from_btrees, tools.cpp:2442; ABI 7): boot_tree_orig_logl[b] (iqtree.h:766) = the logl under which
   sample b's tree was booked -- set at every acceptance of the default and the -distinct_iter_top_boot rule (iqtree.cpp:3716-3718,
   :3617-3619), only ever raised from its initial 0 by -mulhits (:3523-3527: with negative logls, never) --, and
   mpf_ufboot_next_cutoff returns their minimum (:1657-1660) instead of the percentile of the saved trees.  Any time after the
   attach; the array is kept whether or not the switch is on. */
int mpf_ufboot_set_cutoff_from_btrees(mpf_engine *e, int32_t on);
int mpf_ufboot_get_orig_logl(const mpf_engine *e, int32_t *out /* [n_samples] */);
int mpf_ufboot_num_trees(const mpf_engine *e, int64_t *n_trees);         /* treels_logl.size() */
int mpf_ufboot_tree_logl(const mpf_engine *e, double *out /* [n_trees] */);
/* boot_logl / boot_counts / boot_trees (any may be NULL) */
int mpf_ufboot_get_state(const mpf_engine *e, double *boot_logl, int32_t *boot_counts, int32_t *boot_trees);
/* topology of a tree some sample currently points to (boot_trees[b]), as back[] */
int mpf_ufboot_get_tree(const mpf_engine *e, int64_t tree_index, int32_t *back);
/* Iteration-parallel -bb (the reference's parallel form distributes the ITERATIONS of doTreeSearch over processes that meet from time
   to time, README.md:71-78): books another chain of the same run keeps.  Update k offers sample[k] the tree tree_of[k] (one of
   n_trees topologies, backs[n_trees][3 (2n - 1)], lengths[] = their lengths on the original alignment) at REPS length score[k]; a
   sample takes it when it is STRICTLY shorter than what it holds (the strict branch of saveCurrentTree's rule, iqtree.cpp:3686,
   :3710-3720; equal lengths keep the holder, no draw).  Default update rule, unsharded tracker, between two climbs. */
int mpf_ufboot_adopt(mpf_engine *e, int32_t n_updates, const int32_t *sample, const uint32_t *score, const int32_t *tree_of, int32_t n_trees,
                     const int32_t *backs, const uint32_t *lengths, int32_t *n_taken);
int mpf_ufboot_get_counters(const mpf_engine *e, uint64_t *tie_draws, uint64_t *events, uint64_t *reps_rows, double *reps_kernel_ms);

/* REPS -- resampling parsimony scores of candidate trees under B bootstrap weight vectors, the inner loop of
   IQTree::saveCurrentTree (iqtree.cpp:3411-3449): rell[m][b] = -sum_ptn pattern_pars[m][ptn] * boot[b][ptn].
   boot_samples_pars as IQTree::setParams builds them (iqtree.cpp:213-313), uploaded once; pattern_pars rows as
   mpf_pattern_scores / mpf_compute_parsimony return them.  Exact 32-bit sums. */
typedef struct mpf_reps mpf_reps;
int mpf_reps_create(mpf_reps **out, int32_t device, int32_t n_samples, int32_t n_patterns, const uint16_t *boot /* [B][P] */);
int mpf_reps_scores(mpf_reps *r, int32_t n_trees, const uint16_t *pattern_pars /* [M][P] */, int32_t *rell /* [M][B] */);
void mpf_reps_destroy(mpf_reps *r);

/* ---- The steps of IQTree::doTreeSearch BETWEEN two climbs (iqtree.cpp:1631-1965), device-free, on `back` records: for hosts
   without IQ-TREE's tree classes (bench.py, tests, a C++ driver) that want to run the flow the reference runs -- candidate tree ->
   floor(0.5 (n - 3)) random NNIs, or every second iteration a re-weighted alignment -> climb.  All draws come from the host's
   random_double() stream, handed over by state like the climb's own (mpf_set_tie_state / mpf_get_tie_state).  mpboot keeps its own
   code for these steps; the engine never calls them.  (mpboot_amd/host/iqflow.cpp; second witness oracle/iqflow_slow.py.)

   mpf_iq_random_nnis      IQTree::doRandomNNIs(numNNI) (iqtree.cpp:1083-1106) + PhyloTree::doOneRandomNNI (phylotree.cpp:3665-3711):
                           per NNI one random_int(n - 3) for the branch and two random_int(1) (always 0: the first neighbour at each
                           end); a branch that touches a node already used re-lists the branches and takes the same index (:1096-1103).
                           The listing order and which neighbour is "first" are this library's (ring order from tip 1).
   mpf_iq_perturb_weights  Alignment::createPerturbAlignment (alignment.cpp:1915-1969; -ratchet_percent 50, -ratchet_wgt 1,
                           tools.cpp:778-780): n_informative_sites * percent / 100 distinct sites of informative patterns, `add` more
                           copies of each one's pattern.  out[n_patterns].
   mpf_iq_topology_key     128-bit digest of the canonical unrooted topology: the key of CandidateSet::topologies (candidateset.cpp:110-150). */
int mpf_iq_random_nnis(int32_t n_taxa, int32_t *back, int32_t num_nni, uint64_t *tie_state, int32_t *n_relists);
int mpf_iq_perturb_weights(int32_t n_patterns, const int32_t *weights, const uint8_t *informative, int32_t percent, int32_t add,
                           uint64_t *tie_state, int32_t *out);
int mpf_iq_topology_key(int32_t n_taxa, const int32_t *back, uint64_t key[2]);

int mpf_get_stats(const mpf_engine *e, mpf_stats *out);
int mpf_reset_stats(mpf_engine *e);
/* tuning knobs (none of them changes a result):
     "scan_batch"      prune nodes speculated per launch at the start of a climb (then adaptive)
     "words_per_lane"  1|2|4 32-site words per lane in the Fitch kernels
     "reduce"          0 = DPP wave reduction, 1 = ds_bpermute
     "xcd_map"         1 = XCD-aware workgroup -> tile map of the scan kernel
     "scan_mode"       1 = device-walked SPR scan (radius <= 8), 0 = host-planned scan programs
     "scan_prog"       device-walked mode, DNA, radius <= 6: 1 = batches of more than "prog_min_descs" scan parts are first
                       turned into DFS programs on the device (k_walk_plan) and run with the children's vectors requested
                       one expansion ahead (k_scan_prog); 2 = every batch; 0 = never (k_scan_walk walks the tree itself)
     "prog_min_descs"  see "scan_prog" (default 256)
     "views_mode"      2 = chained refresh (stale paths run in registers, one launch; Fitch mode), 1 = all dependency
                       levels of a refresh in one launch, 0 = one launch per level
     "views_pipe"      level kernel, one word per lane: 1 (default) = a tile of "views_tile" words per workgroup, 64 / tile ops
                       per wave instruction, the next round's operands requested before this round's stores, tiles dealt
                       to the XCDs in contiguous runs; 0 = 32-word tiles, half a wave per op
     "views_tile"      32 | 16 | 8 | 4, or 0 (default) = the smallest of them that gives at most 256 workgroups
     "chain_max_ops"   refreshes of up to this many vectors use the chained kernel (default 512; larger ones are wide
                       rather than deep and take the level kernel)
     "split_below"     batches of at most this many prune nodes are cut into four scan parts each
     "split_cands"     larger batches: a prune node's P or Q neighbourhood is cut into four parts only when it holds more
                       insertion tests than this (default 64)
     "sankoff_short"   1 = two 16-bit costs per lane in the weighted kernels when no intermediate can overflow
                       (the reference's default arithmetic), 0 = always 32-bit (its -short_off)
     "plan_cache"      1 (default) = what the host prepares per topology -- the whole-tree refresh schedule, a sweep's scan
                       descriptors and its device program -- is kept while the topology stands (the same tree handed over again,
                       re-weighted, re-evaluated); 0 = everything is planned again every time
     "host_poll"       1 (default) = small batches: the host waits for the flag word the scan's last workgroup raises behind the
                       results it writes to pinned memory, instead of a stream synchronisation
     "check_counts"    1 = compare the kernel's candidate counts with the host's, and check the view bookkeeping
     "force_big"       1 = 64-bit addressing in the scan kernel even below 2 GiB of vectors (set before the first tree)
     "timing"          1 = HIP events around the scan kernels (mpf_stats scan_kernel_ms_total), 2 = around the refresh
                       kernels too (view_kernel_ms_total); an event pair costs about 10 us on the stream */
int mpf_set_option(mpf_engine *e, const char *key, int64_t value);
/* diagnostic (option "scan_trace" = 1): timeline of the last planned-program scan launch, four 64-bit words per workgroup:
   begin, end (100 MHz counter), XCC_ID << 32 | HW_ID, scan << 32 | tile << 16 | insertion tests.  *n_words is always set;
   out is filled when cap >= *n_words. */
int mpf_get_scan_trace(mpf_engine *e, uint64_t *out, uint64_t cap, uint64_t *n_words);
/* current value of an option of mpf_set_option; read-only: "kernel_states" = state rows the kernels carry (4: DNA and binary,
   20: protein and multistate data with at most 20 symbols in use, 32: multistate data beyond that or under a cost matrix) */
int mpf_get_option(const mpf_engine *e, const char *key, int64_t *value);

#ifdef __cplusplus
}
#endif
#endif /* MPFITCH_H */
