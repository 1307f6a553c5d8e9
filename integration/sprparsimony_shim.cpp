// sprparsimony_shim.cpp -- the mpboot-level drop-in: the entry points the reference declares in sprparsimony.h:13-54
// and defines in sprparsimony.cpp, re-implemented on libmpfitch.so with the SAME C++ signatures (IQTree stays an opaque
// class here; mangled names do not depend on its definition), so that a mpboot build can list this file INSTEAD of
// sprparsimony.cpp and link -lmpfitch.  IQTree / Params members are reached through integration/mpboot_hooks.h.
//
//   int  pllOptimizeSprParsimony(pllInstance*, partitionList*, int mintrav, int maxtrav, IQTree*)   .h:31  -> .cpp:3244
//   void _pllComputeRandomizedStepwiseAdditionParsimonyTree(pllInstance*, partitionList*, int, IQTree*) .h:19 -> :3224
//   void _allocateParsimonyDataStructures(pllInstance*, partitionList*)                              .h:21
//   void _pllFreeParsimonyDataStructures(pllInstance*, partitionList*)                               .h:22  -> :3062
//   void pllComputePatternParsimony(..., unsigned short*, int*) / (..., double*, double*)            .h:35-36 -> :3328, :3363
//   void pllComputeSiteParsimony(..., int* | unsigned short*, int nsite, int*)                       .h:39-40 -> :3403, :3424
//   int  pllCalcMinParsScorePattern(pllInstance*, int dataType, int site)                            .h:42  -> :2513
//   void resetGlobalParamOnNewAln()                                                                  .h:13  -> :143
// plus the globals sprparsimony.cpp defines (:128-141): iqtree, bestTreeScoreHits, first_call, doing_stepwise_addition.
//
// -cost (weighted parsimony): as in the reference, every entry point dispatches on the global pllCostMatrix
// (sprparsimony.cpp:556-641, :967-1030): non-NULL -> the engine is created with mpf_engine_create_sankoff on that
// matrix; initializeCostMatrix() (:159-188) and the global highest_cost (:130) are defined here.
//
// Not provided: pllSaveCurrentTreeSprParsimony (the per-candidate call-back disappears: the bookkeeping runs inside
// mpf_optimize_spr, mpboot_hooks.h ufboot_sync) and the tool functions of sprparsimony.h:46-54.
//
// Compiled and exercised by oracle/Makefile's `ref` target (oracle/spr_shim_driver.cpp + the reference's own PLL objects)
// and tests/test_gpu_dropin.py.  Builds only where the reference's pll.h exists: -I/root/reference/pllrepo/src.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

extern "C" {
#include "pll.h"
double randum(long *seed);                 // pllInternal.h:34 (utils.c:335-358)
}
#include "mpboot_hooks.h"

// PLL partition data type (pll.h:238-244) -> libmpfitch data type
static inline int mpf_datatype_of_pll(int pll_type)
{
  switch (pll_type) {
    case PLL_AA_DATA: return MPF_AA;
    case PLL_BINARY_DATA: return MPF_BIN;          // "BIN" partitions (iqtree.cpp:526-527)
    case PLL_GENERIC_32: return MPF_GENERIC;       // "MOR" partitions (iqtree.cpp:524-525)
    default: return MPF_DNA;
  }
}
static inline int mpf_states_of(int dt) { return dt == MPF_AA ? 20 : dt == MPF_BIN ? 2 : dt == MPF_GENERIC ? 32 : 4; }

// globals the reference DEFINES in iqtree.cpp:35-43 and sprparsimony.cpp consumes: the cost matrix of -cost
extern unsigned int *pllCostMatrix;        // cost[i * pllCostNstates + j] = cost of i -> j, NULL = Fitch
extern int pllCostNstates;

IQTree *iqtree = nullptr;                  // sprparsimony.cpp:128
parsimonyNumber highest_cost = 0;          // :130, max(cost) + 1: the cost of a state a tip does not have
unsigned long bestTreeScoreHits = 0;       // :129 (kept for link compatibility; the engine counts ties itself)
bool first_call = true;                    // :140
bool doing_stepwise_addition = false;      // :141

namespace {

mpf_mpboot_hooks g_hooks;
bool g_have_hooks = false;
mpf_engine *g_eng = nullptr;
int g_n = 0, g_P = 0;
std::vector<int32_t> g_weights;            // what the engine currently holds
std::vector<int32_t> g_first_weights;      // the weights of the unperturbed alignment (first allocation)
std::vector<int32_t> g_informative;
bool g_tracking = false;                   // online UFBoot bookkeeping attached

[[noreturn]] void die(const char *what)
{
  // the reference reports through outError() / assert (sprparsimony.cpp:575-577); same effect here
  std::fprintf(stderr, "mpfitch shim: %s: %s\n", what, mpf_last_error());
  std::exit(EXIT_FAILURE);
}

double draw(void *) { return g_hooks.random_double(); }

// The tie stream for the duration of one engine call.  mpboot draws every tie-break of the parsimony search from ONE
// random_double() stream (tools.cpp:3363-3368; sprparsimony.cpp:2171-2172, :3004, :3309-3310; iqtree.cpp:3594).  When the host
// can name the generator's state (hooks rng_get_state / rng_set_state) and it is the lcg64 the engine implements, the state
// travels with the call: in before, out after -- nothing on the mpboot side draws while the engine runs (saveCurrentTree's
// draws are made by the engine itself; ufboot_sync is called after the state is back).  Otherwise every draw is a call-back
// and the sweep loop stays on the host.
bool g_stream_lent = false;
void lend_stream()
{
  uint64_t st = 0, mul = 0, add = 0;
  g_stream_lent = g_hooks.rng_get_state && g_hooks.rng_set_state && g_hooks.rng_get_state(&st, &mul, &add) &&
                  mul == MPF_LCG64_MULTIPLIER && add == MPF_LCG64_ADDEND;
  if (g_stream_lent) {
    if (mpf_set_tie_state(g_eng, st)) die("mpf_set_tie_state");
  } else if (mpf_set_rand_callback(g_eng, draw, nullptr)) die("mpf_set_rand_callback");
}
void return_stream()
{
  if (!g_stream_lent) return;
  uint64_t st = 0;
  if (mpf_get_tie_state(g_eng, &st)) die("mpf_get_tie_state");
  g_hooks.rng_set_state(st);
  g_stream_lent = false;
}

// record id = 3 * number + slot; slot = position in the `next` ring counted from the record nodep[number] pointed to when
// the instance was created (the three records of an inner node are contiguous, pllrepo/src/utils.c:2019-2044)
int rec_of(pllInstance *tr, nodeptr p)
{
  if (p->number <= tr->mxtips) return 3 * p->number;
  const long idx = (long)(p - tr->nodeBaseAddress) - tr->mxtips;
  return 3 * p->number + (2 - (int)(idx % 3));
}
nodeptr ptr_of(pllInstance *tr, int rec)
{
  const int number = rec / 3, slot = rec % 3;
  if (number <= tr->mxtips) return tr->nodeBaseAddress + (number - 1);
  return tr->nodeBaseAddress + tr->mxtips + 3 * (number - tr->mxtips - 1) + (2 - slot);
}

void destroy_engine()
{
  if (g_eng) mpf_engine_destroy(g_eng);
  g_eng = nullptr;
  g_tracking = false;
  g_weights.clear();
  g_first_weights.clear();
}

// _allocateParsimonyDataStructures (:3032-3060): compressDNA reads tr->yVector and tr->aliaswgt.  The tips stay in HBM
// for the life of the alignment; a later "allocation" only re-weights (ratchet climbs, bootstrap replicates).
void ensure_engine(pllInstance *tr, partitionList *pr)
{
  if (!g_have_hooks) { std::fprintf(stderr, "mpfitch shim: mpfitch_shim_install() was not called\n"); std::exit(EXIT_FAILURE); }
  const int n = tr->mxtips, P = tr->originalCrunchedLength;
  if (g_eng && (n != g_n || P != g_P)) destroy_engine();
  if (!g_eng) {
    std::vector<uint8_t> codes((size_t)n * (size_t)P);
    for (int i = 1; i <= n; i++) std::memcpy(&codes[(size_t)(i - 1) * (size_t)P], tr->yVector[i], (size_t)P);
    mpf_config cfg;
    std::memset(&cfg, 0, sizeof cfg);
    cfg.n_taxa = n;
    cfg.n_patterns = P;
    cfg.datatype = mpf_datatype_of_pll(pr->partitionData[0]->dataType);
    cfg.keep_all_sites = !g_hooks.sort_alignment;
    if (pllCostMatrix) {
      // the reference's dispatch on pllCostMatrix (:556-641, :967-1030): weighted engine, same entry points
      if (pllCostNstates != mpf_states_of(cfg.datatype)) { std::fprintf(stderr, "mpfitch shim: cost matrix of %d states on %d-state data\n", pllCostNstates, mpf_states_of(cfg.datatype)); std::exit(EXIT_FAILURE); }
      if (mpf_engine_create_sankoff(&g_eng, &cfg, codes.data(), tr->aliaswgt, pllCostMatrix)) die("mpf_engine_create_sankoff");
    } else if (mpf_engine_create(&g_eng, &cfg, codes.data(), tr->aliaswgt)) die("mpf_engine_create");
    if (mpf_seed_ties(g_eng, MPF_TIE_RANDOM, 0)) die("mpf_seed_ties");     // (the stream itself: lend_stream() before every call)
    g_n = n;
    g_P = P;
    g_weights.assign(tr->aliaswgt, tr->aliaswgt + P);
    g_first_weights = g_weights;
    g_informative.resize((size_t)P);
    if (mpf_get_informative(g_eng, g_informative.data())) die("mpf_get_informative");
    return;
  }
  if (std::memcmp(g_weights.data(), tr->aliaswgt, (size_t)P * sizeof(int32_t)) != 0) {
    // (an attached tracker is suspended while other weights than its own are in force, and resumed afterwards)
    if (mpf_set_weights(g_eng, tr->aliaswgt)) die("mpf_set_weights");
    g_weights.assign(tr->aliaswgt, tr->aliaswgt + P);
  }
}

// IQTree::saveCurrentTree's bookkeeping for the SPR path (perSiteScores = gbo_replicates > 0, :3245).  The tracker is
// attached once, on the unperturbed alignment (its weights are IQTree's original_sample); re-weighted (ratchet) climbs
// are booked by the engine as the reference books them (iqtree.cpp:3283-3295) unless -no_hclimb1_bb (:3280).
void ensure_tracking()
{
  if (g_hooks.gbo_replicates <= 0 || !g_hooks.boot_sample) return;
  if (!g_tracking) {
    if (g_weights != g_first_weights) return;          // (first call on perturbed weights: attach when the original ones are back)
    const int B = g_hooks.gbo_replicates;
    std::vector<uint16_t> s((size_t)B * (size_t)g_P);
    for (int b = 0; b < B; b++) std::memcpy(&s[(size_t)b * (size_t)g_P], g_hooks.boot_sample(iqtree, b), (size_t)g_P * sizeof(uint16_t));
    if (mpf_ufboot_attach(g_eng, B, s.data(), g_hooks.ufboot_epsilon)) die("mpf_ufboot_attach");
    if (mpf_ufboot_set_ratchet_booking(g_eng, g_hooks.no_hclimb1_bb ? 0 : 1)) die("mpf_ufboot_set_ratchet_booking");
    if (g_hooks.store_candidate_trees && mpf_ufboot_set_store_trees(g_eng, 1)) die("mpf_ufboot_set_store_trees");
    if (g_hooks.multiple_hits && mpf_ufboot_set_mulhits(g_eng, 1)) die("mpf_ufboot_set_mulhits");
    if (g_hooks.cutoff_from_btrees && mpf_ufboot_set_cutoff_from_btrees(g_eng, 1)) die("mpf_ufboot_set_cutoff_from_btrees");
    if (g_hooks.multiple_hits && g_hooks.store_top_boot_trees > 0 && mpf_ufboot_set_topboot(g_eng, g_hooks.store_top_boot_trees)) die("mpf_ufboot_set_topboot");
    if (!g_hooks.multiple_hits && g_hooks.distinct_iter_top_boot > 0 && mpf_ufboot_set_distinct_iter(g_eng, g_hooks.distinct_iter_top_boot))
      die("mpf_ufboot_set_distinct_iter");
    g_tracking = true;
  }
  if (g_hooks.cur_iteration && mpf_ufboot_set_iteration(g_eng, g_hooks.cur_iteration(iqtree))) die("mpf_ufboot_set_iteration");
  if (g_hooks.logl_cutoff && mpf_ufboot_set_cutoff(g_eng, g_hooks.logl_cutoff(iqtree))) die("mpf_ufboot_set_cutoff");
}

void push_tree(pllInstance *tr)            // pllInstance -> record links
{
  const int n = tr->mxtips;
  std::vector<int32_t> back(3 * (size_t)(2 * n - 1), -1);
  for (int v = 1; v <= 2 * n - 2; v++)
    for (int s = 0; s < (v <= n ? 1 : 3); s++) {
      nodeptr p = ptr_of(tr, 3 * v + s);
      back[(size_t)(3 * v + s)] = rec_of(tr, p->back);
    }
  if (mpf_set_tree(g_eng, back.data())) die("mpf_set_tree");
}

void pull_tree(pllInstance *tr)            // record links -> pllInstance (what hookupDefault would have left)
{
  const int n = tr->mxtips;
  std::vector<int32_t> back(3 * (size_t)(2 * n - 1));
  if (mpf_get_tree(g_eng, back.data())) die("mpf_get_tree");
  for (int v = 1; v <= 2 * n - 2; v++)
    for (int s = 0; s < (v <= n ? 1 : 3); s++) ptr_of(tr, 3 * v + s)->back = ptr_of(tr, back[(size_t)(3 * v + s)]);
  // tr->nodep[] as nodeRectifierPars leaves it (:2046-2101): tips in place, the inner entries = the records by which a
  // preorder walk from nodep[1]->back enters the inner nodes -- callers that index nodep[] afterwards see the reference's state
  std::vector<int32_t> order((size_t)(2 * n - 2));
  if (mpf_get_node_order(g_eng, order.data())) die("mpf_get_node_order");
  for (int i = 1; i <= 2 * n - 2; i++) tr->nodep[i] = ptr_of(tr, order[(size_t)(i - 1)]);
  tr->start = tr->nodep[1];                // :2089
  tr->ntips = n;
  tr->nextnode = 2 * n - 1;
}

}  // namespace

void mpfitch_shim_install(const mpf_mpboot_hooks *hooks)
{
  g_hooks = *hooks;
  g_have_hooks = hooks->random_double != nullptr;
}

mpf_engine *mpfitch_shim_engine(void) { return g_eng; }

void resetGlobalParamOnNewAln()
{
  destroy_engine();
  iqtree = nullptr;
  bestTreeScoreHits = 0;
  first_call = true;
  doing_stepwise_addition = false;
}

// initializeCostMatrix (:159-188; called from IQTree::initializePLL, iqtree.cpp:609, after pllCostMatrix is set): the
// reference copies the matrix into its SIMD layout here; the engine takes pllCostMatrix itself at its creation, what
// remains is highest_cost and making sure an engine built for another matrix is not reused
void initializeCostMatrix()
{
  unsigned int m = 0;
  for (int i = 0; i < pllCostNstates * pllCostNstates; i++) m = pllCostMatrix[i] > m ? pllCostMatrix[i] : m;
  highest_cost = m + 1;
  destroy_engine();
}

void _allocateParsimonyDataStructures(pllInstance *tr, partitionList *pr) { ensure_engine(tr, pr); }

// idempotent like the reference (:3062-3104).  The engine itself is kept: the tips of the alignment stay resident and
// the next allocation only re-weights; resetGlobalParamOnNewAln() lets go of the device memory.
void _pllFreeParsimonyDataStructures(pllInstance *, partitionList *) {}

void _pllComputeRandomizedStepwiseAdditionParsimonyTree(pllInstance *tr, partitionList *pr, int sprDist, IQTree *_iqtree)
{
  doing_stepwise_addition = true;
  iqtree = _iqtree;
  ensure_engine(tr, pr);
  // perSiteScores = PLL_FALSE here (:3228): the engine's tree builder never runs the UFBoot bookkeeping
  uint32_t score = 0;
  lend_stream();
  if (mpf_make_parsimony_tree(g_eng, (int64_t)tr->randomNumberSeed, sprDist, &score)) die("mpf_make_parsimony_tree");
  return_stream();
  // makePermutationFast (:2221-2242) draws one randum() per taxon from tr->randomNumberSeed: leave the seed where the
  // reference leaves it, for whoever draws from it next
  for (int i = 1; i <= tr->mxtips; i++) (void)randum(&tr->randomNumberSeed);
  pull_tree(tr);
  tr->bestParsimony = score;
  doing_stepwise_addition = false;
}

int pllOptimizeSprParsimony(pllInstance *tr, partitionList *pr, int mintrav, int maxtrav, IQTree *_iqtree)
{
  iqtree = _iqtree;
  if (g_have_hooks && g_hooks.ratchet_climb && g_hooks.ratchet_climb(iqtree)) {
    // _updateInternalPllOnRatchet (:3022-3029)
    for (int i = 0; i < pr->numberOfPartitions; i++)
      for (int ptn = pr->partitionData[i]->lower; ptn < pr->partitionData[i]->upper; ptn++)
        tr->aliaswgt[ptn] = g_hooks.pattern_frequency(iqtree, ptn);
  }
  // the reference re-reads tr->aliaswgt only on ratchet climbs, on the first call and under on_opt_btree (:3249-3254);
  // comparing the weights on every call covers the three cases
  ensure_engine(tr, pr);
  first_call = false;
  ensure_tracking();
  push_tree(tr);
  uint32_t start = 0;
  if (g_hooks.cur_score) {                 // assert(-iqtree->curScore == tr->bestParsimony), :3279
    if (mpf_score_tree(g_eng, &start)) die("mpf_score_tree");
    if ((double)start != -g_hooks.cur_score(iqtree)) {
      std::fprintf(stderr, "mpfitch shim: start tree scores %u, mpboot expects %.0f\n", start, -g_hooks.cur_score(iqtree));
      std::abort();
    }
  }
  uint32_t score = 0;
  lend_stream();
  if (mpf_optimize_spr(g_eng, mintrav, maxtrav, &score)) die("mpf_optimize_spr");
  return_stream();
  pull_tree(tr);
  tr->bestParsimony = score;
  // (a climb that was booked: on the original weights always, on perturbed ones unless -no_hclimb1_bb)
  if (g_tracking && (g_weights == g_first_weights || !g_hooks.no_hclimb1_bb) && g_hooks.ufboot_sync) g_hooks.ufboot_sync(iqtree, g_eng);
  return (int)score;                       // startMP of the last sweep = the final score (:3318)
}

// valid for the tree the last pllOptimizeSprParsimony left behind (the reference: right after an evaluate with
// perSiteScores = 1); entries of patterns the engine dropped are left untouched under sort_alignment (:3382-3384)
void pllComputePatternParsimony(pllInstance *tr, partitionList *pr, unsigned short *ptn_pars, int *cur_pars)
{
  (void)pr;
  if (!g_eng) die("pllComputePatternParsimony before any allocation");
  std::vector<uint16_t> pp((size_t)g_P);
  int32_t total = 0;
  if (mpf_pattern_scores(g_eng, pp.data(), &total)) die("mpf_pattern_scores");
  int sum = 0;
  for (int ptn = 0; ptn < g_P; ptn++) {
    if (g_hooks.sort_alignment && !g_informative[(size_t)ptn]) continue;
    ptn_pars[ptn] = pp[(size_t)ptn];
    sum += (int)pp[(size_t)ptn] * tr->aliaswgt[ptn];
  }
  if (cur_pars) *cur_pars = sum;
}

void pllComputePatternParsimony(pllInstance *tr, partitionList *pr, double *ptn_npars, double *cur_npars)
{
  (void)pr;
  if (!g_eng) die("pllComputePatternParsimony before any allocation");
  std::vector<uint16_t> pp((size_t)g_P);
  int32_t total = 0;
  if (mpf_pattern_scores(g_eng, pp.data(), &total)) die("mpf_pattern_scores");
  int sum = 0;
  for (int ptn = 0; ptn < g_P; ptn++) {    // the double overload covers lower..upper (:3337-3343), negated
    ptn_npars[ptn] = -(double)pp[(size_t)ptn];
    sum += (int)pp[(size_t)ptn] * tr->aliaswgt[ptn];
  }
  if (cur_npars) *cur_npars = -(double)sum;
}

void pllComputeSiteParsimony(pllInstance *tr, partitionList *pr, int *site_pars, int nsite, int *cur_pars)
{
  (void)tr; (void)pr;
  if (!g_eng) die("pllComputeSiteParsimony before any allocation");
  int32_t total = 0;
  if (mpf_site_scores(g_eng, site_pars, nsite, &total)) die("mpf_site_scores");
  if (cur_pars) *cur_pars = total;
}

void pllComputeSiteParsimony(pllInstance *tr, partitionList *pr, unsigned short *site_pars, int nsite, int *cur_pars)
{
  std::vector<int> tmp((size_t)(nsite > 0 ? nsite : 0));
  pllComputeSiteParsimony(tr, pr, tmp.data(), nsite, cur_pars);
  for (int i = 0; i < nsite; i++) site_pars[i] = (unsigned short)tmp[(size_t)i];
}

int pllCalcMinParsScorePattern(pllInstance *tr, int dataType, int site)
{
  const int n = tr->mxtips;
  std::vector<uint8_t> col((size_t)n);
  for (int j = 1; j <= n; j++) col[(size_t)(j - 1)] = tr->yVector[j][site];
  int32_t out = 0;
  if (mpf_min_pars_score_patterns(mpf_datatype_of_pll(dataType), n, 1, col.data(), &out)) die("mpf_min_pars_score_patterns");
  return out;
}
