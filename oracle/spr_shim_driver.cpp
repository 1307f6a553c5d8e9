/*
 * spr_shim_driver.cpp -- TEST INFRASTRUCTURE ONLY.
 * Plays the part of mpboot's IQTree layer for integration/sprparsimony_shim.cpp: an ordinary PLL program (the
 * reference's alignment parser, partitions, instance and tree code from oracle/_ref/obj, and the reference's own SPRNG
 * generator for random_double()) that calls the mpboot-level entry points exactly as iqtree.cpp / phyloanalysis.cpp do
 *     _pllComputeRandomizedStepwiseAdditionParsimonyTree   (phyloanalysis.cpp:1165)
 *     pllOptimizeSprParsimony                              (iqtree.cpp:2132), plain, re-weighted (ratchet) and -bb
 *     pllComputePatternParsimony / pllComputeSiteParsimony (iqtree.cpp:3365)
 *     pllCalcMinParsScorePattern                           (iqtree.cpp:3827)
 * and prints what they leave in the pllInstance.  tests/test_gpu_dropin.py replays the same calls on the CPU oracle.
 *
 * usage: spr_shim_driver <aln.phy> <DNA|WAG> <dedup> <sprng seed> <pll seed> <maxtrav> <B> [cost-matrix file | - [no_hclimb1_bb
 *                         [mulhits [storetrees [stream hand-over: 1 (default) = rng_get_state / rng_set_state hooks, 0 = per-draw call-back only]]]]]
 * With a cost matrix (S x S unsigned entries, row-major) the program sets the globals IQTree::initializePLL sets for -cost
 * (iqtree.cpp:601-615) and calls initializeCostMatrix(): the shim then dispatches to the weighted engine.
 */
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>

extern "C" {
#include "pll.h"
}
#include "sprng/sprng.h"
#include "../integration/mpboot_hooks.h"

// the reference's declarations (sprparsimony.h:13-42)
void resetGlobalParamOnNewAln();
void _pllComputeRandomizedStepwiseAdditionParsimonyTree(pllInstance *tr, partitionList *partitions, int sprDist, IQTree *_iqtree);
void _pllFreeParsimonyDataStructures(pllInstance *tr, partitionList *pr);
int pllOptimizeSprParsimony(pllInstance *tr, partitionList *pr, int mintrav, int maxtrav, IQTree *iqtree);
void pllComputePatternParsimony(pllInstance *tr, partitionList *pr, unsigned short *ptn_pars, int *cur_pars);
void pllComputeSiteParsimony(pllInstance *tr, partitionList *pr, int *site_pars, int nsite, int *cur_pars);
int pllCalcMinParsScorePattern(pllInstance *tr, int dataType, int site);

void initializeCostMatrix();
// the globals iqtree.cpp:35-43 defines
unsigned int *pllCostMatrix = nullptr;
int pllCostNstates = 0;

// ---- the "IQTree" of this program: just the members the shim's hooks read
struct Host {
  int *stream = nullptr;
  bool ratchet = false;
  std::vector<int> freq;                       // aln->at(ptn).frequency
  double cur_score = 0;
  bool check_score = false;
  int B = 0, P = 0;
  std::vector<unsigned short> samples;         // boot_samples_pars
  double logl_cutoff = 0;
  int mulhits = 0;
};
static Host H;

static double hk_random(void) { return sprng(H.stream); }
// the stream's state through SPRNG's public pack / unpack interface (sprng/sprng.h:61-62; layout sprng/lcg64.c:486-497)
static uint64_t be_load(const unsigned char *p, int nbytes)
{
  uint64_t v = 0;
  for (int i = 0; i < nbytes; i++) v = (v << 8) | p[i];
  return v;
}
static int hk_rng_get(uint64_t *st, uint64_t *mul, uint64_t *add)
{
  char *b = nullptr;
  if (pack_sprng(H.stream, &b) <= 0 || !b) return 0;
  const unsigned char *p = (const unsigned char *)b + std::strlen(b) + 1;
  *add = be_load(p + 24, 4);
  *st = be_load(p + 28, 8);
  *mul = be_load(p + 36, 8);
  std::free(b);
  return 1;
}
static void hk_rng_set(uint64_t st)
{
  char *b = nullptr;
  if (pack_sprng(H.stream, &b) <= 0 || !b) { std::fprintf(stderr, "pack_sprng failed\n"); std::exit(2); }
  unsigned char *p = (unsigned char *)b + std::strlen(b) + 1 + 28;
  for (int i = 0; i < 8; i++) p[i] = (unsigned char)(st >> (8 * (7 - i)));
  free_sprng(H.stream);
  H.stream = unpack_sprng(b);
  std::free(b);
}
// what the host's stream and the engine look like after a call: the generator's state (the NEXT random_double() follows from
// it) and how many persistent climb kernels the engine has launched so far
static void print_stream(const char *tag)
{
  uint64_t st = 0, mul = 0, add = 0;
  hk_rng_get(&st, &mul, &add);
  mpf_stats ms;
  std::memset(&ms, 0, sizeof ms);
  if (mpfitch_shim_engine()) mpf_get_stats(mpfitch_shim_engine(), &ms);
  std::printf("%s_rng_state %llu\n%s_climb_launches %llu\n", tag, (unsigned long long)st, tag, (unsigned long long)ms.climb_launches);
}
static int hk_ratchet(IQTree *) { return H.ratchet; }
static int hk_opt_btree(IQTree *) { return 0; }
static int hk_freq(IQTree *, int ptn) { return H.freq[(size_t)ptn]; }
static double hk_score(IQTree *) { return H.cur_score; }
static const unsigned short *hk_boot(IQTree *, int b) { return &H.samples[(size_t)b * (size_t)H.P]; }
static double hk_cutoff(IQTree *) { return H.logl_cutoff; }

static void hk_sync(IQTree *, mpf_engine *e)
{
  int64_t nt = 0;
  mpf_ufboot_num_trees(e, &nt);
  std::vector<double> tl((size_t)nt), bl((size_t)H.B);
  std::vector<int32_t> bc((size_t)H.B), bt((size_t)H.B);
  mpf_ufboot_tree_logl(e, tl.data());
  mpf_ufboot_get_state(e, bl.data(), bc.data(), bt.data());
  std::printf("ufb_ntrees %lld\nufb_tree_logl", (long long)nt);
  for (double v : tl) std::printf(" %.0f", v);
  std::printf("\nufb_boot_logl");
  for (double v : bl) std::printf(" %.0f", v);
  std::printf("\nufb_boot_counts");
  for (int v : bc) std::printf(" %d", v);
  std::printf("\nufb_boot_trees");
  for (int v : bt) std::printf(" %d", v);
  std::printf("\n");
  if (H.mulhits) {                                  // boot_trees_parsimony: "sample size entries..." per sample, flattened
    std::printf("ufb_boot_sets");
    for (int b = 0; b < H.B; b++) {
      int32_t k = 0;
      mpf_ufboot_get_sample_trees(e, b, nullptr, 0, &k);
      std::vector<int64_t> st((size_t)std::max(k, 1));
      mpf_ufboot_get_sample_trees(e, b, st.data(), k, &k);
      std::printf(" %d", (int)k);
      for (int i = 0; i < k; i++) std::printf(" %lld", (long long)st[(size_t)i]);
    }
    std::printf("\n");
  }
}

static unsigned long long g_lcg;
static unsigned lcg3() { g_lcg = g_lcg * 6364136223846793005ULL + 1442695040888963407ULL; return (unsigned)((g_lcg >> 33) % 3ULL); }

static int rec_of(pllInstance *tr, nodeptr p)
{
  if (p->number <= tr->mxtips) return 3 * p->number;
  const long idx = (long)(p - tr->nodeBaseAddress) - tr->mxtips;
  return 3 * p->number + (2 - (int)(idx % 3));
}
static void print_tree(const char *tag, pllInstance *tr)
{
  const int n = tr->mxtips;
  std::printf("%s_score %u\n%s_topology", tag, tr->bestParsimony, tag);
  for (int v = 1; v <= 2 * n - 2; v++)
    for (int s = 0; s < (v <= n ? 1 : 3); s++) {
      nodeptr p = v <= n ? tr->nodeBaseAddress + (v - 1) : tr->nodeBaseAddress + n + 3 * (v - n - 1) + (2 - s);
      std::printf(" %d:%d", 3 * v + s, rec_of(tr, p->back));
    }
  std::printf("\n");
}

static pllInstance *tr = nullptr;
static partitionList *pr = nullptr;

static void load(const char *file, const char *model, int dedup)
{
  pllInstanceAttr attr;
  std::memset(&attr, 0, sizeof attr);
  attr.rateHetModel = PLL_GAMMA;
  attr.fastScaling = PLL_FALSE;
  attr.saveMemory = PLL_FALSE;
  attr.useRecom = PLL_FALSE;
  attr.randomNumberSeed = 12345;
  attr.numberOfThreads = 1;
  tr = pllCreateInstance(&attr);
  pllAlignmentData *aln = pllParseAlignmentFile(PLL_FORMAT_PHYLIP, file);
  if (!aln) { std::fprintf(stderr, "cannot parse %s\n", file); std::exit(2); }
  char pstr[256];
  std::snprintf(pstr, sizeof pstr, "%s, p1 = 1-%d\n", model, aln->sequenceLength);
  pllQueue *parts = pllPartitionParseString(pstr);
  if (!pllPartitionsValidate(parts, aln)) std::exit(2);
  pr = pllPartitionsCommit(parts, aln);
  pllQueuePartitionsDestroy(&parts);
  if (dedup) pllAlignmentRemoveDups(aln, pr);
  pllTreeInitTopologyForAlignment(tr, aln);
  if (!pllLoadAlignment(tr, aln, pr)) std::exit(2);
}

// spr_shim_driver time <aln.phy> <DNA|WAG> <sprng seed> <maxtrav> <newick file> <repetitions> [stream hand-over 1|0]
// bench.py's leg "through the reference-side binding": pllOptimizeSprParsimony (iqtree.cpp:2132) on a pllInstance that holds the
// tree of the Newick file, timed around the call itself -- marshalling of the topology in and out included -- once per
// repetition from the same start tree and the same SPRNG seed (the first one also creates the engine: tips packed, buffers made).
#include <chrono>
static int time_main(int argc, char **argv)
{
  if (argc < 8) { std::fprintf(stderr, "usage: %s time <aln.phy> <DNA|WAG> <sprng seed> <maxtrav> <newick file> <repetitions> [hand-over]\n", argv[0]); return 2; }
  load(argv[2], argv[3], 0);
  const int seed = std::atoi(argv[4]), maxtrav = std::atoi(argv[5]), reps = std::atoi(argv[7]);
  const bool handover = argc <= 8 || std::atoi(argv[8]);
  FILE *f = std::fopen(argv[6], "r");
  if (!f) { std::perror(argv[6]); return 2; }
  static char line[1 << 22];
  if (!std::fgets(line, sizeof line, f)) return 2;
  std::fclose(f);
  H.P = tr->originalCrunchedLength;
  H.freq.assign(tr->aliaswgt, tr->aliaswgt + H.P);
  mpf_mpboot_hooks hooks;
  std::memset(&hooks, 0, sizeof hooks);
  hooks.random_double = hk_random;
  if (handover) { hooks.rng_get_state = hk_rng_get; hooks.rng_set_state = hk_rng_set; }
  hooks.ratchet_climb = hk_ratchet;
  hooks.on_opt_btree = hk_opt_btree;
  hooks.pattern_frequency = hk_freq;
  hooks.sort_alignment = 1;
  resetGlobalParamOnNewAln();
  mpfitch_shim_install(&hooks);
  const int n = tr->mxtips;
  for (int rep = 0; rep < reps; rep++) {
    pllNewickTree *t = pllNewickParseString(line);
    if (!t) { std::fprintf(stderr, "bad newick\n"); return 2; }
    if (!pllValidateNewick(t)) pllNewickUnroot(t);
    pllTreeInitTopologyNewick(tr, t, PLL_FALSE);
    pllNewickParseDestroy(&t);
    if (H.stream) free_sprng(H.stream);
    H.stream = init_sprng(0, 1, seed, SPRNG_DEFAULT);
    if (rep == 0) {                                  // the start tree as record links, for whoever wants to replay the call
      std::printf("time_start_topology");
      for (int v = 1; v <= 2 * n - 2; v++)
        for (int s = 0; s < (v <= n ? 1 : 3); s++) {
          nodeptr p = v <= n ? tr->nodeBaseAddress + (v - 1) : tr->nodeBaseAddress + n + 3 * (v - n - 1) + (2 - s);
          std::printf(" %d:%d", 3 * v + s, rec_of(tr, p->back));
        }
      std::printf("\n");
    }
    mpf_stats m0, m1;
    std::memset(&m0, 0, sizeof m0);
    std::memset(&m1, 0, sizeof m1);
    if (mpfitch_shim_engine()) mpf_get_stats(mpfitch_shim_engine(), &m0);
    const auto t0 = std::chrono::steady_clock::now();
    const int score = pllOptimizeSprParsimony(tr, pr, 1, maxtrav, nullptr);
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    mpf_get_stats(mpfitch_shim_engine(), &m1);
    uint64_t st = 0, mul = 0, add = 0;
    hk_rng_get(&st, &mul, &add);
    std::printf("time_climb rep %d seconds %.6f score %d moves %llu climb_launches %llu insertion_tests %llu rng_state %llu\n", rep, dt, score,
                (unsigned long long)(m1.moves_applied - m0.moves_applied), (unsigned long long)(m1.climb_launches - m0.climb_launches),
                (unsigned long long)(m1.insertion_tests - m0.insertion_tests), (unsigned long long)st);
  }
  resetGlobalParamOnNewAln();
  return 0;
}

int main(int argc, char **argv)
{
  if (argc > 1 && !std::strcmp(argv[1], "time")) return time_main(argc, argv);
  if (argc < 8) { std::fprintf(stderr, "usage: %s <aln.phy> <DNA|WAG> <dedup> <sprng seed> <pll seed> <maxtrav> <B>\n", argv[0]); return 2; }
  load(argv[1], argv[2], std::atoi(argv[3]));
  const int maxtrav = std::atoi(argv[6]);
  const int P = tr->originalCrunchedLength, n = tr->mxtips;
  H.stream = init_sprng(0, 1, std::atoi(argv[4]), SPRNG_DEFAULT);      // init_random, tools.cpp:3326
  H.B = std::atoi(argv[7]);
  H.P = P;
  H.freq.assign(tr->aliaswgt, tr->aliaswgt + P);
  const std::vector<int> w0 = H.freq;
  // boot_samples_pars stand-in: weight x {0, 1, 2}, reproducible from the seed (the test regenerates it)
  g_lcg = 0x9E3779B97F4A7C15ULL ^ (unsigned long long)std::atoll(argv[4]);
  H.samples.resize((size_t)H.B * (size_t)P);
  for (int b = 0; b < H.B; b++)
    for (int p = 0; p < P; p++) H.samples[(size_t)b * (size_t)P + (size_t)p] = (unsigned short)((unsigned)w0[(size_t)p] * lcg3());

  mpf_mpboot_hooks hooks;
  std::memset(&hooks, 0, sizeof hooks);
  hooks.random_double = hk_random;
  if (argc <= 12 || std::atoi(argv[12])) {
    hooks.rng_get_state = hk_rng_get;
    hooks.rng_set_state = hk_rng_set;
  }
  hooks.ratchet_climb = hk_ratchet;
  hooks.on_opt_btree = hk_opt_btree;
  hooks.pattern_frequency = hk_freq;
  hooks.cur_score = hk_score;
  hooks.sort_alignment = 1;
  hooks.gbo_replicates = H.B;
  hooks.boot_sample = hk_boot;
  hooks.ufboot_epsilon = 0.5;
  hooks.logl_cutoff = hk_cutoff;
  hooks.ufboot_sync = hk_sync;
  hooks.no_hclimb1_bb = argc > 9 ? std::atoi(argv[9]) : 0;     // mpboot's default books ratchet climbs too (iqtree.cpp:3280)
  hooks.multiple_hits = H.mulhits = argc > 10 ? std::atoi(argv[10]) : 0;   // -mulhits
  hooks.store_candidate_trees = argc > 11 ? std::atoi(argv[11]) : 0;       // -storetrees
  resetGlobalParamOnNewAln();
  mpfitch_shim_install(&hooks);
  static std::vector<unsigned int> cost;
  if (argc > 8 && std::strcmp(argv[8], "-") != 0) {
    const int S = pr->partitionData[0]->states;
    FILE *cf = std::fopen(argv[8], "r");
    if (!cf) { std::perror(argv[8]); return 2; }
    cost.resize((size_t)S * (size_t)S);
    for (size_t i = 0; i < cost.size(); i++)
      if (std::fscanf(cf, "%u", &cost[i]) != 1) { std::fprintf(stderr, "cost matrix: %d x %d entries expected\n", S, S); return 2; }
    std::fclose(cf);
    pllCostMatrix = cost.data();
    pllCostNstates = S;
    initializeCostMatrix();
  }
  IQTree *iq = nullptr;

  // 1. start tree (phyloanalysis.cpp:1165: sprDist = 0 keeps it a pure stepwise addition)
  tr->randomNumberSeed = std::atol(argv[5]);
  _pllComputeRandomizedStepwiseAdditionParsimonyTree(tr, pr, 0, iq);
  print_tree("ras", tr);
  print_stream("ras");
  std::printf("ras_seed_after %ld\nras_nodep", (long)tr->randomNumberSeed);
  for (int i = 1; i <= 2 * n - 2; i++) std::printf(" %d", rec_of(tr, tr->nodep[i]));
  std::printf("\n");

  // 2. the SPR climb of one search iteration, with the -bb bookkeeping when B > 0 (iqtree.cpp:2132)
  H.cur_score = -(double)tr->bestParsimony;
  pllOptimizeSprParsimony(tr, pr, 1, maxtrav, iq);
  print_tree("spr", tr);
  print_stream("spr");
  {
    std::vector<unsigned short> pp((size_t)P + 16, 65535);
    int cur = 0;
    pllComputePatternParsimony(tr, pr, pp.data(), &cur);
    std::printf("ptn_total %d\nptn_pars", cur);
    for (int p = 0; p < P; p++) std::printf(" %u", (unsigned)pp[(size_t)p]);
    std::printf("\n");
    if (!pllCostMatrix) {                          // per-site counters exist in the Fitch engine only (sprparsimony.cpp:294-376)
      int nsite = 0;
      for (int p = 0; p < P; p++) nsite += tr->aliaswgt[p];
      std::vector<int> sp((size_t)nsite + 8, -1);
      pllComputeSiteParsimony(tr, pr, sp.data(), nsite + 8, &cur);
      std::printf("site_total %d\nsite_pars", cur);
      for (int s = 0; s < nsite + 8; s++) std::printf(" %d", sp[(size_t)s]);
    }
    std::printf("\nmin_pars");
    for (int p = 0; p < P; p++) std::printf(" %d", pllCalcMinParsScorePattern(tr, pr->partitionData[0]->dataType, p));
    std::printf("\n");
  }

  // 3. a ratchet climb: perturbed pattern frequencies (iqtree.cpp:1706-1716), then the climb on the original weights
  // createPerturbAlignment only ADDS copies of sites (alignment.cpp:1915-1969): every frequency stays >= the original one
  for (int p = 0; p < P; p++) H.freq[(size_t)p] = w0[(size_t)p] * (1 + (int)lcg3());
  H.ratchet = true;
  H.cur_score = 0;
  hooks.cur_score = nullptr;                       // the perturbed start score is not known to this driver
  mpfitch_shim_install(&hooks);
  pllOptimizeSprParsimony(tr, pr, 1, maxtrav, iq);
  print_tree("ratchet", tr);
  print_stream("ratchet");
  H.freq = w0;
  pllOptimizeSprParsimony(tr, pr, 1, maxtrav, iq);   // on_ratchet_hclimb2: original weights again
  print_tree("final", tr);
  print_stream("final");
  _pllFreeParsimonyDataStructures(tr, pr);
  resetGlobalParamOnNewAln();
  return 0;
}
