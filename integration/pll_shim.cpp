// pll_shim.cpp -- the PLL-level drop-in: the three parsimony entry points PLL declares in pll.h
//     void allocateParsimonyDataStructures(pllInstance *, partitionList *)      (pll.h, fastDNAparsimony.c:1818)
//     void pllMakeParsimonyTreeFast(pllInstance *, partitionList *, int sprDist) (pll.h:1647, fastDNAparsimony.c:1857)
//     void pllFreeParsimonyDataStructures(pllInstance *, partitionList *)       (pll.h:1650, fastDNAparsimony.c:1843)
// re-implemented on libmpfitch.so.  Linked INSTEAD of the reference's fastDNAparsimony.c into an otherwise
// unmodified PLL program (oracle/shim_driver.c + the PLL objects of oracle/_ref), it makes that program
// build its randomized stepwise-addition tree on the MI355X; tests/test_gpu_dropin.py checks the result against
// what the reference's own fastDNAparsimony.c produced (tests/golden/*.json, "ras" entries).
//
// Compiles only where the reference headers exist (this container): -I/root/reference/pllrepo/src.
// The mpboot-level shim (pllOptimizeSprParsimony etc., which needs iqtree.h) is spelled out in INTEGRATION.md.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

extern "C" {
#include "pll.h"
}
#include "../include/mpfitch.h"

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

static mpf_engine *g_eng = nullptr;

static void die(const char *what)
{
  std::fprintf(stderr, "pll_shim: %s: %s\n", what, mpf_last_error());
  std::exit(3);
}

// record id = 3*number + slot, slot = position in the `next` ring counted from the record that nodep[number]
// pointed to when the tree was created (pllTreeInitDefaults allocates the three records contiguously, utils.c:2019-2044)
static int rec_of(pllInstance *tr, nodeptr p)
{
  if (!p) return -1;
  if (p->number <= tr->mxtips) return 3 * p->number;
  const long idx = (long)(p - tr->nodeBaseAddress) - tr->mxtips;
  return 3 * p->number + (2 - (int)(idx % 3));
}
static nodeptr ptr_of(pllInstance *tr, int rec)
{
  const int number = rec / 3, slot = rec % 3;
  if (number <= tr->mxtips) return tr->nodeBaseAddress + (number - 1);
  return tr->nodeBaseAddress + tr->mxtips + 3 * (number - tr->mxtips - 1) + (2 - slot);
}

extern "C" void allocateParsimonyDataStructures(pllInstance *tr, partitionList *pr)
{
  const int n = tr->mxtips, P = tr->originalCrunchedLength;
  if (g_eng) { mpf_engine_destroy(g_eng); g_eng = nullptr; }
  std::vector<uint8_t> codes((size_t)n * P);
  for (int i = 1; i <= n; i++) std::memcpy(&codes[(size_t)(i - 1) * P], tr->yVector[i], (size_t)P);
  mpf_config cfg;
  std::memset(&cfg, 0, sizeof cfg);
  cfg.device = 0;
  cfg.n_taxa = n;
  cfg.n_patterns = P;
  cfg.datatype = mpf_datatype_of_pll(pr->partitionData[0]->dataType);
  cfg.keep_all_sites = 0;
  if (mpf_engine_create(&g_eng, &cfg, codes.data(), tr->aliaswgt)) die("mpf_engine_create");
  if (mpf_seed_ties(g_eng, MPF_TIE_FIRST, 0)) die("mpf_seed_ties");   // PLL original: strict '<', no random draws
}

extern "C" void pllMakeParsimonyTreeFast(pllInstance *tr, partitionList *pr, int sprDist)
{
  (void)pr;
  if (!g_eng) die("allocateParsimonyDataStructures was not called");
  uint32_t score = 0;
  if (mpf_make_parsimony_tree(g_eng, (int64_t)tr->randomNumberSeed, sprDist, &score)) die("mpf_make_parsimony_tree");
  // hand the topology back to the PLL instance: back links, start, counters
  const int n = tr->mxtips;
  std::vector<int32_t> back(3 * (size_t)(2 * n - 1));
  if (mpf_get_tree(g_eng, back.data())) die("mpf_get_tree");
  for (int v = 1; v <= 2 * n - 2; v++)
    for (int s = 0; s < (v <= n ? 1 : 3); s++) ptr_of(tr, 3 * v + s)->back = ptr_of(tr, back[(size_t)(3 * v + s)]);
  tr->start = tr->nodep[1];
  tr->ntips = n;
  tr->nextnode = 2 * n - 1;
  tr->bestParsimony = score;
}

extern "C" void pllFreeParsimonyDataStructures(pllInstance *tr, partitionList *pr)
{
  (void)tr; (void)pr;
  mpf_engine_destroy(g_eng);
  g_eng = nullptr;
}

// the replaced translation unit also exported this helper, which PLL's likelihood code links against
// (pllInternal.h:149, fastDNAparsimony.c:145)
extern "C" unsigned int bitcount_32_bit(unsigned int i) { return (unsigned int)__builtin_popcount(i); }

// test hook for the driver: record links of the current PLL tree
extern "C" int pll_shim_rec_of(pllInstance *tr, nodeptr p) { return rec_of(tr, p); }
