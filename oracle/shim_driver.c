/*
 * shim_driver.c -- TEST INFRASTRUCTURE ONLY.
 * An ordinary PLL program (alignment parser, partitions, instance, tree -- all the reference's own code from
 * oracle/_ref/obj) that asks PLL for a randomized stepwise-addition parsimony tree.  It is linked against
 * integration/pll_shim.cpp instead of the reference's fastDNAparsimony.c, so the three parsimony entry points
 * it calls run on libmpfitch.so.  Prints the same lines as `pll_ref_driver ras`.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "pll.h"

extern void allocateParsimonyDataStructures(pllInstance *tr, partitionList *pr);
extern int pll_shim_rec_of(pllInstance *tr, nodeptr p);

int main(int argc, char **argv)
{
  pllInstanceAttr attr;
  pllInstance *tr;
  pllAlignmentData *aln;
  pllQueue *parts;
  partitionList *pr;
  char pstr[256];
  int v, s, n;
  if (argc < 6) { fprintf(stderr, "usage: %s <aln.phy> <DNA|WAG> <dedup> <seed> <sprDist>\n", argv[0]); return 2; }
  memset(&attr, 0, sizeof attr);
  attr.rateHetModel = PLL_GAMMA;
  attr.fastScaling = PLL_FALSE;
  attr.saveMemory = PLL_FALSE;
  attr.useRecom = PLL_FALSE;
  attr.randomNumberSeed = 12345;
  attr.numberOfThreads = 1;
  tr = pllCreateInstance(&attr);
  aln = pllParseAlignmentFile(PLL_FORMAT_PHYLIP, argv[1]);
  if (!aln) { fprintf(stderr, "cannot parse %s\n", argv[1]); return 2; }
  snprintf(pstr, sizeof pstr, "%s, p1 = 1-%d\n", argv[2], aln->sequenceLength);
  parts = pllPartitionParseString(pstr);
  if (!pllPartitionsValidate(parts, aln)) return 2;
  pr = pllPartitionsCommit(parts, aln);
  pllQueuePartitionsDestroy(&parts);
  if (atoi(argv[3])) pllAlignmentRemoveDups(aln, pr);
  pllTreeInitTopologyForAlignment(tr, aln);
  if (!pllLoadAlignment(tr, aln, pr)) return 2;
  tr->randomNumberSeed = atol(argv[4]);
  allocateParsimonyDataStructures(tr, pr);
  pllMakeParsimonyTreeFast(tr, pr, atoi(argv[5]));
  n = tr->mxtips;
  printf("ras_score %u\n", tr->bestParsimony);
  printf("ras_topology");
  for (v = 1; v <= 2 * n - 2; v++)
    for (s = 0; s < (v <= n ? 1 : 3); s++) {
      nodeptr p = v <= n ? tr->nodeBaseAddress + (v - 1) : tr->nodeBaseAddress + n + 3 * (v - n - 1) + (2 - s);
      printf(" %d:%d", 3 * v + s, pll_shim_rec_of(tr, p->back));
    }
  printf("\n");
  pllFreeParsimonyDataStructures(tr, pr);
  return 0;
}
