/*
 * ref_driver.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * A small command-line driver of OUR OWN that compiles the reference's PLL
 * parsimony translation unit *where it lies* (no copy is made):
 *
 *     #include "pllrepo/src/fastDNAparsimony.c"      (-I/root/reference)
 *
 * so that its file-static functions (evaluateParsimony, testInsertParsimony,
 * rearrangeParsimony, stepwiseAddition, compressDNA ...) can be exercised
 * one by one and their results written out as golden vectors.  The rest of
 * PLL (alignment parser, newick parser, tree plumbing) is compiled from
 * /root/reference/pllrepo/src/ by oracle/Makefile into oracle/_ref/.
 *
 * The reference functions exercised here are the PLL-original twins
 * (pllrepo/src/fastDNAparsimony.c) of mpboot's
 * sprparsimony.cpp: same Fitch arithmetic, same tip packing, same SPR
 * enumeration order, deterministic "first best wins" tie rule.
 *
 * Output is line-oriented text; tests/golden/make_golden.py turns it into
 * the committed JSON fixtures.
 *
 * Record ids ("rec"): every PLL node record gets the integer
 *     rec = 3*number + slot,
 * slot 0 = the record nodep[number] pointed to at tree creation, slot 1 its
 * ->next, slot 2 its ->next->next (utils.c:2019-2044 allocates the three
 * records contiguously, highest address first in the cycle). Tips have slot 0
 * only.  `back[rec]` = rec id of the record across the branch, or -1.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>

#include "pllrepo/src/fastDNAparsimony.c"

static pllInstance *TR;
static partitionList *PR;
static pllAlignmentData *ALN;

static int rec_of(nodeptr p)
{
  if (p == NULL) return -1;
  if (p->number <= TR->mxtips) return 3 * p->number;
  {
    long idx = (long)(p - TR->nodeBaseAddress) - TR->mxtips;
    int j = (int)(idx % 3);           /* 0,1,2 in allocation order            */
    return 3 * p->number + (2 - j);   /* highest address = creation nodep = 0 */
  }
}

static nodeptr ptr_of(int rec)
{
  int number = rec / 3, slot = rec % 3;
  if (number <= TR->mxtips) return TR->nodeBaseAddress + (number - 1);
  return TR->nodeBaseAddress + TR->mxtips + 3 * (number - TR->mxtips - 1) + (2 - slot);
}

static void print_topology(const char *tag)
{
  int v, s, n = TR->mxtips;
  printf("%s", tag);
  for (v = 1; v <= 2 * n - 2; v++)
    for (s = 0; s < (v <= n ? 1 : 3); s++)
      printf(" %d:%d", 3 * v + s, rec_of(ptr_of(3 * v + s)->back));
  printf("\n");
}

static void load_alignment(const char *file, const char *type, int dedup)
{
  pllInstanceAttr attr;
  pllQueue *parts;
  char pstr[256];

  memset(&attr, 0, sizeof(attr));
  attr.rateHetModel = PLL_GAMMA;
  attr.fastScaling = PLL_FALSE;
  attr.saveMemory = PLL_FALSE;
  attr.useRecom = PLL_FALSE;
  attr.randomNumberSeed = 12345;
  attr.numberOfThreads = 1;
  TR = pllCreateInstance(&attr);
  ALN = pllParseAlignmentFile(PLL_FORMAT_PHYLIP, file);
  if (!ALN) { fprintf(stderr, "cannot parse %s\n", file); exit(2); }
  snprintf(pstr, sizeof pstr, "%s, p1 = 1-%d\n", type, ALN->sequenceLength);
  parts = pllPartitionParseString(pstr);
  if (!pllPartitionsValidate(parts, ALN)) { fprintf(stderr, "bad partition\n"); exit(2); }
  PR = pllPartitionsCommit(parts, ALN);
  pllQueuePartitionsDestroy(&parts);
  if (dedup) pllAlignmentRemoveDups(ALN, PR);
  pllTreeInitTopologyForAlignment(TR, ALN);
  if (!pllLoadAlignment(TR, ALN, PR)) { fprintf(stderr, "load failed\n"); exit(2); }
}

static void load_newick(const char *nwk)
{
  pllNewickTree *t = pllNewickParseString(nwk);
  if (!t) { fprintf(stderr, "bad newick\n"); exit(2); }
  if (!pllValidateNewick(t)) pllNewickUnroot(t);
  pllTreeInitTopologyNewick(TR, t, PLL_FALSE);
  pllNewickParseDestroy(&t);
}

static void reset_orientation(void)
{
  int i;
  for (i = TR->mxtips + 1; i <= 2 * TR->mxtips - 1; i++) {
    nodeptr p = TR->nodep[i];
    p->xPars = 1; p->next->xPars = 0; p->next->next->xPars = 0;
  }
}

/* ---- dump: encoded tips, weights, informative flags, packed tip vectors ---- */
static void cmd_dump(void)
{
  int i, k, n = TR->mxtips, P = TR->originalCrunchedLength;
  int *inf = (int *)malloc(sizeof(int) * P);
  size_t W, S, w;
  determineUninformativeSites(TR, PR, inf);
  compressDNA(TR, PR, inf);
  W = PR->partitionData[0]->parsimonyLength;
  S = PR->partitionData[0]->states;
  printf("n %d\nP %d\nS %zu\nW %zu\n", n, P, S, W);
  for (i = 1; i <= n; i++) printf("name %d %s\n", i, TR->nameList[i]);
  printf("weights");
  for (k = 0; k < P; k++) printf(" %d", TR->aliaswgt[k]);
  printf("\ninformative");
  for (k = 0; k < P; k++) printf(" %d", inf[k]);
  printf("\n");
  for (i = 1; i <= n; i++) {
    printf("codes %d", i);
    for (k = 0; k < P; k++) printf(" %d", (int)TR->yVector[i][k]);
    printf("\n");
  }
  for (i = 1; i <= n; i++) {
    printf("tipvec %d", i);
    for (w = 0; w < S * W; w++)
      printf(" %08x", PR->partitionData[0]->parsVect[W * S * (size_t)i + w]);
    printf("\n");
  }
  free(inf);
}

/* ---- score: Fitch length of each user tree (full traversal) ---- */
static void cmd_score(const char *treefile)
{
  FILE *f = fopen(treefile, "r");
  static char line[1 << 22];
  int t = 0;
  if (!f) { perror(treefile); exit(2); }
  allocateParsimonyDataStructures(TR, PR);
  while (fgets(line, sizeof line, f)) {
    unsigned int s;
    if (strlen(line) < 3) continue;
    load_newick(line);
    reset_orientation();
    TR->bestParsimony = UINT_MAX;
    s = evaluateParsimony(TR, PR, TR->start, PLL_TRUE);
    printf("tree %d score %u\n", t, s);
    print_topology("topology");
    t++;
  }
  fclose(f);
}

/* ---- own enumeration of one side, one candidate at a time ---- */
static int QUIET = 0;
static unsigned long long NTESTS = 0;
static void enum_side(nodeptr p, nodeptr q, int mintrav, int maxtrav)
{
  if (--mintrav <= 0) {
    TR->bestParsimony = UINT_MAX;
    TR->insertNode = TR->removeNode = NULL;
    testInsertParsimony(TR, PR, p, q, PLL_FALSE);
    NTESTS++;
    if (!QUIET) printf(" %d:%u", rec_of(q), TR->bestParsimony);
  }
  if (q->number > TR->mxtips && --maxtrav > 0) {
    enum_side(p, q->next->back, mintrav, maxtrav);
    enum_side(p, q->next->next->back, mintrav, maxtrav);
  }
}

/* ---- scan: SPR neighbourhood of every prune node of one user tree ---- */
static void cmd_scan(const char *treefile, int maxtrav)
{
  FILE *f = fopen(treefile, "r");
  static char line[1 << 22];
  int i, n;
  unsigned int cur;
  if (!f) { perror(treefile); exit(2); }
  if (!fgets(line, sizeof line, f)) exit(2);
  fclose(f);
  allocateParsimonyDataStructures(TR, PR);
  load_newick(line);
  reset_orientation();
  n = TR->mxtips;
  TR->ntips = n;
  nodeRectifierPars(TR);
  TR->bestParsimony = UINT_MAX;
  cur = evaluateParsimony(TR, PR, TR->start, PLL_TRUE);
  printf("score %u\n", cur);
  print_topology("topology");
  printf("order");
  for (i = 1; i <= 2 * n - 2; i++) printf(" %d", rec_of(TR->nodep[i]));
  printf("\n");
  for (i = 1; i <= 2 * n - 2; i++) {
    nodeptr p = TR->nodep[i], q = p->back;
    int mt = maxtrav;
    if (mt > TR->ntips - 3) mt = TR->ntips - 3;

    /* (1) the reference's own enumeration + first-best rule, tree left unchanged */
    TR->bestParsimony = cur;
    TR->insertNode = TR->removeNode = NULL;
    rearrangeParsimony(TR, PR, p, 1, maxtrav, PLL_FALSE);
    printf("best %d %u %d %d\n", rec_of(p), TR->bestParsimony,
           rec_of(TR->removeNode), rec_of(TR->insertNode));

    /* (2) every candidate's score, reference testInsertParsimony, our enumeration */
    printf("cands %d P", rec_of(p));
    if (mt >= 1) {
      evaluateParsimony(TR, PR, p, PLL_FALSE);
      if (p->number > n) {
        nodeptr p1 = p->next->back, p2 = p->next->next->back;
        if (p1->number > n || p2->number > n) {
          removeNodeParsimony(p);
          if (p1->number > n) { enum_side(p, p1->next->back, 1, mt); enum_side(p, p1->next->next->back, 1, mt); }
          if (p2->number > n) { enum_side(p, p2->next->back, 1, mt); enum_side(p, p2->next->next->back, 1, mt); }
          hookupDefault(p->next, p1); hookupDefault(p->next->next, p2);
          newviewParsimony(TR, PR, p);
        }
      }
      printf(" Q");
      if (q->number > n && mt > 0) {
        nodeptr q1 = q->next->back, q2 = q->next->next->back;
        if ((q1->number > n && (q1->next->back->number > n || q1->next->next->back->number > n)) ||
            (q2->number > n && (q2->next->back->number > n || q2->next->next->back->number > n))) {
          removeNodeParsimony(q);
          if (q1->number > n) { enum_side(q, q1->next->back, 2, mt); enum_side(q, q1->next->next->back, 2, mt); }
          if (q2->number > n) { enum_side(q, q2->next->back, 2, mt); enum_side(q, q2->next->next->back, 2, mt); }
          hookupDefault(q->next, q1); hookupDefault(q->next->next, q2);
          newviewParsimony(TR, PR, q);
        }
      }
    }
    printf("\n");
  }
  TR->bestParsimony = UINT_MAX;
  printf("score_after %u\n", evaluateParsimony(TR, PR, TR->start, PLL_TRUE));
}

/* ---- time: wall-clock of the reference's insertion tests (AVX) over the prune nodes of one tree,
        bounded by a time budget; bench.py's cpu_baseline leg ("kind": "reference") ---- */
#include <time.h>
static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
static void cmd_time(const char *treefile, int maxtrav, double budget)
{
  FILE *f = fopen(treefile, "r");
  static char line[1 << 22];
  int i, n, done = 0, pass = 0;
  double t0, t1;
  if (!f) { perror(treefile); exit(2); }
  if (!fgets(line, sizeof line, f)) exit(2);
  fclose(f);
  allocateParsimonyDataStructures(TR, PR);
  load_newick(line);
  reset_orientation();
  n = TR->mxtips;
  TR->ntips = n;
  nodeRectifierPars(TR);
  TR->bestParsimony = UINT_MAX;
  printf("score %u\n", evaluateParsimony(TR, PR, TR->start, PLL_TRUE));
  QUIET = 1;
  NTESTS = 0;
  t0 = now_s();
  for (pass = 0; now_s() - t0 < budget; pass++)
  for (i = 1; i <= 2 * n - 2; i++) {
    nodeptr p = TR->nodep[i], q = p->back;
    int mt = maxtrav;
    if (mt > TR->ntips - 3) mt = TR->ntips - 3;
    evaluateParsimony(TR, PR, p, PLL_FALSE);
    if (p->number > n) {
      nodeptr p1 = p->next->back, p2 = p->next->next->back;
      if (p1->number > n || p2->number > n) {
        removeNodeParsimony(p);
        if (p1->number > n) { enum_side(p, p1->next->back, 1, mt); enum_side(p, p1->next->next->back, 1, mt); }
        if (p2->number > n) { enum_side(p, p2->next->back, 1, mt); enum_side(p, p2->next->next->back, 1, mt); }
        hookupDefault(p->next, p1); hookupDefault(p->next->next, p2);
        newviewParsimony(TR, PR, p);
      }
    }
    if (q->number > n && mt > 0) {
      nodeptr q1 = q->next->back, q2 = q->next->next->back;
      if ((q1->number > n && (q1->next->back->number > n || q1->next->next->back->number > n)) ||
          (q2->number > n && (q2->next->back->number > n || q2->next->next->back->number > n))) {
        removeNodeParsimony(q);
        if (q1->number > n) { enum_side(q, q1->next->back, 2, mt); enum_side(q, q1->next->next->back, 2, mt); }
        if (q2->number > n) { enum_side(q, q2->next->back, 2, mt); enum_side(q, q2->next->next->back, 2, mt); }
        hookupDefault(q->next, q1); hookupDefault(q->next->next, q2);
        newviewParsimony(TR, PR, q);
      }
    }
    done++;
    if (now_s() - t0 > budget) break;
  }
  t1 = now_s();
  printf("timed prune_nodes %d of %d tests %llu seconds %.6f passes %d\n", done, 2 * n - 2, NTESTS, t1 - t0, pass);
}

/* ---- spr: PLL-original hill climb from a user tree (loop of fastDNAparsimony.c:1919-1938) ---- */
static void cmd_spr(const char *treefile, int maxtrav)
{
  FILE *f = fopen(treefile, "r");
  static char line[1 << 22];
  int i, n, sweep = 0, total_moves = 0;
  unsigned int randomMP, startMP;
  double t_climb;
  if (!f) { perror(treefile); exit(2); }
  if (!fgets(line, sizeof line, f)) exit(2);
  fclose(f);
  QUIET = getenv("REF_DRIVER_QUIET") != NULL;      /* timing runs (bench.py): no per-move lines, no topologies */
  allocateParsimonyDataStructures(TR, PR);
  load_newick(line);
  reset_orientation();
  n = TR->mxtips;
  TR->ntips = n;
  nodeRectifierPars(TR);
  TR->bestParsimony = UINT_MAX;
  TR->bestParsimony = evaluateParsimony(TR, PR, TR->start, PLL_TRUE);
  printf("start_score %u\n", TR->bestParsimony);
  if (!QUIET) print_topology("start_topology");
  randomMP = TR->bestParsimony;
  t_climb = now_s();
  do {
    int moves = 0;
    startMP = randomMP;
    nodeRectifierPars(TR);
    for (i = 1; i <= 2 * n - 2; i++) {
      rearrangeParsimony(TR, PR, TR->nodep[i], 1, maxtrav, PLL_FALSE);
      if (TR->bestParsimony < randomMP) {
        if (!QUIET) printf("move %d %d %d %u\n", sweep, rec_of(TR->removeNode), rec_of(TR->insertNode), TR->bestParsimony);
        total_moves++;
        restoreTreeRearrangeParsimony(TR, PR);
        randomMP = TR->bestParsimony;
        moves++;
      }
    }
    printf("sweep %d score %u moves %d\n", sweep, randomMP, moves);
    sweep++;
  } while (randomMP < startMP);
  printf("climb moves %d sweeps %d seconds %.6f\n", total_moves, sweep, now_s() - t_climb);
  printf("final_score %u\n", randomMP);
  if (!QUIET) print_topology("final_topology");
  TR->bestParsimony = UINT_MAX;
  printf("final_check %u\n", evaluateParsimony(TR, PR, TR->start, PLL_TRUE));
}

/* ---- refine: one bootstrap-refinement replicate as IQTree::optimizeBootTrees runs it (iqtree.cpp:2515-2866 ->
        pllOptimizeSprParsimony with on_opt_btree): the pattern weights of the replicate go into tr->aliaswgt (what
        _updateInternalPllOnRatchet does, sprparsimony.cpp:3022-3029), the parsimony structures are rebuilt (compressDNA
        on the new weights) and the sample's tree is hill-climbed.  Timed as a whole: bench.py's CPU side of the
        bootstrap metric ("kind": "reference").  PLL-original first-best rule (the tie rule does not change the cost of
        an insertion test). ---- */
static void cmd_refine(const char *treefile, int maxtrav, const char *wfile)
{
  FILE *f = fopen(treefile, "r");
  static char line[1 << 22];
  int i, n, k, sweep = 0, moves = 0, P = TR->originalCrunchedLength;
  unsigned int randomMP, startMP, first;
  unsigned long long tests0;
  double t0, t1;
  if (!f) { perror(treefile); exit(2); }
  if (!fgets(line, sizeof line, f)) exit(2);
  fclose(f);
  f = fopen(wfile, "r");
  if (!f) { perror(wfile); exit(2); }
  for (k = 0; k < P; k++) { int w; if (fscanf(f, "%d", &w) != 1) { fprintf(stderr, "weights: %d values expected\n", P); exit(2); } TR->aliaswgt[k] = w; }
  fclose(f);
  load_newick(line);
  t0 = now_s();
  allocateParsimonyDataStructures(TR, PR);
  reset_orientation();
  n = TR->mxtips;
  TR->ntips = n;
  nodeRectifierPars(TR);
  TR->bestParsimony = UINT_MAX;
  first = TR->bestParsimony = evaluateParsimony(TR, PR, TR->start, PLL_TRUE);
  randomMP = TR->bestParsimony;
  (void)tests0;
  do {
    startMP = randomMP;
    nodeRectifierPars(TR);
    for (i = 1; i <= 2 * n - 2; i++) {
      rearrangeParsimony(TR, PR, TR->nodep[i], 1, maxtrav, PLL_FALSE);
      if (TR->bestParsimony < randomMP) {
        restoreTreeRearrangeParsimony(TR, PR);
        randomMP = TR->bestParsimony;
        moves++;
      }
    }
    sweep++;
  } while (randomMP < startMP);
  t1 = now_s();
  printf("refined start_score %u final_score %u moves %d sweeps %d seconds %.6f\n", first, randomMP, moves, sweep, t1 - t0);
  TR->bestParsimony = UINT_MAX;
  printf("final_check %u\n", evaluateParsimony(TR, PR, TR->start, PLL_TRUE));
}

/* ---- ras: the reference's own pllMakeParsimonyTreeFast, untouched ---- */
static void cmd_ras(long seed, int sprDist)
{
  int i, n = TR->mxtips;
  int *perm = (int *)malloc(sizeof(int) * (n + 1));
  long s2 = seed;
  /* the permutation the call below will draw (same generator, same seed) */
  { long save = TR->randomNumberSeed; TR->randomNumberSeed = s2; makePermutationFast(perm, n, TR); TR->randomNumberSeed = save; }
  if (!getenv("REF_DRIVER_QUIET")) {
    printf("perm");
    for (i = 1; i <= n; i++) printf(" %d", perm[i]);
    printf("\n");
  }
  free(perm);
  TR->randomNumberSeed = seed;
  QUIET = getenv("REF_DRIVER_QUIET") != NULL;      /* timing runs (bench.py): no topology */
  {
    double t0 = now_s();
    allocateParsimonyDataStructures(TR, PR);
    pllMakeParsimonyTreeFast(TR, PR, sprDist);
    printf("ras_seconds %.6f\n", now_s() - t0);    /* compressDNA + randomized stepwise addition + the SPR sweeps behind it */
  }
  printf("ras_score %u\n", TR->bestParsimony);
  if (!QUIET) print_topology("ras_topology");
  printf("start %d\n", rec_of(TR->start));
  TR->bestParsimony = UINT_MAX;
  printf("ras_check %u\n", evaluateParsimony(TR, PR, TR->start, PLL_TRUE));
}

/* ---- rasx: stepwise addition only, with a checkpoint per added taxon ---- */
static void cmd_rasx(long seed)
{
  int n = TR->mxtips, nextsp;
  int *perm = (int *)malloc(sizeof(int) * (n + 1));
  nodeptr p, f;
  allocateParsimonyDataStructures(TR, PR);
  TR->randomNumberSeed = seed;
  makePermutationFast(perm, n, TR);
  TR->ntips = 0;
  TR->nextnode = n + 1;
  buildSimpleTree(TR, PR, perm[1], perm[2], perm[3]);
  f = TR->start;
  while (TR->ntips < n) {
    nodeptr q, r;
    int counter = 4;
    TR->bestParsimony = INT_MAX;
    nextsp = ++(TR->ntips);
    p = TR->nodep[perm[nextsp]];
    q = TR->nodep[(TR->nextnode)++];
    p->back = q; q->back = p;
    stepwiseAddition(TR, PR, q, f->back);
    printf("add %d tip %d best %u insert %d\n", nextsp, perm[nextsp], TR->bestParsimony, rec_of(TR->insertNode));
    r = TR->insertNode->back;
    hookupDefault(q->next, TR->insertNode);
    hookupDefault(q->next->next, r);
    computeTraversalInfoParsimony(q, TR->ti, &counter, n, PLL_FALSE);
    TR->ti[0] = counter;
    newviewParsimonyIterativeFast(TR, PR);
  }
  print_topology("rasx_topology");
  TR->bestParsimony = UINT_MAX;
  printf("rasx_check %u\n", evaluateParsimony(TR, PR, TR->start, PLL_TRUE));
  free(perm);
}

int main(int argc, char **argv)
{
  if (argc < 5) {
    fprintf(stderr,
      "usage: %s <cmd> <aln.phy> <DNA|WAG> <dedup 0|1> [args]\n"
      "  dump | score <trees> | scan <tree> <maxtrav> | spr <tree> <maxtrav> | time <tree> <maxtrav> <seconds> | ras <seed> <sprDist> | rasx <seed>\n"
      "  refine <tree> <maxtrav> <pattern-weights file>\n",
      argv[0]);
    return 2;
  }
  load_alignment(argv[2], argv[3], atoi(argv[4]));
  if (!strcmp(argv[1], "dump")) cmd_dump();
  else if (!strcmp(argv[1], "score")) cmd_score(argv[5]);
  else if (!strcmp(argv[1], "scan")) cmd_scan(argv[5], atoi(argv[6]));
  else if (!strcmp(argv[1], "spr")) cmd_spr(argv[5], atoi(argv[6]));
  else if (!strcmp(argv[1], "time")) cmd_time(argv[5], atoi(argv[6]), atof(argv[7]));
  else if (!strcmp(argv[1], "refine")) cmd_refine(argv[5], atoi(argv[6]), argv[7]);
  else if (!strcmp(argv[1], "ras")) cmd_ras(atol(argv[5]), atoi(argv[6]));
  else if (!strcmp(argv[1], "rasx")) cmd_rasx(atol(argv[5]));
  else { fprintf(stderr, "unknown command %s\n", argv[1]); return 2; }
  return 0;
}
