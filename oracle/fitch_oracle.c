/*
 * fitch_oracle.c -- CPU restatement of the reference's Fitch parsimony path.
 * TEST INFRASTRUCTURE ONLY (see fitch_oracle.h).  Parity status: PINNED against
 * oracle/_ref/pll_ref_driver (the reference's PLL parsimony sources compiled
 * where they lie) through tests/golden/ fixtures.
 *
 * Reference files restated (all under /root/reference):
 *   sprparsimony.cpp             -- mpboot's engine (variant ORC_TIE_RANDOM)
 *   pllrepo/src/fastDNAparsimony.c -- PLL original (variant ORC_TIE_FIRST)
 * The two differ on this path only by (a) the tie rule and (b) the extra
 * evaluateParsimony(p) at the top of rearrangeParsimony (sprparsimony.cpp:2285),
 * both switched by o->tie_mode below.
 *
 * Data model (ours): node records rec = 3*number + slot, back[rec], x[rec]
 * (the xPars flag, pll.h:622-660), one vector per node vec[number][S][W] as in
 * parsVect (sprparsimony.cpp:732-734).
 */
#include "fitch_oracle.h"
#include "rng.h"
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <assert.h>

struct orc {
  int n, P, S, W, datatype, keep_all, nrec;
  unsigned char *codes;
  int *wgt, *inf, ninf;
  uint32_t *vec;
  unsigned *score;
  uint32_t *persite;          /* [2n][W*32] or NULL */
  int persite_on;
  /* Sankoff (weighted) mode: cost matrix given (reference pllCostMatrix != NULL) */
  int sankoff;
  uint32_t *cost;             /* [S][S], cost[i*S+j] = cost i -> j (sprparsimony.cpp:190) */
  uint32_t highest_cost;      /* max(cost) + 1 (sprparsimony.cpp:160) */
  int Pinf;                   /* informative patterns kept (not weight-replicated) */
  uint32_t *svec;             /* [2n][S][Pinf] state costs per pattern */
  uint32_t *pwgt;             /* [Pinf] informativePtnWgt */
  uint32_t *pscore;           /* [Pinf] informativePtnScore (per-pattern score of the last evaluate) */
  int *back;
  unsigned char *x;
  int *nodep;
  int start, ntips, nextnode;
  int *ti;
  unsigned best;
  int insert_rec, remove_rec;
  unsigned long hits;         /* bestTreeScoreHits */
  int tie_mode;
  orc_lcg64 rng;
  double (*rand_fn)(void *);
  void *rand_arg;
  long randum_seed;
  int trace_on, trace_len, trace_cap;
  int *trace_q;
  unsigned *trace_mp;
  int moves_len, moves_cap;
  int *moves_rem, *moves_ins;
  unsigned *moves_score;
  unsigned long long c_newview, c_eval, c_test;
  /* UFBoot-MP online bookkeeping (IQTree::saveCurrentTree, iqtree.cpp:3271-3785, default options) */
  int pre_eval;               /* -1 = as the variant does (mpboot yes, PLL original no); 0 / 1 = forced */
  long max_visits;            /* test aid: orc_optimize_spr returns after this many prune-node visits (0 = no limit) */
  int ufb_on, ufb_B, ufb_bad;
  int ufb_ratchet;                   /* on_ratchet_hclimb1: other weights than the attach-time ones in force */
  int ufb_ratchet_booking;           /* !params->no_hclimb1_bb (tools.cpp:795) */
  int *ufb_w0;                       /* pattern weights at attach time */
  unsigned short *ufb_samples;    /* boot_samples_pars [B][P] */
  double ufb_eps, ufb_cutoff;     /* params->ufboot_epsilon (0.5, tools.cpp:725), logl_cutoff */
  double *ufb_logl;               /* boot_logl */
  int *ufb_counts, *ufb_trees;    /* boot_counts, boot_trees */
  int *ufb_orig;                  /* boot_tree_orig_logl (iqtree.h:766; -cutoff_from_btrees): the logl under which each sample's tree was booked */
  int ufb_cut_btrees;             /* params->cutoff_from_btrees (tools.cpp:2442) */
  double *ufb_treels;             /* treels_logl */
  int ufb_ntrees, ufb_treels_cap;
  int *ufb_store_idx, **ufb_store_back, ufb_nstore, ufb_store_cap;   /* topologies of the trees some sample accepted */
  unsigned short *ufb_ptn;
  unsigned long long ufb_draws;
  /* -mulhits (params->multiple_hits, iqtree.cpp:3498-3540): treels (canonical topology -> tree index) and
     boot_trees_parsimony (per sample: the set of tree indices that reach its best REPS) */
  int ufb_mulhits;
  int ufb_store_trees, ufb_dups;              /* params->store_candidate_trees, duplication_counter */
  unsigned *ufb_key_hash;
  /* -mulhits -topboot N (params->store_top_boot_trees, iqtree.cpp:3542-3585): per sample the N best NEW trees, sorted by
     decreasing REPS, and boot_threshold */
  int ufb_topboot;
  int *ufb_top_idx, *ufb_top_rell, *ufb_top_n, *ufb_thr;
  /* -distinct_iter_top_boot k (iqtree.cpp:3587-3680, without -mulhits): the same arrays + the iteration each entry stands for */
  int ufb_distinct, ufb_cur_it;
  int *ufb_top_iter;
  int **ufb_keys, *ufb_key_idx, ufb_nkeys, ufb_keys_cap;
  int **ufb_set, *ufb_set_n, *ufb_set_cap;
};

#define NUM(r) ((r) / 3)
#define TIP(o, r) (NUM(r) <= (o)->n)
static inline int NX(int r) { int v = r / 3, s = r % 3; return 3 * v + (s + 1) % 3; }

/* PLL tip code -> state set. DNA: bitVectorIdentity; AA: bitVectorAA (pllrepo/src/globalVariables.h:60-78) */
static uint32_t state_mask(int datatype, unsigned code)
{
  if (datatype == ORC_DNA || datatype == ORC_BIN) return code;          /* bitVectorIdentity (binary: 1, 2, 3) */
  if (datatype == ORC_GENERIC) return code < 32 ? 1u << code : 0xFFFFFFFFu;   /* bitVector32, globalVariables.h:98-102 */
  if (code < 20) return 1u << code;
  if (code == 20) return 12u;        /* B = N|D */
  if (code == 21) return 96u;        /* Z = Q|E */
  return 1048575u;                   /* 22: - ? * X */
}
/* globalVariables.h pLengths: undetermined code per data type (binary 3, DNA 15, protein 22, 32-state 32) */
static int undetermined_code(int datatype) { return datatype == ORC_DNA ? 15 : datatype == ORC_BIN ? 3 : datatype == ORC_GENERIC ? 32 : 22; }
static int states_of(int datatype) { return datatype == ORC_DNA ? 4 : datatype == ORC_BIN ? 2 : datatype == ORC_GENERIC ? 32 : 20; }

static double tie_draw(orc *o)
{
  if (o->rand_fn) return o->rand_fn(o->rand_arg);
  return orc_lcg64_next(&o->rng);
}

/* ---- isInformative / determineUninformativeSites: sprparsimony.cpp:2460-2499, :2596-2634 ---- */
static int is_informative(const orc *o, int site)
{
  int check[256], j, cnt = 0, und = undetermined_code(o->datatype);
  if (o->keep_all) return 1;                      /* !sort_alignment, :2462-2463 */
  memset(check, 0, sizeof check);
  for (j = 0; j < o->n; j++) check[o->codes[(size_t)j * o->P + site]] = 1;
  for (j = 0; j < und; j++) if (check[j]) cnt++;
  return cnt > 1;
}

/* ---- compressDNA: sprparsimony.cpp:2828-2973 (PLL: fastDNAparsimony.c:1647-1774) ---- */
/* ---- compressSankoffDNA: sprparsimony.cpp:2636-2825 (values per pattern and state: 0 if the state is in the
        tip's set, highest_cost otherwise; patterns are NOT weight-replicated, weights go to informativePtnWgt).
        The SIMD block interleaving of the reference (:2738-2764) is a layout detail we do not restate;
        arithmetic is exact 32-bit (the reference's -short_off mode; identical to its 16-bit default
        whenever that does not wrap, SURVEY parity hazard 4). ---- */
static void pack_tips_sankoff(orc *o)
{
  int i, k, site, S = o->S, j;
  o->ninf = 0;
  for (site = 0; site < o->P; site++) {
    o->inf[site] = is_informative(o, site);
    if (o->inf[site]) o->ninf++;
  }
  o->Pinf = o->ninf;
  o->W = o->Pinf;                              /* parsimonyLength counts patterns in this mode (:2805) */
  free(o->svec); free(o->pwgt); free(o->pscore);
  o->svec = (uint32_t *)calloc((size_t)2 * o->n * S * (o->Pinf + 1) + 1, sizeof(uint32_t));
  o->pwgt = (uint32_t *)calloc((size_t)o->Pinf + 1, sizeof(uint32_t));
  o->pscore = (uint32_t *)calloc((size_t)o->Pinf + 1, sizeof(uint32_t));
  for (i = 0; i < o->n; i++) {
    uint32_t *tip = o->svec + (size_t)S * o->Pinf * (i + 1);
    for (site = 0, j = 0; site < o->P; site++) {
      uint32_t m;
      if (!o->inf[site]) continue;
      m = state_mask(o->datatype, o->codes[(size_t)i * o->P + site]);
      for (k = 0; k < S; k++) tip[(size_t)k * o->Pinf + j] = (m & (1u << k)) ? 0u : o->highest_cost;   /* :2748-2756 */
      if (i == 0) o->pwgt[j] = (uint32_t)o->wgt[site];
      j++;
    }
  }
  memset(o->score, 0, sizeof(unsigned) * 2 * o->n);
}

static void pack_tips(orc *o)
{
  size_t entries = 0, ce, cep;
  int i, k, site, S = o->S;
  if (o->sankoff) { pack_tips_sankoff(o); return; }
  o->ninf = 0;
  for (site = 0; site < o->P; site++) {
    o->inf[site] = is_informative(o, site);
    if (o->inf[site]) { entries += (size_t)o->wgt[site]; o->ninf++; }
  }
  ce = entries / 32 + (entries % 32 != 0);
  cep = (ce % 8) ? ce + (8 - ce % 8) : ce;        /* INTS_PER_VECTOR = 8 (AVX), :2870-2882 */
  if ((int)cep != o->W || !o->vec) {
    free(o->vec);
    o->W = (int)cep;
    o->vec = (uint32_t *)calloc((size_t)2 * o->n * S * o->W + 1, sizeof(uint32_t));
    free(o->persite);
    o->persite = NULL;
  } else {
    memset(o->vec, 0, sizeof(uint32_t) * 2 * o->n * S * o->W);
  }
  if (o->persite_on && !o->persite)
    o->persite = (uint32_t *)calloc((size_t)2 * o->n * o->W * 32 + 1, sizeof(uint32_t));
  for (i = 0; i < o->n; i++) {
    uint32_t *tip = o->vec + (size_t)o->W * S * (i + 1), val[32];
    size_t ci = 0;
    int cc = 0, w;
    memset(val, 0, sizeof val);
    for (site = 0; site < o->P; site++) {
      uint32_t m;
      if (!o->inf[site]) continue;
      m = state_mask(o->datatype, o->codes[(size_t)i * o->P + site]);
      for (w = 0; w < o->wgt[site]; w++) {
        for (k = 0; k < S; k++) if (m & (1u << k)) val[k] |= 1u << cc;
        if (++cc == 32) {
          for (k = 0; k < S; k++) { tip[(size_t)k * o->W + ci] = val[k]; val[k] = 0; }
          cc = 0; ci++;
        }
      }
    }
    for (; ci < (size_t)o->W; ci++) {               /* padding bits all ones, :2947-2960 */
      for (; cc < 32; cc++) for (k = 0; k < S; k++) val[k] |= 1u << cc;
      for (k = 0; k < S; k++) { tip[(size_t)k * o->W + ci] = val[k]; val[k] = 0; }
      cc = 0;
    }
  }
  memset(o->score, 0, sizeof(unsigned) * 2 * o->n);
}

orc *orc_create(int n, int P, int datatype, const unsigned char *codes, const int *weights, int keep_all)
{
  orc *o = (orc *)calloc(1, sizeof(orc));
  int i;
  o->n = n; o->P = P; o->datatype = datatype; o->keep_all = keep_all;
  o->pre_eval = -1;
  o->S = states_of(datatype);
  o->nrec = 3 * (2 * n - 1) + 3;
  o->codes = (unsigned char *)malloc((size_t)n * P);
  memcpy(o->codes, codes, (size_t)n * P);
  o->wgt = (int *)malloc(sizeof(int) * P);
  memcpy(o->wgt, weights, sizeof(int) * P);
  o->inf = (int *)calloc(P, sizeof(int));
  o->score = (unsigned *)calloc(2 * n, sizeof(unsigned));
  o->back = (int *)malloc(sizeof(int) * o->nrec);
  for (i = 0; i < o->nrec; i++) o->back[i] = -1;
  o->x = (unsigned char *)calloc(o->nrec, 1);
  o->nodep = (int *)calloc(2 * n, sizeof(int));
  o->ti = (int *)calloc((size_t)4 * n + 8, sizeof(int));
  o->W = -1;
  pack_tips(o);
  orc_reset_nodep(o);
  orc_lcg64_init(&o->rng, 1);
  o->tie_mode = ORC_TIE_FIRST;
  o->randum_seed = 12345;
  o->start = 3;
  o->ntips = n;
  return o;
}

/* Sankoff engine: cost[S*S] as loaded by ParsTree::loadCostMatrixFile (parstree.cpp:31-95), which also closes
   it under the triangle inequality (:74-80); initializeCostMatrix (sprparsimony.cpp:159-188) */
orc *orc_create_sankoff(int n, int P, int datatype, const unsigned char *codes, const int *weights, int keep_all,
                        const unsigned *cost)
{
  orc *o = orc_create(n, P, datatype, codes, weights, keep_all);
  int S = o->S, i, j, k;
  o->cost = (uint32_t *)malloc(sizeof(uint32_t) * S * S);
  for (i = 0; i < S * S; i++) o->cost[i] = cost[i];
  for (k = 0; k < S; k++)
    for (i = 0; i < S; i++)
      for (j = 0; j < S; j++)
        if (o->cost[i * S + j] > o->cost[i * S + k] + o->cost[k * S + j]) o->cost[i * S + j] = o->cost[i * S + k] + o->cost[k * S + j];
  o->highest_cost = 0;
  for (i = 0; i < S * S; i++) if (o->cost[i] > o->highest_cost) o->highest_cost = o->cost[i];
  o->highest_cost += 1;
  o->sankoff = 1;
  pack_tips(o);
  return o;
}

void orc_ufboot_detach(orc *o);
void orc_destroy(orc *o)
{
  if (!o) return;
  orc_ufboot_detach(o);
  free(o->cost); free(o->svec); free(o->pwgt); free(o->pscore);
  free(o->codes); free(o->wgt); free(o->inf); free(o->vec); free(o->score); free(o->persite);
  free(o->back); free(o->x); free(o->nodep); free(o->ti); free(o->trace_q); free(o->trace_mp);
  free(o->moves_rem); free(o->moves_ins); free(o->moves_score);
  free(o);
}

int orc_words(const orc *o) { return o->W; }
int orc_states(const orc *o) { return o->S; }
int orc_num_informative(const orc *o) { return o->ninf; }
const int *orc_informative(const orc *o) { return o->inf; }
const uint32_t *orc_node_vector(const orc *o, int node) { return o->vec + (size_t)o->W * o->S * node; }

/* orientation flags as _allocateParsimonyDataStructures leaves them: sprparsimony.cpp:3047-3055 */
static void reset_flags(orc *o)
{
  int i;
  for (i = o->n + 1; i <= 2 * o->n - 1; i++) {
    int p = o->nodep[i];
    o->x[p] = 1; o->x[NX(p)] = 0; o->x[NX(NX(p))] = 0;
  }
}

/* _updateInternalPllOnRatchet + _allocateParsimonyDataStructures: sprparsimony.cpp:3022-3060 */
void orc_set_weights(orc *o, const int *weights)
{
  memcpy(o->wgt, weights, sizeof(int) * o->P);
  pack_tips(o);
  reset_flags(o);
  /* an attached UFBoot tracker: under other weights than the attach-time ones (the perturbed alignment of a ratchet
     iteration, iqtree.cpp:1694-1716) saveCurrentTree keeps running unless -no_hclimb1_bb (:3280), with the cur_logl of
     :3283-3295.  Weights that take an attach-time pattern's last site away (weight 0; mpboot's ratchet only adds copies
     of sites, alignment.cpp:1915-1969) are outside what the reference can produce: the tracker rests until they go. */
  if (o->ufb_samples) {
    int k, other = memcmp(o->wgt, o->ufb_w0, sizeof(int) * o->P) != 0, lost = 0;
    if (other) for (k = 0; k < o->P; k++) if (o->ufb_w0[k] > 0 && o->wgt[k] <= 0) lost = 1;
    o->ufb_on = !other || (!lost && o->ufb_ratchet_booking);
    o->ufb_ratchet = other && o->ufb_on;
  }
  if (o->persite) { free(o->persite); o->persite = NULL; }
  if (o->ufb_samples) o->persite_on = o->ufb_on;
  if (o->persite_on) orc_enable_persite(o, 1);
}

void orc_enable_persite(orc *o, int on)
{
  o->persite_on = on;
  if (on && !o->persite) o->persite = (uint32_t *)calloc((size_t)2 * o->n * o->W * 32 + 1, sizeof(uint32_t));
}

/* nodep / orientation flags as left by pllTreeInitDefaults (utils.c:2019-2044) + _allocate (:3047-3055) */
void orc_reset_nodep(orc *o)
{
  int i;
  memset(o->x, 0, o->nrec);
  for (i = 1; i <= 2 * o->n - 1; i++) {
    o->nodep[i] = 3 * i;
    if (i > o->n) o->x[3 * i] = 1;
  }
}

void orc_get_nodep(const orc *o, int *nodep) { memcpy(nodep, o->nodep, sizeof(int) * 2 * o->n); }

void orc_set_tree(orc *o, const int *back)
{
  memcpy(o->back, back, sizeof(int) * 3 * (2 * o->n - 1));
  o->start = o->nodep[1];
  o->ntips = o->n;
}
void orc_get_tree(const orc *o, int *back) { memcpy(back, o->back, sizeof(int) * 3 * (2 * o->n - 1)); }

static void hookup(orc *o, int p, int q) { o->back[p] = q; o->back[q] = p; }  /* hookupDefault utils.c:456 */

/* ---- per-site helpers: sprparsimony.cpp:294-376 ---- */
static void ps_reset(orc *o, int node) { memset(o->persite + (size_t)o->W * 32 * node, 0, sizeof(uint32_t) * o->W * 32); }
static void ps_add(orc *o, int p, int q, int r)
{
  size_t L = (size_t)o->W * 32, k;
  uint32_t *pb = o->persite + L * p, *qb = o->persite + L * q, *rb = o->persite + L * r;
  for (k = 0; k < L; k++) pb[k] += qb[k] + rb[k];
}
static void ps_store(orc *o, uint32_t vN, int word, int node)
{
  uint32_t *b = o->persite + (size_t)o->W * 32 * node + (size_t)word * 32;
  int j;
  for (j = 0; j < 32; j++) b[j] += (vN >> j) & 1u;
}

/* ---- getxnodeLocal / computeTraversalInfoParsimony: sprparsimony.cpp:420-467 ---- */
static void getx(orc *o, int p)
{
  int s = NX(p);
  if (o->x[s] || o->x[(s = NX(s))]) { o->x[p] = o->x[s]; o->x[s] = 0; }
  assert(o->x[p] || o->x[NX(p)] || o->x[NX(NX(p))]);
}

static void traversal(orc *o, int p, int *counter, int full)
{
  int q, r;
  if (o->persite_on && !o->sankoff) ps_reset(o, NUM(p));   /* :437-439 (perSiteScores && pllCostMatrix == NULL) */
  q = o->back[NX(p)];
  r = o->back[NX(NX(p))];
  if (!o->x[p]) getx(o, p);
  if (full) {
    if (!TIP(o, q)) traversal(o, q, counter, full);
    if (!TIP(o, r)) traversal(o, r, counter, full);
  } else {
    if (!TIP(o, q) && !o->x[q]) traversal(o, q, counter, full);
    if (!TIP(o, r) && !o->x[r]) traversal(o, r, counter, full);
  }
  o->ti[*counter] = NUM(p);
  o->ti[*counter + 1] = NUM(q);
  o->ti[*counter + 2] = NUM(r);
  *counter += 4;
}

/* ---- newviewParsimonyIterativeFast, Fitch: sprparsimony.cpp:554-878 (generic-S body :841-869) ---- */
/* ---- newviewSankoffParsimonyIterativeFastSIMD: sprparsimony.cpp:477-551 ----
        cur[z] = min_x(left[x] + cost[z][x]) + min_x(right[x] + cost[z][x]);
        parsimonyScore[p] = sum over patterns of min_z cur[z] (unweighted; only tested > 0, :544-548, :3014) */
static void newview_iter_sankoff(orc *o)
{
  int idx, count = o->ti[0], S = o->S, P = o->Pinf, z, x, i;
  for (idx = 4; idx < count; idx += 4) {
    size_t pN = (size_t)o->ti[idx], qN = (size_t)o->ti[idx + 1], rN = (size_t)o->ti[idx + 2];
    const uint32_t *L = o->svec + (size_t)S * P * qN, *R = o->svec + (size_t)S * P * rN;
    uint32_t *C = o->svec + (size_t)S * P * pN;
    unsigned total = 0;
    for (i = 0; i < P; i++) {
      uint32_t cur_contrib = UINT32_MAX;
      for (z = 0; z < S; z++) {
        const uint32_t *c = o->cost + (size_t)z * S;
        uint32_t lc = L[i] + c[0], rc = R[i] + c[0];
        for (x = 1; x < S; x++) {
          uint32_t v = L[(size_t)x * P + i] + c[x];
          if (v < lc) lc = v;
          v = R[(size_t)x * P + i] + c[x];
          if (v < rc) rc = v;
        }
        C[(size_t)z * P + i] = lc + rc;
        if (lc + rc < cur_contrib) cur_contrib = lc + rc;
      }
      total += cur_contrib;
    }
    o->score[pN] = total;
    o->c_newview++;
  }
}

/* ---- evaluateSankoffParsimonyIterativeFastSIMD: sprparsimony.cpp:880-961 ----
        sum over patterns of w * min_x(left[x] + min_y(cost[x][y] + right[y])), left = q, right = p.
        The segment-wise early return of a lower-bound estimate (:946-955) is not restated: it only changes
        the value reported for moves that are rejected anyway (SURVEY parity hazard 5). */
static unsigned evaluate_iter_sankoff(orc *o)
{
  size_t pN = (size_t)o->ti[1], qN = (size_t)o->ti[2];
  int S = o->S, P = o->Pinf, x, y, i;
  const uint32_t *L, *R;
  unsigned total = 0;
  if (o->ti[0] > 4) newview_iter_sankoff(o);
  L = o->svec + (size_t)S * P * qN;
  R = o->svec + (size_t)S * P * pN;
  for (i = 0; i < P; i++) {
    uint32_t best = UINT32_MAX;
    for (x = 0; x < S; x++) {
      const uint32_t *c = o->cost + (size_t)x * S;
      uint32_t t = c[0] + R[i];
      for (y = 1; y < S; y++) {
        uint32_t v = c[y] + R[(size_t)y * P + i];
        if (v < t) t = v;
      }
      t += L[(size_t)x * P + i];
      if (t < best) best = t;
    }
    o->pscore[i] = best;
    total += best * o->pwgt[i];
  }
  o->c_eval++;
  return total;
}

static void newview_iter(orc *o)
{
  int idx, count = o->ti[0], S = o->S, W = o->W, k, i;
  if (o->sankoff) { newview_iter_sankoff(o); return; }
  for (idx = 4; idx < count; idx += 4) {
    size_t pN = (size_t)o->ti[idx], qN = (size_t)o->ti[idx + 1], rN = (size_t)o->ti[idx + 2];
    const uint32_t *L = o->vec + (size_t)W * S * qN, *R = o->vec + (size_t)W * S * rN;
    uint32_t *C = o->vec + (size_t)W * S * pN;
    unsigned total = 0;
    if (o->persite_on) {                              /* :660-663 */
      if ((int)qN <= o->n) ps_reset(o, (int)qN);
      if ((int)rN <= o->n) ps_reset(o, (int)rN);
    }
    for (i = 0; i < W; i++) {
      uint32_t lA[32], vA[32], vN = 0;
      for (k = 0; k < S; k++) {
        uint32_t sl = L[(size_t)k * W + i], sr = R[(size_t)k * W + i];
        lA[k] = sl & sr; vA[k] = sl | sr; vN |= lA[k];
      }
      for (k = 0; k < S; k++) C[(size_t)k * W + i] = lA[k] | (~vN & vA[k]);
      vN = ~vN;
      total += (unsigned)__builtin_popcount(vN);
      if (o->persite_on) ps_store(o, vN, i, (int)pN);
    }
    o->score[pN] = total + o->score[rN] + o->score[qN];   /* :874 */
    if (o->persite_on) ps_add(o, (int)pN, (int)qN, (int)rN);
    o->c_newview++;
  }
}

/* ---- evaluateParsimonyIterativeFast, Fitch: sprparsimony.cpp:965-1206 (no early exit, :1122-1123) ---- */
static unsigned evaluate_iter(orc *o)
{
  size_t pN = (size_t)o->ti[1], qN = (size_t)o->ti[2];
  int S = o->S, W = o->W, k, i;
  unsigned sum;
  const uint32_t *L, *R;
  if (o->sankoff) return evaluate_iter_sankoff(o);
  if (o->ti[0] > 4) newview_iter(o);
  sum = o->score[pN] + o->score[qN];
  if (o->persite_on) {                                /* :1051-1054 */
    ps_reset(o, NUM(o->start));
    ps_add(o, NUM(o->start), (int)pN, (int)qN);
  }
  L = o->vec + (size_t)W * S * qN;
  R = o->vec + (size_t)W * S * pN;
  for (i = 0; i < W; i++) {
    uint32_t vN = 0;
    for (k = 0; k < S; k++) vN |= L[(size_t)k * W + i] & R[(size_t)k * W + i];
    vN = ~vN;
    sum += (unsigned)__builtin_popcount(vN);
    if (o->persite_on) ps_store(o, vN, i, NUM(o->start));
  }
  o->c_eval++;
  return sum;
}

/* ---- evaluateParsimony / newviewParsimony: sprparsimony.cpp:1889-1934 ---- */
unsigned orc_evaluate(orc *o, int p, int full)
{
  int q = o->back[p], counter = 4;
  o->ti[1] = NUM(p);
  o->ti[2] = NUM(q);
  if (full) {
    if (!TIP(o, p)) traversal(o, p, &counter, full);
    if (!TIP(o, q)) traversal(o, q, &counter, full);
  } else {
    if (!TIP(o, p) && !o->x[p]) traversal(o, p, &counter, full);
    if (!TIP(o, q) && !o->x[q]) traversal(o, q, &counter, full);
  }
  o->ti[0] = counter;
  return evaluate_iter(o);
}

static void newview(orc *o, int p)
{
  int counter = 4;
  if (TIP(o, p)) return;
  traversal(o, p, &counter, 0);
  o->ti[0] = counter;
  newview_iter(o);
}

/* ---- reorderNodes / nodeRectifierPars: sprparsimony.cpp:2046-2101 ---- */
static void reorder(orc *o, int p, int *count)
{
  if (TIP(o, p)) return;
  o->nodep[*count + o->n + 1] = p;     /* the record reached from the parent */
  (*count)++;
  reorder(o, o->back[NX(p)], count);
  reorder(o, o->back[NX(NX(p))], count);
}
void orc_node_rectifier(orc *o)
{
  int count = 0;
  o->start = o->nodep[1];
  reorder(o, o->back[o->start], &count);
}

unsigned orc_score_tree(orc *o)
{
  orc_node_rectifier(o);
  return orc_evaluate(o, o->start, 1);
}

/* ---- pllComputePatternParsimony: sprparsimony.cpp:3363-3392 ---- */
int orc_pattern_scores(orc *o, unsigned short *ptn)
{
  const uint32_t *p;
  int k, site = 0, sum = 0, upper = o->keep_all ? o->P : o->ninf, j = 0;
  if (o->sankoff) {                                   /* pllComputeSankoffPatternParsimony :3341-3355 */
    for (k = 0; k < o->P; k++) ptn[k] = 0;
    for (k = 0; k < o->P; k++) {
      if (!o->inf[k]) continue;
      ptn[k] = (unsigned short)o->pscore[j];
      sum += (int)ptn[k] * o->wgt[k];
      j++;
    }
    return sum;
  }
  p = o->persite + (size_t)o->W * 32 * NUM(o->start);
  /* the reference indexes ptn by sorted-pattern position and assumes the first
     numInformativePatterns patterns are the kept ones (:3380-3387); we return the
     score at each KEPT pattern's original index and 0 elsewhere, which is the same
     thing whenever the reference's assumption holds. */
  for (k = 0; k < o->P; k++) ptn[k] = 0;
  for (k = 0; k < o->P && j < upper; k++) {
    if (!o->inf[k]) continue;
    ptn[k] = (unsigned short)p[site];
    sum += (int)ptn[k] * o->wgt[k];
    site += o->wgt[k];
    j++;
  }
  return sum;
}

/* ---- pllComputeSiteParsimony: sprparsimony.cpp:3403-3450 (the per-site row of tr->start, then zeros) ---- */
int orc_site_scores(orc *o, int *site_pars, int nsite)
{
  const uint32_t *p = o->persite + (size_t)o->W * 32 * NUM(o->start);
  int k, site = 0, sum = 0, width = 0;
  for (k = 0; k < o->P; k++) if (o->inf[k]) width += o->wgt[k];
  for (k = 0; k < width && site < nsite; k++) { site_pars[site] = (int)p[k]; sum += site_pars[site]; site++; }
  for (; site < nsite; site++) site_pars[site] = 0;
  return sum;
}

void orc_seed_ties(orc *o, int tie_mode, int seed)
{
  o->tie_mode = tie_mode;
  orc_lcg64_init(&o->rng, seed);
}
void orc_set_tie_state(orc *o, unsigned long long state) { o->rng.state = state; }
unsigned long long orc_get_tie_state(const orc *o) { return o->rng.state; }
void orc_set_rand_callback(orc *o, double (*fn)(void *), void *arg) { o->rand_fn = fn; o->rand_arg = arg; }
void orc_set_pre_evaluate(orc *o, int mode) { o->pre_eval = mode; }
void orc_set_max_visits(orc *o, long k) { o->max_visits = k; }

/* ---- traces ---- */
void orc_trace(orc *o, int on) { o->trace_on = on; o->trace_len = 0; o->moves_len = 0; }
int orc_trace_len(const orc *o) { return o->trace_len; }
void orc_trace_get(const orc *o, int *q, unsigned *mp)
{
  memcpy(q, o->trace_q, sizeof(int) * o->trace_len);
  memcpy(mp, o->trace_mp, sizeof(unsigned) * o->trace_len);
}
static void trace_push(orc *o, int q, unsigned mp)
{
  if (!o->trace_on) return;
  if (o->trace_len == o->trace_cap) {
    o->trace_cap = o->trace_cap ? 2 * o->trace_cap : 1024;
    o->trace_q = (int *)realloc(o->trace_q, sizeof(int) * o->trace_cap);
    o->trace_mp = (unsigned *)realloc(o->trace_mp, sizeof(unsigned) * o->trace_cap);
  }
  o->trace_q[o->trace_len] = q;
  o->trace_mp[o->trace_len++] = mp;
}
int orc_moves_len(const orc *o) { return o->moves_len; }
void orc_moves_get(const orc *o, int *rem, int *ins, unsigned *score)
{
  memcpy(rem, o->moves_rem, sizeof(int) * o->moves_len);
  memcpy(ins, o->moves_ins, sizeof(int) * o->moves_len);
  memcpy(score, o->moves_score, sizeof(unsigned) * o->moves_len);
}
static void moves_push(orc *o, int rem, int ins, unsigned score)
{
  if (!o->trace_on) return;
  if (o->moves_len == o->moves_cap) {
    o->moves_cap = o->moves_cap ? 2 * o->moves_cap : 256;
    o->moves_rem = (int *)realloc(o->moves_rem, sizeof(int) * o->moves_cap);
    o->moves_ins = (int *)realloc(o->moves_ins, sizeof(int) * o->moves_cap);
    o->moves_score = (unsigned *)realloc(o->moves_score, sizeof(unsigned) * o->moves_cap);
  }
  o->moves_rem[o->moves_len] = rem;
  o->moves_ins[o->moves_len] = ins;
  o->moves_score[o->moves_len++] = score;
}

/* ---- IQTree::saveCurrentTree, maximum-parsimony branch with the default options (iqtree.cpp:3271-3785):
        store_candidate_trees off (tools.cpp:736) => no string lookup before the cut-off test (:3343), only when a
        sample accepts the tree (:3689-3707: treels maps the sorted tree string to the index of the first accepted
        tree of that topology);
        pattern scores from pllComputePatternParsimony (:3365); REPS as the exact integer sum (the auto_vectorize
        loop :3418-3422; the Vec16us segment sums :3423-3449 are the same number whenever they do not wrap, and the
        remain-bound skip :3435-3445 only skips samples for which neither branch below can fire); then the DEFAULT
        update rule (:3684-3731).  The tree "string" of the reference is kept here as the back[] of the tentatively
        inserted topology. ---- */
static void ufb_store_tree(orc *o, int tree_index)
{
  int nrec = 3 * (2 * o->n - 1);
  if (o->ufb_nstore && o->ufb_store_idx[o->ufb_nstore - 1] == tree_index) return;
  { int i; for (i = 0; i < o->ufb_nstore; i++) if (o->ufb_store_idx[i] == tree_index) return; }
  if (o->ufb_nstore == o->ufb_store_cap) {
    o->ufb_store_cap = o->ufb_store_cap ? 2 * o->ufb_store_cap : 64;
    o->ufb_store_idx = (int *)realloc(o->ufb_store_idx, sizeof(int) * o->ufb_store_cap);
    o->ufb_store_back = (int **)realloc(o->ufb_store_back, sizeof(int *) * o->ufb_store_cap);
  }
  o->ufb_store_idx[o->ufb_nstore] = tree_index;
  o->ufb_store_back[o->ufb_nstore] = (int *)malloc(sizeof(int) * nrec);
  memcpy(o->ufb_store_back[o->ufb_nstore], o->back, sizeof(int) * nrec);
  o->ufb_nstore++;
}

/* REPS inner product (iqtree.cpp:3418-3432).  The reference runs it on Vec16us lanes; so that the CPU baseline is not
   handicapped, an AVX2 build of the same loop is used when the host has it (same integer result). */
static int ufb_dot_scalar(const unsigned short *a, const unsigned short *b, int n)
{
  int res = 0, k;
  for (k = 0; k < n; k++) res += (int)a[k] * (int)b[k];
  return res;
}
__attribute__((target("avx2"), optimize("O3"))) static int ufb_dot_avx2(const unsigned short *a, const unsigned short *b, int n)
{
  int res = 0, k;
  for (k = 0; k < n; k++) res += (int)a[k] * (int)b[k];
  return res;
}
static int ufb_dot(const unsigned short *a, const unsigned short *b, int n)
{
  static int have = -1;
  if (have < 0) have = __builtin_cpu_supports("avx2") ? 1 : 0;
  return have ? ufb_dot_avx2(a, b, n) : ufb_dot_scalar(a, b, n);
}

/* what printTree(WT_TAXON_ID | WT_SORT_TAXA) stands for: a canonical form of the unrooted topology -- the tree hung from
   tip 1, an inner node written as -1 followed by its two subtrees, the one holding the smaller tip number first */
static int canon_min(const orc *o, int r, int *mins)
{
  int a, b;
  if (r / 3 <= o->n) return mins[r] = r / 3;
  a = canon_min(o, o->back[NX(r)], mins);
  b = canon_min(o, o->back[NX(NX(r))], mins);
  return mins[r] = a < b ? a : b;
}
static void canon_emit(const orc *o, int r, const int *mins, int *out, int *k)
{
  int a, b;
  if (r / 3 <= o->n) { out[(*k)++] = r / 3; return; }
  out[(*k)++] = -1;
  a = o->back[NX(r)]; b = o->back[NX(NX(r))];
  if (mins[a] > mins[b]) { int t = a; a = b; b = t; }
  canon_emit(o, a, mins, out, k);
  canon_emit(o, b, mins, out, k);
}
/* treels.find(tree_str) / treels[tree_str] = tree_index (iqtree.cpp:3503-3513; with -storetrees :3302-3311, :3346) */
static int *ufb_topology_key(const orc *o)
{
  int len = 2 * o->n - 2, k = 0;
  int *mins = (int *)malloc(sizeof(int) * 3 * (2 * o->n - 1)), *key = (int *)malloc(sizeof(int) * len);
  canon_min(o, o->back[3], mins);
  canon_emit(o, o->back[3], mins, key, &k);
  free(mins);
  return key;
}
static unsigned ufb_key_hash(const orc *o, const int *key)
{
  unsigned h = 2166136261u;
  int i;
  for (i = 0; i < 2 * o->n - 3; i++) h = (h ^ (unsigned)key[i]) * 16777619u;     /* n - 1 tips + n - 2 inner nodes */
  return h;
}
static int ufb_find_key(const orc *o, const int *key)
{
  const unsigned h = ufb_key_hash(o, key);
  int i;
  for (i = 0; i < o->ufb_nkeys; i++)
    if (o->ufb_key_hash[i] == h && memcmp(o->ufb_keys[i], key, sizeof(int) * (size_t)(2 * o->n - 3)) == 0) return o->ufb_key_idx[i];
  return -1;
}
static void ufb_add_key(orc *o, int *key, int tree_index)
{
  if (o->ufb_nkeys == o->ufb_keys_cap) {
    o->ufb_keys_cap = o->ufb_keys_cap ? 2 * o->ufb_keys_cap : 64;
    o->ufb_keys = (int **)realloc(o->ufb_keys, sizeof(int *) * o->ufb_keys_cap);
    o->ufb_key_idx = (int *)realloc(o->ufb_key_idx, sizeof(int) * o->ufb_keys_cap);
    o->ufb_key_hash = (unsigned *)realloc(o->ufb_key_hash, sizeof(unsigned) * o->ufb_keys_cap);
  }
  o->ufb_keys[o->ufb_nkeys] = key;
  o->ufb_key_hash[o->ufb_nkeys] = ufb_key_hash(o, key);
  o->ufb_key_idx[o->ufb_nkeys++] = tree_index;
}
static int ufb_lookup_topology(orc *o, int tree_index)
{
  int *key = ufb_topology_key(o);
  const int found = ufb_find_key(o, key);
  if (found >= 0) { free(key); return found; }
  ufb_add_key(o, key, tree_index);
  return tree_index;
}

static void ufb_save_current_tree(orc *o, double cur_logl)
{
  int tree_index = -1, sample, test_pars, looked_up = 0;
  int *key = NULL;
  if (o->ufb_ratchet) {
    /* :3283-3295 "if on_ratchet_hclimb1, update cur_logl": REPS of _pattern_pars -- as the array stands, i.e. still
       holding the tree of the previous call that got past the filter below (or what the IQ-TREE kernel left there for
       the climb's start tree) -- against original_sample; the segments cover the informative patterns */
    int k, score = 0;
    for (k = 0; k < o->P; k++) if (o->inf[k]) score += (int)o->ufb_ptn[k] * o->ufb_w0[k];
    cur_logl = -(double)score;
  }
  if (o->ufb_store_trees) {
    /* :3302-3311 -storetrees: every tree that comes here is looked up by its topology first */
    key = ufb_topology_key(o);
    tree_index = ufb_find_key(o, key);
    looked_up = 1;                                                           /* tree_str is set: no lookup further down */
  }
  if (tree_index >= 0) {                                                     /* :3313-3341 already in treels */
    free(key);
    o->ufb_dups++;
    if (cur_logl <= o->ufb_treels[tree_index] + 1e-4) return;
    o->ufb_treels[tree_index] = cur_logl;                                    /* (and on: no cut-off test on this side) */
  } else {
    if (o->ufb_cutoff != 0.0 && cur_logl <= o->ufb_cutoff - 1e-4) { free(key); return; }     /* :3343 */
    tree_index = o->ufb_ntrees;                                              /* :3345-3348 */
    if (key) ufb_add_key(o, key, tree_index);
    if (o->ufb_ntrees == o->ufb_treels_cap) {
      o->ufb_treels_cap = o->ufb_treels_cap ? 2 * o->ufb_treels_cap : 1024;
      o->ufb_treels = (double *)realloc(o->ufb_treels, sizeof(double) * o->ufb_treels_cap);
    }
    o->ufb_treels[o->ufb_ntrees++] = cur_logl;
  }
  test_pars = orc_pattern_scores(o, o->ufb_ptn);                             /* :3365 */
  if (!o->ufb_ratchet && test_pars != -(int)cur_logl) o->ufb_bad++;          /* :3366-3367 outError (not on ratchet climbs) */
  for (sample = 0; sample < o->ufb_B; sample++) {                            /* :3411 */
    const unsigned short *bs = o->ufb_samples + (size_t)sample * o->P;
    int res = ufb_dot(o->ufb_ptn, bs, o->P);
    double rell = -(double)res;
    if (o->ufb_distinct && !o->ufb_mulhits) {                                /* :3587-3680 */
      const int K = o->ufb_distinct;
      int *ti = o->ufb_top_idx + (size_t)sample * K, *tr = o->ufb_top_rell + (size_t)sample * K, *tit = o->ufb_top_iter + (size_t)sample * K;
      int *cnt = &o->ufb_top_n[sample];
      const double thr = (double)o->ufb_thr[sample];
      if (rell >= thr) o->ufb_counts[sample]++;
      if (rell > thr || (rell == thr && (o->ufb_draws++, tie_draw(o)) <= (double)K * 1.0 / (double)o->ufb_counts[sample])) {
        int t, c, exists = 0, d;
        if (rell > o->ufb_logl[sample]) o->ufb_counts[sample] = 1;
        if (!looked_up) { tree_index = ufb_lookup_topology(o, o->ufb_ntrees - 1); looked_up = 1; }
        ufb_store_tree(o, tree_index);
        if (o->ufb_cut_btrees) o->ufb_orig[sample] = (int)cur_logl;          /* :3617-3619 */
        o->ufb_trees[sample] = tree_index;
        if (rell > o->ufb_logl[sample]) o->ufb_logl[sample] = rell;
        t = K < *cnt ? K : *cnt;
        for (c = 0; c < t; c++) if (ti[c] == tree_index) { exists = 1; break; }
        if (exists) continue;
        for (c = 0; c < t; c++)
          if (tit[c] == o->ufb_cur_it) {
            if (rell > (double)tr[c]) { tr[c] = (int)rell; ti[c] = tree_index; }
            break;
          }
        if (c == t && t < K) { tit[*cnt] = o->ufb_cur_it; ti[*cnt] = tree_index; tr[*cnt] = (int)rell; (*cnt)++; }
        else if (c == t && t == K) {
          int worst = 0;
          for (d = 1; d < t; d++) if (tr[d] < tr[worst]) worst = d;
          ti[worst] = tree_index; tr[worst] = (int)rell; tit[worst] = o->ufb_cur_it;
        }
        o->ufb_thr[sample] = tr[0];
        for (d = 1; d < *cnt; d++) if (tr[d] < o->ufb_thr[sample]) o->ufb_thr[sample] = tr[d];
      }
      continue;
    }
    if (o->ufb_mulhits && o->ufb_topboot) {                                  /* :3542-3585 */
      const int N = o->ufb_topboot;
      int *ti = o->ufb_top_idx + (size_t)sample * N, *tr = o->ufb_top_rell + (size_t)sample * N, *cnt = &o->ufb_top_n[sample];
      if (*cnt < N || rell > (double)o->ufb_thr[sample]) {
        if (!looked_up) { tree_index = ufb_lookup_topology(o, o->ufb_ntrees - 1); looked_up = 1; }
        if (tree_index == o->ufb_ntrees - 1) {                                /* "if newly added" */
          int pos, i;
          if (*cnt < N) {
            for (pos = 0; pos < *cnt; pos++) if ((double)tr[pos] < rell) break;
            for (i = *cnt; i > pos; i--) { ti[i] = ti[i - 1]; tr[i] = tr[i - 1]; }
            ti[pos] = tree_index; tr[pos] = (int)rell; (*cnt)++;
            if (!((double)o->ufb_thr[sample] < rell)) o->ufb_thr[sample] = (int)rell;
            ufb_store_tree(o, tree_index);
          } else if (rell > (double)o->ufb_thr[sample]) {
            (*cnt)--;                                                         /* pop_back */
            for (pos = 0; pos < *cnt; pos++) if ((double)tr[pos] < rell) break;
            for (i = *cnt; i > pos; i--) { ti[i] = ti[i - 1]; tr[i] = tr[i - 1]; }
            ti[pos] = tree_index; tr[pos] = (int)rell; (*cnt)++;
            o->ufb_thr[sample] = tr[N - 1];
            ufb_store_tree(o, tree_index);
          }
        }
      }
      continue;
    }
    if (o->ufb_mulhits) {                                                    /* :3498-3540, no draw, no boot_counts */
      if (rell >= o->ufb_logl[sample]) {
        int i, have = 0;
        if (!looked_up) { tree_index = ufb_lookup_topology(o, o->ufb_ntrees - 1); looked_up = 1; }   /* :3500-3514 */
        if (rell > o->ufb_logl[sample]) { o->ufb_set_n[sample] = 0; o->ufb_logl[sample] = rell; }      /* :3516-3519 */
        if (o->ufb_cut_btrees && (int)cur_logl > o->ufb_orig[sample]) o->ufb_orig[sample] = (int)cur_logl;   /* :3523-3527 (the array starts at 0: lengths being negative logls, this never fires -- the reference's own quirk) */
        for (i = 0; i < o->ufb_set_n[sample]; i++) if (o->ufb_set[sample][i] == tree_index) have = 1;
        if (!have) {                                                         /* :3530-3533 */
          if (o->ufb_set_n[sample] == o->ufb_set_cap[sample]) {
            o->ufb_set_cap[sample] = o->ufb_set_cap[sample] ? 2 * o->ufb_set_cap[sample] : 4;
            o->ufb_set[sample] = (int *)realloc(o->ufb_set[sample], sizeof(int) * o->ufb_set_cap[sample]);
          }
          o->ufb_set[sample][o->ufb_set_n[sample]++] = tree_index;
          ufb_store_tree(o, tree_index);
        }
      }
      continue;
    }
    if (rell > o->ufb_logl[sample] + o->ufb_eps ||                           /* :3686-3688 */
        (rell > o->ufb_logl[sample] - o->ufb_eps &&
         (o->ufb_draws++, tie_draw(o)) <= 1.0 / (double)(o->ufb_counts[sample] + 1))) {
      /* :3689-3707: the tree string, once per call; a topology that was accepted before keeps its first index */
      if (!looked_up) { tree_index = ufb_lookup_topology(o, o->ufb_ntrees - 1); looked_up = 1; }
      ufb_store_tree(o, tree_index);
      if (rell > o->ufb_logl[sample]) o->ufb_counts[sample] = 1;             /* :3710-3713 */
      if (o->ufb_cut_btrees) o->ufb_orig[sample] = (int)cur_logl;            /* :3716-3718 */
      if (rell > o->ufb_logl[sample]) o->ufb_logl[sample] = rell;            /* :3719 max() */
      o->ufb_trees[sample] = tree_index;                                     /* :3720 */
    }
    if (rell == o->ufb_logl[sample]) o->ufb_counts[sample]++;                /* :3728-3730 */
  }
}

/* ---- insertParsimony / testInsertParsimony: sprparsimony.cpp:1942-1952, :2106-2188 ---- */
static void insert_node(orc *o, int p, int q)
{
  int r = o->back[q];
  hookup(o, NX(p), q);
  hookup(o, NX(NX(p)), r);
  newview(o, p);
}

static void test_insert(orc *o, int p, int q)
{
  int r = o->back[q];
  unsigned mp;
  insert_node(o, p, q);
  mp = orc_evaluate(o, NX(NX(p)), 0);
  o->c_test++;
  trace_push(o, q, mp);
  if (o->ufb_on) ufb_save_current_tree(o, -(double)mp);   /* :2163-2166 pllSaveCurrentTreeSprParsimony, before the tie rule */
  if (o->tie_mode == ORC_TIE_RANDOM) {                /* :2168-2176 */
    if (mp < o->best) o->hits = 1;
    else if (mp == o->best) o->hits++;
    if (mp < o->best || (mp == o->best && tie_draw(o) <= 1.0 / (double)o->hits)) {
      o->best = mp; o->insert_rec = q; o->remove_rec = p;
    }
  } else if (mp < o->best) {                          /* fastDNAparsimony.c:1224-1229 */
    o->best = mp; o->insert_rec = q; o->remove_rec = p;
  }
  hookup(o, q, r);
  o->back[NX(NX(p))] = o->back[NX(p)] = -1;
}

/* ---- addTraverseParsimony: sprparsimony.cpp:2208-2218 (doAll = FALSE) ---- */
static void add_traverse(orc *o, int p, int q, int mintrav, int maxtrav)
{
  if (--mintrav <= 0) test_insert(o, p, q);
  if (!TIP(o, q) && --maxtrav > 0) {
    add_traverse(o, p, o->back[NX(q)], mintrav, maxtrav);
    add_traverse(o, p, o->back[NX(NX(q))], mintrav, maxtrav);
  }
}

/* ---- removeNodeParsimony: sprparsimony.cpp:2245-2257 ---- */
static void remove_node(orc *o, int p)
{
  int q = o->back[NX(p)], r = o->back[NX(NX(p))];
  hookup(o, q, r);
  o->back[NX(NX(p))] = o->back[NX(p)] = -1;
}

/* ---- rearrangeParsimony: sprparsimony.cpp:2259-2376 (PLL: fastDNAparsimony.c:1316-1428) ---- */
int orc_rearrange(orc *o, int p, int mintrav, int maxtrav)
{
  int q, p1, p2, q1, q2, mintrav2;
  if (maxtrav > o->ntips - 3) maxtrav = o->ntips - 3;
  assert(mintrav == 1);
  if (maxtrav < mintrav) return 0;
  q = o->back[p];
  if (o->pre_eval < 0 ? o->tie_mode == ORC_TIE_RANDOM : o->pre_eval) {
    const unsigned mp0 = orc_evaluate(o, p, 0);       /* :2285, mpboot only */
    if (o->ufb_on) ufb_save_current_tree(o, -(double)mp0);   /* :2286-2289: the CURRENT tree is booked once per prune node */
  }
  trace_push(o, -1, 0);
  if (!TIP(o, p)) {
    p1 = o->back[NX(p)];
    p2 = o->back[NX(NX(p))];
    if (!TIP(o, p1) || !TIP(o, p2)) {
      remove_node(o, p);
      if (!TIP(o, p1)) {
        add_traverse(o, p, o->back[NX(p1)], mintrav, maxtrav);
        add_traverse(o, p, o->back[NX(NX(p1))], mintrav, maxtrav);
      }
      if (!TIP(o, p2)) {
        add_traverse(o, p, o->back[NX(p2)], mintrav, maxtrav);
        add_traverse(o, p, o->back[NX(NX(p2))], mintrav, maxtrav);
      }
      hookup(o, NX(p), p1);
      hookup(o, NX(NX(p)), p2);
      newview(o, p);
    }
  }
  trace_push(o, -2, 0);
  if (!TIP(o, q) && maxtrav > 0) {
    q1 = o->back[NX(q)];
    q2 = o->back[NX(NX(q))];
    if ((!TIP(o, q1) && (!TIP(o, o->back[NX(q1)]) || !TIP(o, o->back[NX(NX(q1))]))) ||
        (!TIP(o, q2) && (!TIP(o, o->back[NX(q2)]) || !TIP(o, o->back[NX(NX(q2))])))) {
      remove_node(o, q);
      mintrav2 = mintrav > 2 ? mintrav : 2;
      if (!TIP(o, q1)) {
        add_traverse(o, q, o->back[NX(q1)], mintrav2, maxtrav);
        add_traverse(o, q, o->back[NX(NX(q1))], mintrav2, maxtrav);
      }
      if (!TIP(o, q2)) {
        add_traverse(o, q, o->back[NX(q2)], mintrav2, maxtrav);
        add_traverse(o, q, o->back[NX(NX(q2))], mintrav2, maxtrav);
      }
      hookup(o, NX(q), q1);
      hookup(o, NX(NX(q)), q2);
      newview(o, q);
    }
  }
  return 1;
}

void orc_set_best(orc *o, unsigned best) { o->best = best; o->insert_rec = o->remove_rec = -1; o->hits = 1; }
unsigned orc_get_best(const orc *o, int *remove_rec, int *insert_rec)
{
  if (remove_rec) *remove_rec = o->remove_rec;
  if (insert_rec) *insert_rec = o->insert_rec;
  return o->best;
}

/* ---- restoreTreeParsimony / restoreTreeRearrangeParsimony: sprparsimony.cpp:2191-2205, :2379-2384 ---- */
static void restore_rearrange(orc *o)
{
  int p = o->remove_rec, q = o->insert_rec, r, counter = 4;
  remove_node(o, p);
  r = o->back[q];
  hookup(o, NX(p), q);
  hookup(o, NX(NX(p)), r);
  traversal(o, p, &counter, 0);
  o->ti[0] = counter;
  newview_iter(o);
}

/* ---- the sweep loop shared by pllOptimizeSprParsimony (:3295-3316) and
        _pllMakeParsimonyTreeFast (:3185-3206); PLL original: fastDNAparsimony.c:1919-1938 ---- */
static unsigned spr_sweeps(orc *o, int mintrav, int maxtrav, unsigned randomMP)
{
  unsigned startMP;
  unsigned iter_hits = 1;
  int i;
  long visits = 0;
  do {
    startMP = randomMP;
    orc_node_rectifier(o);
    for (i = 1; i <= 2 * o->n - 2; i++) {
      /* (test aid: the climb is cut short behind a given number of visits -- the state stays as that visit left it) */
      if (o->max_visits > 0 && visits++ >= o->max_visits) return randomMP;
      if (o->tie_mode == ORC_TIE_RANDOM) {
        o->insert_rec = o->remove_rec = -1;
        o->hits = 1;
        orc_rearrange(o, o->nodep[i], mintrav, maxtrav);
        if (o->best == randomMP) iter_hits++;
        if (o->best < randomMP) iter_hits = 1;
        if ((o->best < randomMP || (o->best == randomMP && tie_draw(o) <= 1.0 / (double)iter_hits)) &&
            o->remove_rec >= 0 && o->insert_rec >= 0) {
          moves_push(o, o->remove_rec, o->insert_rec, o->best);
          restore_rearrange(o);
          randomMP = o->best;
        }
      } else {
        orc_rearrange(o, o->nodep[i], mintrav, maxtrav);
        if (o->best < randomMP) {
          moves_push(o, o->remove_rec, o->insert_rec, o->best);
          restore_rearrange(o);
          randomMP = o->best;
        }
      }
    }
  } while (randomMP < startMP);
  return startMP;
}

/* ---- pllOptimizeSprParsimony: sprparsimony.cpp:3244-3319 ---- */
unsigned orc_optimize_spr(orc *o, int mintrav, int maxtrav)
{
  orc_node_rectifier(o);
  o->best = UINT_MAX;
  o->best = orc_evaluate(o, o->start, 1);
  o->ntips = o->n;
  o->insert_rec = o->remove_rec = -1;
  /* a ratchet climb starts with _pattern_pars as the IQ-TREE kernel left it: the per-pattern lengths of the tree the climb
     starts from (optimizeAllBranches -> computeParsimony on the perturbed alignment, iqtree.cpp:1712-1714; the full
     evaluate above has just filled the per-site counters of that tree) */
  if (o->ufb_on && o->ufb_ratchet) (void)orc_pattern_scores(o, o->ufb_ptn);
  return spr_sweeps(o, mintrav, maxtrav, o->best);
}

/* ---- makePermutationFast: sprparsimony.cpp:2221-2242 ---- */
static void make_permutation(orc *o, int *perm)
{
  int i, j, k, n = o->n;
  for (i = 1; i <= n; i++) perm[i] = i;
  for (i = 1; i <= n; i++) {
    double d = orc_randum(&o->randum_seed);
    k = (int)((double)(n + 1 - i) * d);
    j = perm[i]; perm[i] = perm[i + k]; perm[i + k] = j;
  }
}

/* ---- stepwiseAddition: sprparsimony.cpp:2977-3019 (PLL: fastDNAparsimony.c:1776-1814) ---- */
static void stepwise(orc *o, int p, int q)
{
  int r = o->back[q], counter = 4;
  unsigned mp;
  o->back[NX(p)] = q; o->back[q] = NX(p);
  o->back[NX(NX(p))] = r; o->back[r] = NX(NX(p));
  traversal(o, p, &counter, 0);
  o->ti[0] = counter;
  o->ti[1] = NUM(p);
  o->ti[2] = NUM(o->back[p]);
  mp = evaluate_iter(o);
  o->c_test++;
  trace_push(o, q, mp);
  if (o->tie_mode == ORC_TIE_RANDOM) {
    if (mp < o->best) o->hits = 1;
    else if (mp == o->best) o->hits++;
    if (mp < o->best || (mp == o->best && tie_draw(o) <= 1.0 / (double)o->hits)) { o->best = mp; o->insert_rec = q; }
  } else if (mp < o->best) { o->best = mp; o->insert_rec = q; }
  o->back[q] = r; o->back[r] = q;
  if (!TIP(o, q) && o->score[NUM(q)] > 0) {
    stepwise(o, p, o->back[NX(q)]);
    stepwise(o, p, o->back[NX(NX(q))]);
  }
}

/* ---- buildNewTip / buildSimpleTree + the addition loop of _pllMakeParsimonyTreeFast:
        sprparsimony.cpp:1955-1981, :3107-3181 ---- */
static void addition_phase(orc *o, long seed, int *perm, unsigned *best_per_step, int *insert_per_step)
{
  int n = o->n, ip, iq, ir, i, p, s, f, nextsp;
  reset_flags(o);                                    /* _allocateParsimonyDataStructures, :3228 */
  memset(o->score, 0, sizeof(unsigned) * 2 * o->n);
  o->randum_seed = seed;
  make_permutation(o, perm);
  o->ntips = 0;
  o->nextnode = n + 1;
  ip = perm[1]; iq = perm[2]; ir = perm[3];
  i = ip < iq ? ip : iq;
  if (ir < i) i = ir;
  o->start = o->nodep[i];
  o->ntips = 3;
  p = o->nodep[ip];
  hookup(o, p, o->nodep[iq]);
  s = o->nodep[o->nextnode++];                       /* buildNewTip */
  hookup(o, o->nodep[ir], s);
  o->back[NX(s)] = o->back[NX(NX(s))] = -1;
  insert_node(o, s, p);
  f = o->start;
  o->hits = 1;
  while (o->ntips < n) {
    int q, r, counter = 4;
    o->best = INT_MAX;
    nextsp = ++o->ntips;
    p = o->nodep[perm[nextsp]];
    q = o->nodep[o->nextnode++];
    o->back[p] = q; o->back[q] = p;
    trace_push(o, -1, 0);
    stepwise(o, q, o->back[f]);
    if (best_per_step) best_per_step[nextsp] = o->best;
    if (insert_per_step) insert_per_step[nextsp] = o->insert_rec;
    r = o->back[o->insert_rec];
    hookup(o, NX(q), o->insert_rec);
    hookup(o, NX(NX(q)), r);
    traversal(o, q, &counter, 0);
    o->ti[0] = counter;
    newview_iter(o);
  }
}

unsigned orc_stepwise(orc *o, long seed, unsigned *best_per_step, int *insert_per_step)
{
  int *perm = (int *)malloc(sizeof(int) * (o->n + 2));
  addition_phase(o, seed, perm, best_per_step, insert_per_step);
  free(perm);
  return o->best;
}

unsigned orc_make_tree(orc *o, long seed, int spr_dist, int *perm_out)
{
  int *perm = (int *)malloc(sizeof(int) * (o->n + 2));
  /* _pllMakeParsimonyTreeFast runs with perSiteScores = 0 (sprparsimony.cpp:3126, :3185-3206): no saveCurrentTree here,
     whatever tracker is attached */
  const int ufb_saved = o->ufb_on, ps_saved = o->persite_on;
  o->ufb_on = 0;
  o->persite_on = 0;
  addition_phase(o, seed, perm, NULL, NULL);
  if (perm_out) memcpy(perm_out, perm, sizeof(int) * (o->n + 1));
  free(perm);
  orc_node_rectifier(o);
  spr_sweeps(o, 1, spr_dist, o->best);
  o->ufb_on = ufb_saved;
  o->persite_on = ps_saved;
  return o->best;
}

void orc_counters(const orc *o, unsigned long long *nv, unsigned long long *ev, unsigned long long *ts)
{
  if (nv) *nv = o->c_newview;
  if (ev) *ev = o->c_eval;
  if (ts) *ts = o->c_test;
}

/* ---- UFBoot-MP attachment (IQTree::setParams' allocation, iqtree.cpp:213-262) ---- */
void orc_ufboot_detach(orc *o)
{
  int i;
  for (i = 0; i < o->ufb_nstore; i++) free(o->ufb_store_back[i]);
  free(o->ufb_store_back); free(o->ufb_store_idx);
  free(o->ufb_w0); o->ufb_w0 = NULL;
  for (i = 0; i < o->ufb_nkeys; i++) free(o->ufb_keys[i]);
  free(o->ufb_keys); free(o->ufb_key_idx); free(o->ufb_key_hash);
  o->ufb_keys = NULL; o->ufb_key_idx = NULL; o->ufb_key_hash = NULL; o->ufb_nkeys = o->ufb_keys_cap = 0;
  o->ufb_store_trees = 0; o->ufb_dups = 0;
  if (o->ufb_set) for (i = 0; i < o->ufb_B; i++) free(o->ufb_set[i]);
  free(o->ufb_set); free(o->ufb_set_n); free(o->ufb_set_cap);
  o->ufb_set = NULL; o->ufb_set_n = o->ufb_set_cap = NULL;
  o->ufb_mulhits = 0;
  o->ufb_cut_btrees = 0;
  free(o->ufb_top_idx); free(o->ufb_top_rell); free(o->ufb_top_n); free(o->ufb_thr); free(o->ufb_top_iter);
  o->ufb_top_idx = o->ufb_top_rell = o->ufb_top_n = o->ufb_thr = o->ufb_top_iter = NULL;
  o->ufb_topboot = o->ufb_distinct = o->ufb_cur_it = 0;
  free(o->ufb_samples); free(o->ufb_logl); free(o->ufb_counts); free(o->ufb_orig); free(o->ufb_trees); free(o->ufb_treels); free(o->ufb_ptn);
  o->ufb_store_back = NULL; o->ufb_store_idx = NULL; o->ufb_nstore = o->ufb_store_cap = 0;
  o->ufb_samples = NULL; o->ufb_logl = NULL; o->ufb_counts = NULL; o->ufb_orig = NULL; o->ufb_trees = NULL; o->ufb_treels = NULL; o->ufb_ptn = NULL;
  o->ufb_ntrees = o->ufb_treels_cap = 0;
  o->ufb_on = 0;
}
void orc_ufboot_attach(orc *o, int B, const unsigned short *samples, double epsilon)
{
  int b;
  orc_ufboot_detach(o);
  o->ufb_B = B;
  o->ufb_samples = (unsigned short *)malloc(sizeof(unsigned short) * (size_t)B * o->P);
  memcpy(o->ufb_samples, samples, sizeof(unsigned short) * (size_t)B * o->P);
  o->ufb_w0 = (int *)malloc(sizeof(int) * o->P);
  memcpy(o->ufb_w0, o->wgt, sizeof(int) * o->P);
  o->ufb_eps = epsilon;
  o->ufb_cutoff = 0.0;                                   /* iqtree.cpp:68 */
  o->ufb_logl = (double *)malloc(sizeof(double) * B);
  o->ufb_counts = (int *)calloc(B, sizeof(int));
  o->ufb_orig = (int *)calloc(B, sizeof(int));                 /* iqtree.cpp:254: resize(gbo_replicates, 0) */
  o->ufb_trees = (int *)malloc(sizeof(int) * B);
  for (b = 0; b < B; b++) { o->ufb_logl[b] = -(double)LONG_MAX; o->ufb_trees[b] = -1; }   /* :248-253 */
  o->ufb_ptn = (unsigned short *)calloc((size_t)o->P + 16, sizeof(unsigned short));
  o->ufb_bad = 0;
  o->ufb_draws = 0;
  o->ufb_on = 1;
  o->ufb_ratchet = 0;
  o->ufb_ratchet_booking = 1;
  o->ufb_set = (int **)calloc(B, sizeof(int *));
  o->ufb_set_n = (int *)calloc(B, sizeof(int));
  o->ufb_set_cap = (int *)calloc(B, sizeof(int));
  orc_enable_persite(o, 1);                              /* perSiteScores = gbo_replicates > 0, sprparsimony.cpp:3245 */
}
void orc_ufboot_set_cutoff(orc *o, double logl_cutoff) { o->ufb_cutoff = logl_cutoff; }
void orc_ufboot_set_ratchet_booking(orc *o, int on) { o->ufb_ratchet_booking = on != 0; }   /* !no_hclimb1_bb; next set_weights */
void orc_ufboot_set_store_trees(orc *o, int on) { o->ufb_store_trees = on != 0; }          /* params->store_candidate_trees */
int orc_ufboot_duplicates(const orc *o) { return o->ufb_dups; }
void orc_ufboot_set_mulhits(orc *o, int on) { o->ufb_mulhits = on != 0; }                  /* params->multiple_hits */
void orc_ufboot_set_topboot(orc *o, int n_top)                                              /* params->store_top_boot_trees; with -mulhits */
{
  int b;
  free(o->ufb_top_idx); free(o->ufb_top_rell); free(o->ufb_top_n); free(o->ufb_thr);
  o->ufb_topboot = n_top > 0 ? n_top : 0;
  o->ufb_top_idx = (int *)calloc((size_t)o->ufb_B * (size_t)(n_top > 0 ? n_top : 1), sizeof(int));
  o->ufb_top_rell = (int *)calloc((size_t)o->ufb_B * (size_t)(n_top > 0 ? n_top : 1), sizeof(int));
  o->ufb_top_n = (int *)calloc((size_t)o->ufb_B, sizeof(int));
  o->ufb_thr = (int *)malloc(sizeof(int) * (size_t)o->ufb_B);
  for (b = 0; b < o->ufb_B; b++) o->ufb_thr[b] = -INT_MAX;                                   /* iqtree.cpp:267 */
}
void orc_ufboot_set_distinct_iter(orc *o, int k)                                            /* params->distinct_iter_top_boot; without -mulhits */
{
  orc_ufboot_set_topboot(o, k);                                                             /* the same arrays (iqtree.cpp:270-277) */
  o->ufb_topboot = 0;
  o->ufb_distinct = k > 0 ? k : 0;
  free(o->ufb_top_iter);
  o->ufb_top_iter = (int *)calloc((size_t)o->ufb_B * (size_t)(k > 0 ? k : 1), sizeof(int));
}
void orc_ufboot_set_iteration(orc *o, int cur_it) { o->ufb_cur_it = cur_it; }               /* IQTree::curIt */
int orc_ufboot_sample_iters(const orc *o, int sample, int *iters)
{
  int i, n = o->ufb_top_n ? o->ufb_top_n[sample] : 0;
  for (i = 0; i < n && o->ufb_top_iter; i++) iters[i] = o->ufb_top_iter[(size_t)sample * o->ufb_distinct + i];
  return n;
}
int orc_ufboot_sample_top(const orc *o, int sample, int *idx, int *rell, int *threshold)    /* boot_trees_parsimony_top[sample] */
{
  const int stride = o->ufb_topboot ? o->ufb_topboot : o->ufb_distinct;
  int i, n = o->ufb_top_n ? o->ufb_top_n[sample] : 0;
  for (i = 0; i < n; i++) { idx[i] = o->ufb_top_idx[(size_t)sample * stride + i]; rell[i] = o->ufb_top_rell[(size_t)sample * stride + i]; }
  if (threshold) *threshold = o->ufb_thr ? o->ufb_thr[sample] : 0;
  return n;
}
int orc_ufboot_sample_trees(const orc *o, int sample, int *out, int cap)                    /* boot_trees_parsimony[sample] */
{
  int i, n = o->ufb_set_n[sample];
  for (i = 0; i < n && i < cap; i++) out[i] = o->ufb_set[sample][i];
  return n;
}
int orc_ufboot_ntrees(const orc *o) { return o->ufb_ntrees; }
int orc_ufboot_bad(const orc *o) { return o->ufb_bad; }
unsigned long long orc_ufboot_draws(const orc *o) { return o->ufb_draws; }
void orc_ufboot_tree_logl(const orc *o, double *out) { memcpy(out, o->ufb_treels, sizeof(double) * o->ufb_ntrees); }
void orc_ufboot_state(const orc *o, double *boot_logl, int *boot_counts, int *boot_trees)
{
  memcpy(boot_logl, o->ufb_logl, sizeof(double) * o->ufb_B);
  memcpy(boot_counts, o->ufb_counts, sizeof(int) * o->ufb_B);
  memcpy(boot_trees, o->ufb_trees, sizeof(int) * o->ufb_B);
}
int orc_ufboot_tree(const orc *o, int tree_index, int *back)
{
  int i;
  for (i = o->ufb_nstore - 1; i >= 0; i--)
    if (o->ufb_store_idx[i] == tree_index) { memcpy(back, o->ufb_store_back[i], sizeof(int) * 3 * (2 * o->n - 1)); return 1; }
  return 0;
}
/* Books of another search chain of the same run (iteration-parallel -bb; the engine's mpf_ufboot_adopt): sample[k] is offered tree
   tree_of[k] at REPS length score[k] and takes it when strictly better -- the strict branch of the default update rule
   (iqtree.cpp:3686, :3710-3720: rell > boot_logl + epsilon -> boot_logl, boot_counts = 1 and counted once more at :3728-3730,
   boot_trees through the tree-string map :3689-3707); the tree enters treels_logl under its length on the original alignment. */
int orc_ufboot_adopt(orc *o, int n_upd, const int *sample, const unsigned *score, const int *tree_of, int n_trees, const int *backs,
                     const unsigned *lengths)
{
  const int nrec = 3 * (2 * o->n - 1);
  int *save = (int *)malloc(sizeof(int) * nrec), *idx = (int *)malloc(sizeof(int) * (n_trees > 0 ? n_trees : 1));
  int k, taken = 0;
  memcpy(save, o->back, sizeof(int) * nrec);
  for (k = 0; k < n_trees; k++) idx[k] = -1;
  for (k = 0; k < n_upd; k++) {
    const int b = sample[k], t = tree_of[k];
    const double rell = -(double)score[k];
    if (!(rell > o->ufb_logl[b] + o->ufb_eps)) continue;
    if (idx[t] < 0) {
      int ti;
      memcpy(o->back, backs + (size_t)t * nrec, sizeof(int) * nrec);
      ti = ufb_lookup_topology(o, o->ufb_ntrees);
      if (ti == o->ufb_ntrees) {
        if (o->ufb_ntrees == o->ufb_treels_cap) {
          o->ufb_treels_cap = o->ufb_treels_cap ? 2 * o->ufb_treels_cap : 1024;
          o->ufb_treels = (double *)realloc(o->ufb_treels, sizeof(double) * o->ufb_treels_cap);
        }
        o->ufb_treels[o->ufb_ntrees++] = -(double)lengths[t];
      }
      ufb_store_tree(o, ti);
      idx[t] = ti;
    }
    o->ufb_logl[b] = rell;
    o->ufb_counts[b] = 2;
    o->ufb_trees[b] = idx[t];
    if (o->ufb_cut_btrees) o->ufb_orig[b] = -(int)lengths[t];
    taken++;
  }
  memcpy(o->back, save, sizeof(int) * nrec);
  free(save); free(idx);
  return taken;
}
/* the per-iteration cut-off update, "top cutoff_percent %" rule (iqtree.cpp:1662-1676, cutoff_percent = 10 tools.cpp:793) */
static int cmp_desc(const void *a, const void *b) { double x = *(const double *)a, y = *(const double *)b; return x < y ? 1 : x > y ? -1 : 0; }
void orc_ufboot_set_cutoff_from_btrees(orc *o, int on) { o->ufb_cut_btrees = on != 0; }
void orc_ufboot_orig_logl(const orc *o, int *out) { memcpy(out, o->ufb_orig, sizeof(int) * (size_t)o->ufb_B); }

double orc_ufboot_next_cutoff(const orc *o, int percent)
{
  double *l, c;
  if (o->ufb_cut_btrees) {                               /* :1657-1660: logl_cutoff = min(boot_tree_orig_logl) */
    int b, mn = o->ufb_orig[0];
    for (b = 1; b < o->ufb_B; b++) if (o->ufb_orig[b] < mn) mn = o->ufb_orig[b];
    return (double)mn;
  }
  if (o->ufb_ntrees <= 1000) return o->ufb_cutoff;
  l = (double *)malloc(sizeof(double) * o->ufb_ntrees);
  memcpy(l, o->ufb_treels, sizeof(double) * o->ufb_ntrees);
  qsort(l, o->ufb_ntrees, sizeof(double), cmp_desc);     /* nth_element(..., greater) then [k]: the k-th largest */
  c = l[(size_t)o->ufb_ntrees * percent / 100];
  free(l);
  return c;
}
