// bounds.cpp -- the small host-side lower-bound helpers of the reference's parsimony path (no device work).
//
// The reference uses them to cut its CPU loops short: the REPS loop skips a bootstrap sample once
// "partial sum + bound of the remaining segments" is already worse than the sample's best (iqtree.cpp:3435-3445), and
// the Sankoff evaluate returns early on the same kind of estimate (sprparsimony.cpp:946-955).  The engine computes
// exact full sums instead (neither skip changes a result), so these are provided for hosts that still want the
// numbers: mpboot's IQTree::doSegmenting / pllComputeRellRemainBound bookkeeping, and its reports.
#include <climits>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mpfitch.h"

namespace mpf {
void set_error(const std::string &msg);
}
using mpf::set_error;

extern "C" {

// pllCalcMinParsScorePattern (reference sprparsimony.cpp:2513-2547) for every pattern:
// (number of distinct unambiguous tip codes present) - 1.  DNA: unambiguous = {1, 2, 4, 8}.  Protein: the reference's
// isUnambiguous (:2500-2507) tests PLL_DNA_DATA twice, so for protein EVERY code below `undetermined` (22), including
// B (20) and Z (21), counts -- restated as it behaves.
int mpf_min_pars_score_patterns(int32_t datatype, int32_t n_taxa, int32_t n_patterns, const uint8_t *codes, int32_t *min_score)
{
  if (!codes || !min_score || n_taxa < 1 || n_patterns < 1) { set_error("mpf_min_pars_score_patterns: bad argument"); return MPF_E_INVALID; }
  if (datatype != MPF_DNA && datatype != MPF_AA && datatype != MPF_BIN && datatype != MPF_GENERIC) { set_error("mpf_min_pars_score_patterns: unsupported data type"); return MPF_E_UNSUPPORTED; }
  const int undetermined = datatype == MPF_DNA ? 15 : datatype == MPF_AA ? 22 : datatype == MPF_BIN ? 3 : 32;
  for (int p = 0; p < n_patterns; p++) {
    bool seen[256] = {false};
    for (int t = 0; t < n_taxa; t++) seen[codes[(size_t)t * (size_t)n_patterns + (size_t)p]] = true;
    int cnt = 0;
    for (int j = 0; j < undetermined; j++) {
      if (!seen[j]) continue;
      const bool unamb = datatype == MPF_DNA ? (j == 1 || j == 2 || j == 4 || j == 8) : true;
      if (unamb) cnt++;
    }
    min_score[p] = cnt - 1;
  }
  return MPF_OK;
}

// ParsTree::findMstScore (reference parstree.cpp:606-680) for every pattern: weight of the minimum spanning tree
// (Prim, started at the lowest state present) over the unambiguous states present, edge weights cost[a * S + b].
// states: IQ-TREE codes ([n_taxa][n_patterns], values >= n_states are ambiguous/unknown and ignored, :615-618).
int mpf_mst_scores(int32_t n_states, const uint32_t *cost, int32_t n_taxa, int32_t n_patterns, const int8_t *states, uint32_t *mst)
{
  if (!cost || !states || !mst || n_states < 2 || n_states > 64 || n_taxa < 1 || n_patterns < 1) { set_error("mpf_mst_scores: bad argument"); return MPF_E_INVALID; }
  const int S = n_states;
  std::vector<uint32_t> label((size_t)S);
  std::vector<char> present((size_t)S), added((size_t)S);
  for (int p = 0; p < n_patterns; p++) {
    int count_present = 0;
    for (int s = 0; s < S; s++) { present[(size_t)s] = 0; added[(size_t)s] = 0; label[(size_t)s] = UINT_MAX; }
    for (int t = 0; t < n_taxa; t++) {
      const int st = states[(size_t)t * (size_t)n_patterns + (size_t)p];
      if (st >= 0 && st < S && !present[(size_t)st]) { present[(size_t)st] = 1; count_present++; }
    }
    if (count_present <= 1) { mst[p] = 0; continue; }
    int count = 0;
    for (int c = 0; c < S; c++) if (present[(size_t)c]) { label[(size_t)c] = 0; break; }     // :632-640, first pass only
    while (count < S) {
      int add = -1;
      uint32_t best = UINT_MAX;
      for (int c = 0; c < S; c++)
        if (!added[(size_t)c] && present[(size_t)c] && label[(size_t)c] < best) { best = label[(size_t)c]; add = c; }
      if (add < 0) break;
      added[(size_t)add] = 1;
      count++;
      for (int c = 0; c < S; c++)
        if (present[(size_t)c] && !added[(size_t)c] && label[(size_t)c] > cost[(size_t)add * S + c]) label[(size_t)c] = cost[(size_t)add * S + c];
    }
    uint32_t score = 0;
    for (int c = 0; c < S; c++) if (present[(size_t)c]) score += label[(size_t)c];
    mst[p] = score;
  }
  return MPF_OK;
}

// IQTree::doSegmenting (reference iqtree.cpp:3793-3820): cut the (sorted) patterns into segments whose
// sum(ras_pars_score * frequency) stays below USHRT_MAX / 16, closing a segment only on a multiple of the SIMD
// width `vcsize` (VCSIZE_USHORT: 16 with AVX, 8 with SSE); segment_upper[s] = first pattern of segment s + 1.
int mpf_segment_patterns(int32_t n_patterns, int32_t n_informative, int32_t vcsize, const int32_t *ras_pars_score,
                         const int32_t *frequency, int32_t *segment_upper, int32_t *n_segments)
{
  if (!ras_pars_score || !frequency || !segment_upper || !n_segments || n_patterns < 1 || vcsize < 1) { set_error("mpf_segment_patterns: bad argument"); return MPF_E_INVALID; }
  int seg = 0, sum = 0;
  for (int i = 0; i < n_patterns; i++) {
    sum += ras_pars_score[i] * frequency[i];
    if ((i + 1) % vcsize == 0 && sum > USHRT_MAX / 16) { segment_upper[seg++] = i + 1; sum = 0; }
  }
  if (sum) segment_upper[seg++] = n_informative;
  *n_segments = seg;
  return MPF_OK;
}

// The remain bounds of one weight vector (IQTree::pllComputeRellRemainBound, reference iqtree.cpp:3842-3853, and
// pllRemainderLowerBounds, sprparsimony.cpp:2813-2819): remain[s] = sum_{pos >= segment_upper[s]} min_unit_pars[pos] * weight[pos]
// for s = 0 .. n_segments - 2.
int mpf_remain_bounds(int32_t n_units, int32_t n_segments, const int32_t *segment_upper, const int32_t *min_unit_pars,
                      const uint16_t *weight, int32_t *remain)
{
  if (!segment_upper || !min_unit_pars || !weight || (!remain && n_segments > 1) || n_units < 1 || n_segments < 1) { set_error("mpf_remain_bounds: bad argument"); return MPF_E_INVALID; }
  for (int s = 0; s < n_segments - 1; s++) {
    int r = 0;
    for (int pos = segment_upper[s]; pos < n_units; pos++) r += min_unit_pars[pos] * (int)weight[pos];
    remain[s] = r;
  }
  return MPF_OK;
}


// ParsTree::loadCostMatrixFile (reference parstree.cpp:31-95): "fitch" / "e" = unit costs (:42-49), otherwise a text file
// "<nstates> then nstates x nstates unsigned entries" (:50-68); then the in-place closure under the triangle inequality,
// literally the reference's k-i-j loop over unsigned entries (:74-80) -- the repaired matrix is what every later
// computation uses (IQTree::initializePLL copies it into pllCostMatrix, iqtree.cpp:601-615).
int mpf_cost_matrix_triangle_fix(int32_t n_states, uint32_t *cost, int32_t *changed)
{
  if (!cost || n_states < 1) { set_error("mpf_cost_matrix_triangle_fix: bad argument"); return MPF_E_INVALID; }
  bool ch = false;
  const int S = n_states;
  for (int k = 0; k < S; k++)
    for (int i = 0; i < S; i++)
      for (int j = 0; j < S; j++)
        if (cost[i * S + j] > cost[i * S + k] + cost[k * S + j]) {
          ch = true;
          cost[i * S + j] = cost[i * S + k] + cost[k * S + j];
        }
  if (changed) *changed = ch ? 1 : 0;
  return MPF_OK;
}

int mpf_cost_matrix_load(const char *file_or_keyword, int32_t n_states_alignment, int32_t cap_states, uint32_t *cost,
                         int32_t *n_states, int32_t *changed)
{
  if (!file_or_keyword || !cost || !n_states || cap_states < 1) { set_error("mpf_cost_matrix_load: bad argument"); return MPF_E_INVALID; }
  int S = 0;
  if (!std::strcmp(file_or_keyword, "fitch") || !std::strcmp(file_or_keyword, "e")) {
    S = n_states_alignment;
    if (S < 1 || S > cap_states) { set_error("mpf_cost_matrix_load: state count does not fit the caller's buffer"); return MPF_E_INVALID; }
    for (int i = 0; i < S; i++)
      for (int j = 0; j < S; j++) cost[i * S + j] = i == j ? 0u : 1u;
  } else {
    std::FILE *f = std::fopen(file_or_keyword, "r");
    if (!f) { set_error(std::string("cannot read cost matrix file ") + file_or_keyword); return MPF_E_INVALID; }   // outError, :54-56
    if (std::fscanf(f, "%d", &S) != 1 || S < 1 || S > cap_states) { std::fclose(f); set_error("cost matrix file: bad state count"); return MPF_E_INVALID; }
    for (int i = 0; i < S * S; i++)
      if (std::fscanf(f, "%u", &cost[i]) != 1) { std::fclose(f); set_error("cost matrix file: fewer than nstates x nstates entries"); return MPF_E_INVALID; }
    std::fclose(f);
  }
  *n_states = S;
  return mpf_cost_matrix_triangle_fix(S, cost, changed);
}
}  // extern "C"
