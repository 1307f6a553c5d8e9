// ufboot.hpp -- launch interface of the online UFBoot-MP kernels (ufboot.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.hpp"

namespace mpf {

struct UfbEvent { uint32_t idx, b, s; };     // scan output index, sample, parsimony length under that sample

constexpr int kUfbRowTile = 512, kUfbColTile = 128;   // padding units of k_bitgemm: rows of the mask matrix, samples

// masks[op][Wp] = sites mutating on join (a, b)
hipError_t launch_join_masks(hipStream_t st, const Geometry &g, const uint32_t *vec, const EvOp *ops, int n_ops, uint32_t *masks);
// C[rows_padded][Bp] = (accumulate ? C : 0) + mult * masks x Wt; rows_padded % kUfbRowTile == 0, Bp % kUfbColTile == 0
// rowsel (optional, rows_padded entries): output row i multiplies mask row rowsel[i]
// row_limit (optional, device word): row blocks that start at or behind *row_limit are skipped (their part of C is left as it is)
hipError_t launch_bitgemm(hipStream_t st, const uint32_t *masks, int rows_padded, int Wp, const uint8_t *Wt, int Bp, int32_t *C,
                          int mult, int accumulate, const uint32_t *rowsel = nullptr, const uint32_t *row_limit = nullptr);
hipError_t launch_colsum(hipStream_t st, const int32_t *C, int rows, int Bp, int32_t *rt);
// rt += C[row] - C[home]
hipError_t launch_rt_update(hipStream_t st, int32_t *rt, const int32_t *C, int Bp, uint32_t row, uint32_t home);
uint32_t ufb_chunks(uint32_t n_idx);
// cmin, pre: scratch of ufb_chunks(n_idx) * Bp words each
// crow (optional): scan output index -> row of C (compacted product)
hipError_t launch_ufb_events(hipStream_t st, const uint2 *info, const uint32_t *cost, const uint32_t *thr, const uint32_t *home,
                             const uint32_t *crow, const int32_t *C, int Bp, int B, const int32_t *rt, const uint32_t *best, uint32_t n_idx,
                             uint32_t *cmin, uint32_t *pre, UfbEvent *ev, uint32_t ev_cap, uint32_t *ev_count,
                             // 1: best[b] is a FIXED bound (an event = score <= best[b]; the top-N rules), not the start of a running minimum
                             int fixed_bound = 0);

// ---- the batches inside a climb: the extraction kernel's last workgroup writes the results into the host's pinned
// buffers itself and raises a flag the host polls -- no copy dispatches, no stream synchronisation
constexpr uint32_t kUfbEvents2Max = 4096;    // scan output indices up to which the one-launch extraction (k_ufb_events2) is used
struct UfbPublishArgs {
  const uint32_t *src[3] = {nullptr, nullptr, nullptr};   // device word ranges copied to ...
  uint32_t *dst[3] = {nullptr, nullptr, nullptr};         // ... pinned host memory
  uint32_t words[3] = {0, 0, 0};
  UfbEvent *h_ev = nullptr;                  // pinned: the first h_ev_cap events
  uint32_t h_ev_cap = 0;
  uint32_t *h_flag = nullptr;                // pinned: [0] = event count, [1] = 1 once everything has arrived (zeroed by the host before)
  uint32_t *done = nullptr;                  // device: zeroed word (left zeroed)
};
// C[0, c_words) <- 0, *ev_count <- 0, info[self_idx[i]] = (0, code) for i < n_self
hipError_t launch_ufb_prep(hipStream_t st, int32_t *C, size_t c_words, uint2 *info, const uint32_t *self_idx, uint32_t n_self, uint32_t code,
                           uint32_t *ev_count);
// launch_ufb_events + the publication of its results (one launch for n_idx <= kUfbEvents2Max, else the chunked kernels + a
// publishing launch); ev_count must be zero
hipError_t launch_ufb_events_publish(hipStream_t st, const uint2 *info, const uint32_t *cost, const uint32_t *thr, const uint32_t *home,
                                     const uint32_t *crow, const int32_t *C, int Bp, int B, const int32_t *rt, const uint32_t *best,
                                     uint32_t n_idx, uint32_t *cmin, uint32_t *pre, UfbEvent *ev, uint32_t ev_cap, uint32_t *ev_count,
                                     int fixed_bound, const UfbPublishArgs &a,
                                     // 1: start every sample's running minimum at min(best[b], rt[b]) (see k_ufb_events2); only with n_idx <= kUfbEvents2Max
                                     int clamp_rt = 0,
                                     // device word: indices from *cut on are not looked at (only with n_idx <= kUfbEvents2Max)
                                     const uint32_t *cut = nullptr);
// launch_ufb_prep + publication of a.src/dst ranges behind a.h_flag[1] (the scan's results, in front of the product)
// ... and *cut = the batch's certain end (see k_ufb_mid; also written to a.h_flag[2]): cost = the scan's output, home / plan_end per part
hipError_t launch_ufb_mid(hipStream_t st, int32_t *C, size_t c_words, uint2 *info, const uint32_t *self_idx, uint32_t n_self, uint32_t code,
                          uint32_t *ev_count, const UfbPublishArgs &a, const uint32_t *cost, const uint32_t *home, const uint32_t *plan_end,
                          uint32_t n_idx, uint32_t *cut);
// the row padding launch_bitgemm needs for this many rows (small products run on 128-row tiles)
int ufb_row_padding(int rows, int Bp);

// Wt (zeroed here) <- the samples' weights at the first expanded site of every pattern of the packing in force
hipError_t launch_ufb_layout(hipStream_t st, const uint16_t *src /* [n_cols][P] */, int n_cols, int P, const int32_t *first_site,
                             const int32_t *cur_weight, uint8_t *Wt, int Bp, int planes, size_t plane_bytes);
// info[idx[i]] = (0, code), i < n: the current tree's slots (code 0xFFFFFFFE: scored with R_T; 0xFFFFFFFF: skipped)
hipError_t launch_ufb_self(hipStream_t st, uint2 *info, const uint32_t *idx, uint32_t n, uint32_t code);
// weighted engine: planes[k][row][Wp] <- bit k of vals[row][npat], k < K (rows_p rows per plane; Wp >= ceil(npat / 64) * 2)
hipError_t launch_vals_planes(hipStream_t st, const uint16_t *vals, uint32_t rows, uint32_t npat, int K, uint32_t *planes, uint32_t rows_p,
                              uint32_t Wp);
// out[i] = C[i][col], i < rows
hipError_t launch_ufb_column(hipStream_t st, const int32_t *C, int Bp, int col, uint32_t rows, int32_t *out);

}  // namespace mpf
