// ufboot_common.hpp -- what the translation units of the online UFBoot bookkeeping share (host/ufboot*.cpp).
#pragma once
#include <algorithm>
#include <climits>
#include <cmath>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <pthread.h>
#include <sched.h>
#include "../csrc/engine.hpp"
#include "lcg_block.hpp"
#include "simd_util.hpp"

namespace mpf {

#define UCHK(expr)                                                                                   \
  do {                                                                                               \
    hipError_t e__ = (expr);                                                                         \
    if (e__ != hipSuccess) { set_error(std::string(#expr) + ": " + hipGetErrorString(e__)); return MPF_E_HIP; } \
  } while (0)

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
static inline bool ufb_trace_env() { static const bool on = std::getenv("MPF_UFB_TRACE") != nullptr; return on; }
static inline double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// events into replay order: by scan output index, then by sample.  The current tree, booked once per prune-node visit, ties
// with every sample it is the best tree of -- millions of events per sweep -- so large batches take two stable counting
// passes (sample, then index) instead of a comparison sort.
static inline void sort_events(std::vector<UfbEvent> &ev, std::vector<UfbEvent> &tmp, std::vector<uint32_t> &count, uint32_t n_idx, uint32_t n_samples)
{
  const size_t n = ev.size();
  if (n < 512) {
    std::sort(ev.begin(), ev.end(), [](const UfbEvent &x, const UfbEvent &y) { return x.idx != y.idx ? x.idx < y.idx : x.b < y.b; });
    return;
  }
  tmp.resize(n);
  auto pass = [&](const std::vector<UfbEvent> &src, std::vector<UfbEvent> &dst, uint32_t nkeys, bool by_idx) {
    count.assign((size_t)nkeys + 1, 0u);
    for (const UfbEvent &e : src) count[(size_t)(by_idx ? e.idx : e.b) + 1]++;
    for (size_t k = 1; k <= nkeys; k++) count[k] += count[k - 1];
    for (const UfbEvent &e : src) dst[count[by_idx ? e.idx : e.b]++] = e;
  };
  uint32_t max_idx = 0;
  for (const UfbEvent &e : ev) max_idx = std::max(max_idx, e.idx);     // (a sharded run's merged events: other ranks' indices too)
  pass(ev, tmp, n_samples, false);
  pass(tmp, ev, std::max(n_idx, max_idx + 1u), true);
}

// the -distinct_iter_top_boot block of saveCurrentTree for one (tree, sample); tree_index / looked_up: the call's tree string
// state (resolved through `lookup` at the first acceptance); true if some list or boot_trees entry now names tree_index
template <class Lookup>
bool Engine::ufb_distinct_offer(uint32_t b, int32_t rell, int64_t &tree_index, bool &looked_up, Lookup lookup)
{
  UfbState &u = *ufb_;
  auto &top = u.top[b];
  auto &its = u.top_iter[b];
  int32_t &thr = u.top_thr[b];
  const int k = u.distinct;
  if (rell >= thr) u.boot_counts[b]++;                                      // :3589-3591
  bool take = rell > thr;
  if (!take && rell == thr) {                                               // :3593-3595, the draw only on a tie
    u.draws++;
    take = tie_draw() <= (double)k * 1.0 / (double)u.boot_counts[b];
  }
  if (!take) return false;
  uint32_t &bs = u.boot_score[b];
  const uint32_t len = (uint32_t)(-(int64_t)rell);
  if (len < bs) u.boot_counts[b] = 1;                                       // :3598-3600
  if (u.cut_btrees) u.boot_orig[b] = u.cur_logl_now;                                          // :3617-3619
  if (!looked_up) { tree_index = lookup(tree_index); looked_up = true; }
  bool named = false;
  auto ref = [&](int64_t t) { u.refs[(size_t)t]++; named = true; };
  auto unref = [&](int64_t t) { if (--u.refs[(size_t)t] == 0) u.store.erase(t); };
  int64_t &bt = u.boot_trees[b];
  if (bt != tree_index) {                                                   // :3620
    ref(tree_index);
    if (bt >= 0) unref(bt);
    bt = tree_index;
  }
  if (len < bs) bs = len;                                                   // :3621 max()
  const int t = std::min(k, (int)its.size());
  for (int c = 0; c < t; c++) if (top[(size_t)c].first == tree_index) return named;      // :3627-3634 tree exists
  int c = 0;
  for (; c < t; c++)
    if (its[(size_t)c] == u.cur_it) {                                       // :3637-3645 this iteration's representative
      if (rell > top[(size_t)c].second) {
        ref(tree_index);
        unref(top[(size_t)c].first);
        top[(size_t)c] = std::make_pair(tree_index, rell);
      }
      break;
    }
  if (c == t && t < k) {                                                    // :3648-3651
    its.push_back(u.cur_it);
    top.push_back(std::make_pair(tree_index, rell));
    ref(tree_index);
  } else if (c == t && t == k) {                                            // :3654-3667 replace the worst
    int worst = 0;
    for (int d = 1; d < t; d++) if (top[(size_t)d].second < top[(size_t)worst].second) worst = d;
    ref(tree_index);
    unref(top[(size_t)worst].first);
    top[(size_t)worst] = std::make_pair(tree_index, rell);
    its[(size_t)worst] = u.cur_it;
  }
  thr = top[0].second;                                                      // :3670-3675
  for (size_t d = 1; d < top.size(); d++) thr = std::min(thr, top[d].second);
  return named;
}

}  // namespace mpf
