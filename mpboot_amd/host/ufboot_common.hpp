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

// The update rule of one booked tree for one sample (b, score s) -- iqtree.cpp:3684-3731 and its -mulhits / -topboot /
// -distinct_iter_top_boot variants (:3498-3680): shared by the events the device extracted and by the current tree's own
// bookings the host walks through itself.  lookup(tree_index, cand_code) = treels.find(tree_str) / treels[tree_str] = tree_index
// (:3500-3514, :3689-3707): a topology that some sample accepted before keeps the index of that first tree.
template <class Lookup>
void Engine::ufb_one_event(const uint32_t b, const uint32_t s, int64_t &tree_index, bool &looked_up, const uint32_t cand_code, Lookup lookup, const UfbDeferCtx &dc)
{
  UfbState &u = *ufb_;
  uint32_t &bs = u.boot_score[b];
  if (u.distinct && !u.mulhits) {
    if (ufb_distinct_offer(b, -(int32_t)s, tree_index, looked_up, [&](int64_t ti) { return lookup(ti, cand_code); }) &&
        (u.pending.empty() || u.pending.back().tree_index != tree_index)) u.pending.push_back(UfbState::Pending{tree_index, cand_code});
    return;
  }
  if (u.mulhits && u.topboot) {
    // iqtree.cpp:3542-3585: the sample's list is not full yet, or the tree beats its threshold
    const int32_t rell = -(int32_t)s;
    if ((int)u.top[b].size() < u.topboot || rell > u.top_thr[b]) {
      const int64_t newest = (int64_t)u.treels.size() - 1;
      if (!looked_up) { tree_index = lookup(tree_index, cand_code); looked_up = true; }
      if (ufb_topboot_offer(b, rell, tree_index, tree_index == newest) &&
          (u.pending.empty() || u.pending.back().tree_index != tree_index)) u.pending.push_back(UfbState::Pending{tree_index, cand_code});
    }
    return;
  }
  if (u.mulhits) {
    // iqtree.cpp:3498-3540: rell >= boot_logl (an event is exactly that); no draw, boot_counts untouched
    if (s > bs) return;
    if (!looked_up) { tree_index = lookup(tree_index, cand_code); looked_up = true; }       // :3500-3514
    std::set<int64_t> &hs = u.hit_sets[b];
    if (s < bs) {                                               // :3516-3519
      for (int64_t t : hs) if (--u.refs[(size_t)t] == 0) u.store.erase(t);
      hs.clear();
      bs = s;
    }
    if (u.cut_btrees && u.cur_logl_now > u.boot_orig[b]) u.boot_orig[b] = u.cur_logl_now;     // :3523-3527
    if (hs.insert(tree_index).second) {                         // :3530-3533
      u.refs[(size_t)tree_index]++;
      if (u.pending.empty() || u.pending.back().tree_index != tree_index) u.pending.push_back(UfbState::Pending{tree_index, cand_code});
    }
    return;
  }
  bool accept = false;
  if (s < bs) accept = true;                                    // rell > boot_logl + epsilon (:3686)
  else if (s == bs) {                                           // rell > boot_logl - epsilon: tie, draw (:3687-3688)
    u.draws++;
    accept = tie_draw() <= 1.0 / (double)(u.boot_counts[b] + 1);
  }
  if (accept && u.cut_btrees) u.boot_orig[b] = u.cur_logl_now;                  // :3716-3718
  if (accept && dc.on) {
    u.log.push_back(UfbState::LogEntry{b, cand_code, tree_index, dc.cur_plan});
    *dc.log_open = true;
    if (dc.moot && cand_code != 0xFFFFFFFFu && dc.moot->flag[b]) { dc.moot->flag[b] = 0; dc.moot->n_set--; }   // (it points elsewhere now)
    if (dc.moot && s < bs) dc.moot->n_le = -1;
    if (s < bs) { u.boot_counts[b] = 1; bs = s; }              // :3710-3719
  } else if (accept) {
    // the tree "string" (:3689-3707): looked up once per booked tree; the topology itself is remembered as
    // (prune node, candidate) and materialised after this prune node's scan only if some sample still points to it
    if (!looked_up) { tree_index = lookup(tree_index, cand_code); looked_up = true; }
    if (u.pending.empty() || u.pending.back().tree_index != tree_index) u.pending.push_back(UfbState::Pending{tree_index, cand_code});
    if (s < bs) { u.boot_counts[b] = 1; bs = s; }              // :3710-3719
    int64_t &bt = u.boot_trees[b];
    if (bt != tree_index) {
      if (bt >= 0 && --u.refs[(size_t)bt] == 0) u.store.erase(bt);
      u.refs[(size_t)tree_index]++;
      bt = tree_index;                                          // :3720
    }
  }
  if (s == bs) u.boot_counts[b]++;                              // :3728-3730
}

// One tree arriving at saveCurrentTree with length cur_len: its index in treels_logl, or -1 when nothing is booked.
// Default: the cut-off test, then a new index (iqtree.cpp:3343-3348).  -storetrees: looked up by topology first
// (:3302-3341; key(cand_code) = its canonical form); one met before is skipped unless the length improved on the recorded one,
// and then it goes on under its old index without the cut-off test.
template <class KeyFn>
int64_t Engine::ufb_book_tree(uint32_t cur_len, bool passes_cut, uint32_t cand_code, bool store_trees, KeyFn key)
{
  UfbState &u = *ufb_;
  if (store_trees) {
    const std::string &k = key(cand_code);
    auto it = u.topo_index.find(k);
    if (it != u.topo_index.end()) {
      u.duplicates++;
      if (cur_len >= u.treels[(size_t)it->second]) return -1;
      u.treels[(size_t)it->second] = cur_len;
      return it->second;
    }
    if (!passes_cut) return -1;
    u.topo_index.emplace(k, (int64_t)u.treels.size());
  } else if (!passes_cut) return -1;
  u.treels.push_back(cur_len);
  u.refs.push_back(0);
  return (int64_t)u.treels.size() - 1;
}

}  // namespace mpf
