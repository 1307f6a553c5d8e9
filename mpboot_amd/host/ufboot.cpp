// ufboot.cpp -- online UFBoot-MP bookkeeping around the batched SPR scans.
//
// Reference: with -bb, testInsertParsimony calls IQTree::saveCurrentTree for EVERY insertion test
// (sprparsimony.cpp:2163-2166 -> iqtree.cpp:3271-3785): cut-off filter, per-pattern lengths of the tentative tree,
// REPS against every bootstrap sample, and the per-sample update rule with its random tie-breaks (drawn from
// the same stream as the SPR tie-breaks).  Here the arithmetic runs on the device for a whole batch of scans
// (ufboot.hip) and the host replays only the order-dependent part: the candidates in scan order, and for each of
// them the few (sample) events where its score reaches that sample's running best.


#include "ufboot_common.hpp"

namespace mpf {

int Engine::ufboot_attach(int n_samples, const uint16_t *samples, double epsilon, int n_local, const int32_t *sample_ids,
                          mpf_ufb_exchange_fn exchange, void *exchange_arg)
{
  if (n_samples < 1 || !samples) { set_error("ufboot_attach: bad argument"); return MPF_E_INVALID; }
  if (!(epsilon > 0.0 && epsilon < 1.0)) {
    set_error("ufboot_attach: epsilon must lie in (0, 1) -- with integer scores every such value acts like the default 0.5");
    return MPF_E_UNSUPPORTED;
  }
  ufboot_detach();
  std::unique_ptr<UfbState> u(new UfbState());
  const int n_all = n_samples;
  u->B = n_all;
  if (n_local < 0) n_local = n_all;                        // unsharded: every sample is local, ids = identity
  u->Bl = n_local;
  u->ids.resize((size_t)n_local);
  u->ids_identity = n_local == n_all;
  for (int c = 0; c < n_local; c++) {
    u->ids[(size_t)c] = sample_ids ? sample_ids[c] : c;
    if (u->ids[(size_t)c] != c) u->ids_identity = false;
    if (u->ids[(size_t)c] < 0 || u->ids[(size_t)c] >= n_all) { set_error("ufboot_attach: sample id out of range"); return MPF_E_INVALID; }
  }
  u->exchange = exchange;
  u->exchange_arg = exchange_arg;
  n_samples = n_local;                                     // from here on: the local weight vectors
  // one extra column behind the local samples: the pattern frequencies in force now (IQTree's original_sample).  Its
  // REPS is the length of a tree on the original alignment, which re-weighted (ratchet) climbs book instead of their own
  u->Bp = round_up(n_samples + 1, kUfbColTile);
  u->eps = epsilon;
  uint32_t wmax = 0;
  for (size_t i = 0; i < (size_t)n_samples * (size_t)P_; i++) wmax = std::max<uint32_t>(wmax, samples[i]);
  std::vector<uint16_t> orig((size_t)P_);
  for (int p = 0; p < P_; p++) {
    if (wgt_[(size_t)p] > 65535) { set_error("ufboot_attach: pattern frequency above 65535"); return MPF_E_UNSUPPORTED; }
    orig[(size_t)p] = (uint16_t)wgt_[(size_t)p];
    wmax = std::max<uint32_t>(wmax, orig[(size_t)p]);
  }
  u->planes = wmax < 128 ? 1 : (wmax < 16384 ? 2 : 3);
  u->snk = sankoff_;
  u->Wp_s = sankoff_ ? round_up((g_.Wp + 31) / 32, 8) : 0;          // (weighted: g_.Wp counts patterns)
  const int nkb = (sankoff_ ? u->Wp_s : g_.Wp) / 2;
  u->plane_bytes = (size_t)nkb * (size_t)u->Bp * 64;
  UCHK(u->d_samples.reserve((size_t)(n_samples + 1) * (size_t)P_));
  UCHK(hipMemcpyAsync(u->d_samples.p, samples, (size_t)n_samples * (size_t)P_ * sizeof(uint16_t), hipMemcpyHostToDevice, st_));
  UCHK(hipMemcpyAsync(u->d_samples.p + (size_t)n_samples * (size_t)P_, orig.data(), (size_t)P_ * sizeof(uint16_t), hipMemcpyHostToDevice, st_));
  UCHK(hipStreamSynchronize(st_));                         // (orig is a local)
  UCHK(u->rt.reserve((size_t)u->Bp));
  UCHK(u->best.reserve((size_t)u->Bp));
  UCHK(u->evcount.reserve(4));
  UCHK(hipMemsetAsync(u->rt.p, 0, (size_t)u->Bp * sizeof(int32_t), st_));
  UCHK(hipStreamSynchronize(st_));
  u->boot_score.assign((size_t)n_all, UINT32_MAX);           // boot_logl = -LONG_MAX (iqtree.cpp:248)
  u->boot_counts.assign((size_t)n_all, 0);                   // :253
  u->boot_orig.assign((size_t)n_all, 0);                     // :254
  u->boot_trees.assign((size_t)n_all, -1);                   // :252
  u->attach_wgt = wgt_;
  ufb_pool_swap(*u);                               // scratch buffers of an earlier tracker, if any
  ufb_ = std::move(u);
  return ufb_layout_weights();
}

// (re)build the product's right-hand side for the packing in force: called at attach time and after every re-weighting
int Engine::ufb_layout_weights()
{
  UfbState &u = *ufb_;
  const int nkb = (u.snk ? u.Wp_s : g_.Wp) / 2;
  u.plane_bytes = (size_t)nkb * (size_t)u.Bp * 64;          // (Wp follows the packing)
  UCHK(u.wt.reserve(u.plane_bytes * (size_t)u.planes));
  UCHK(u.d_first.reserve((size_t)P_));
  UCHK(u.d_cur.reserve((size_t)P_));
  std::vector<int32_t> snk_first, snk_cur;
  if (u.snk) {
    // weighted engine: "site" = position of the pattern among the informative ones (the order of the cost vectors), every
    // informative pattern takes part whatever its current weight (the weights live in pwgt, not in the packing)
    snk_first.assign((size_t)P_, -1);
    snk_cur.assign((size_t)P_, 0);
    for (int j = 0; j < ninf_; j++) { snk_first[(size_t)inf_index_[(size_t)j]] = j; snk_cur[(size_t)inf_index_[(size_t)j]] = 1; }
  }
  UCHK(hipMemcpyAsync(u.d_first.p, u.snk ? snk_first.data() : first_site_.data(), (size_t)P_ * sizeof(int32_t), hipMemcpyHostToDevice, st_));
  UCHK(hipMemcpyAsync(u.d_cur.p, u.snk ? snk_cur.data() : wgt_.data(), (size_t)P_ * sizeof(int32_t), hipMemcpyHostToDevice, st_));
  UCHK(launch_ufb_layout(st_, u.d_samples.p, u.Bl + 1, P_, u.d_first.p, u.d_cur.p, u.wt.p, u.Bp, u.planes, u.plane_bytes));
  UCHK(hipStreamSynchronize(st_));
  u.rt_valid = false;
  return MPF_OK;
}

void Engine::ufboot_detach()
{
  if (ufb_ && std::getenv("MPF_UFB_PROFILE"))
    std::fprintf(stderr, "[ufboot] batches %llu events %llu stored %llu | ms: scan %.1f prep %.1f device %.1f (of which deferred bookkeeping beside the device %.1f) sort %.1f replay %.1f (topology lookups %llu: %.1f) rt %.1f (product kernels %.1f)\n",
                 (unsigned long long)ufb_->batches, (unsigned long long)ufb_->events, (unsigned long long)ufb_->stored, ufb_->t_scan,
                 ufb_->t_prep, ufb_->t_dev, ufb_->t_defer, ufb_->t_sort, ufb_->t_replay, (unsigned long long)ufb_->lookups, ufb_->t_lookup, ufb_->t_rt, ufb_->gemm_ms);
  if (ufb_) ufb_pool_swap(*ufb_);                  // keep the large scratch buffers for the next attach
  ufb_.reset();
}

void Engine::ufb_pool_swap(UfbState &u)
{
  u.masks.swap(ufb_pool_.masks);
  u.jmasks.swap(ufb_pool_.jmasks);
  u.sel2.swap(ufb_pool_.sel2);
  u.info.swap(ufb_pool_.info);
  u.C.swap(ufb_pool_.C);
  u.C2.swap(ufb_pool_.C2);
  u.ev.swap(ufb_pool_.ev);
  u.h_ev.swap(ufb_pool_.h_ev);
  u.h_info.swap(ufb_pool_.h_info);
  u.thr.swap(ufb_pool_.thr);
  u.home.swap(ufb_pool_.home);
  u.cmin.swap(ufb_pool_.cmin);
  u.pre.swap(ufb_pool_.pre);
  u.h_small.swap(ufb_pool_.h_small);
}

int Engine::ufboot_set_cutoff(double logl_cutoff)
{
  if (!ufb_) { set_error("no UFBoot tracker attached"); return MPF_E_STATE; }
  ufb_->logl_cutoff = logl_cutoff;
  return MPF_OK;
}

// params->no_hclimb1_bb (tools.cpp:795; iqtree.cpp:3280): 0 = climbs on other weights than the attach-time ones run without
// saveCurrentTree; takes effect at the next mpf_set_weights
int Engine::ufboot_set_ratchet_booking(int on)
{
  if (!ufb_) { set_error("no UFBoot tracker attached"); return MPF_E_STATE; }
  ufb_->ratchet_booking = on != 0;
  return MPF_OK;
}

// params->cutoff_from_btrees (tools.cpp:2442): the next cut-off is the smallest boot_tree_orig_logl (iqtree.cpp:1657-1660)
int Engine::ufboot_set_cutoff_from_btrees(int on)
{
  if (!ufb_) { set_error("no UFBoot tracker attached"); return MPF_E_STATE; }
  ufb_->cut_btrees = on != 0;
  return MPF_OK;
}
int Engine::ufboot_orig_logl(int32_t *out) const
{
  if (!ufb_) { set_error("no UFBoot tracker attached"); return MPF_E_STATE; }
  std::copy(ufb_->boot_orig.begin(), ufb_->boot_orig.end(), out);
  return MPF_OK;
}

// params->multiple_hits (tools.cpp, -mulhits): the update rule of iqtree.cpp:3498-3540 instead of the default one.  Before
// the first booked tree only.
int Engine::ufboot_set_mulhits(int on)
{
  if (!ufb_) { set_error("no UFBoot tracker attached"); return MPF_E_STATE; }
  if (!ufb_->treels.empty()) { set_error("ufboot_set_mulhits: trees have been booked under the other rule already"); return MPF_E_STATE; }
  ufb_->mulhits = on != 0;
  ufb_->hit_sets.assign(on ? (size_t)ufb_->B : 0, std::set<int64_t>());
  return MPF_OK;
}

// params->store_candidate_trees (-storetrees, iqtree.cpp:3302-3346).  Right after the attach.
int Engine::ufboot_set_store_trees(int on)
{
  if (!ufb_) { set_error("no UFBoot tracker attached"); return MPF_E_STATE; }
  if (!ufb_->treels.empty()) { set_error("ufboot_set_store_trees: trees have been booked without the topology map already"); return MPF_E_STATE; }
  ufb_->store_trees = on != 0;
  return MPF_OK;
}
int Engine::ufboot_duplicates(uint64_t *n) const
{
  if (!ufb_) return MPF_E_STATE;
  if (n) *n = ufb_->duplicates;
  return MPF_OK;
}

// params->store_top_boot_trees (-topboot N, with -mulhits): the rule of iqtree.cpp:3542-3585.  After mpf_ufboot_set_mulhits(1),
// before the first booked tree.
int Engine::ufboot_set_topboot(int n_top)
{
  if (!ufb_) { set_error("no UFBoot tracker attached"); return MPF_E_STATE; }
  if (!ufb_->mulhits) { set_error("ufboot_set_topboot: the top list belongs to the -mulhits rule (set it first)"); return MPF_E_STATE; }
  if (!ufb_->treels.empty()) { set_error("ufboot_set_topboot: trees have been booked already"); return MPF_E_STATE; }
  if (n_top < 0 || n_top > 4096) { set_error("ufboot_set_topboot: bad list length"); return MPF_E_INVALID; }
  ufb_->topboot = n_top;
  ufb_->top.assign(n_top ? (size_t)ufb_->B : 0, {});
  ufb_->top_thr.assign(n_top ? (size_t)ufb_->B : 0, -INT_MAX);            // iqtree.cpp:267
  return MPF_OK;
}

// params->distinct_iter_top_boot (-distinct_iter_top_boot k, without -mulhits): the rule of iqtree.cpp:3587-3680.  Before the
// first booked tree; the caller keeps mpf_ufboot_set_iteration in step with IQTree::curIt.
int Engine::ufboot_set_distinct_iter(int k)
{
  if (!ufb_) { set_error("no UFBoot tracker attached"); return MPF_E_STATE; }
  if (ufb_->mulhits) { set_error("ufboot_set_distinct_iter: not together with -mulhits (the reference ignores it there)"); return MPF_E_STATE; }
  if (!ufb_->treels.empty()) { set_error("ufboot_set_distinct_iter: trees have been booked already"); return MPF_E_STATE; }
  if (k < 0 || k > 4096) { set_error("ufboot_set_distinct_iter: bad list length"); return MPF_E_INVALID; }
  ufb_->distinct = k;
  ufb_->top.assign(k ? (size_t)ufb_->B : 0, {});
  ufb_->top_iter.assign(k ? (size_t)ufb_->B : 0, {});
  ufb_->top_thr.assign(k ? (size_t)ufb_->B : 0, -INT_MAX);                // iqtree.cpp:273
  return MPF_OK;
}

int Engine::ufboot_set_iteration(int cur_it)
{
  if (!ufb_) { set_error("no UFBoot tracker attached"); return MPF_E_STATE; }
  ufb_->cur_it = cur_it;
  return MPF_OK;
}

int Engine::ufboot_sample_iters(int sample, int32_t *iters, int cap, int *n) const
{
  if (!ufb_) { set_error("no UFBoot tracker attached"); return MPF_E_STATE; }
  if (!ufb_->distinct) { set_error("ufboot_sample_iters: the -distinct_iter_top_boot rule is not in force"); return MPF_E_STATE; }
  if (sample < 0 || sample >= ufb_->B || !n) { set_error("ufboot_sample_iters: bad argument"); return MPF_E_INVALID; }
  const auto &it = ufb_->top_iter[(size_t)sample];
  *n = (int)it.size();
  for (int i = 0; i < *n && i < cap && iters; i++) iters[i] = it[(size_t)i];
  return MPF_OK;
}

int Engine::ufboot_sample_top(int sample, int64_t *trees, int32_t *rell, int cap, int *n, int32_t *threshold) const
{
  if (!ufb_) { set_error("no UFBoot tracker attached"); return MPF_E_STATE; }
  if (!ufb_->topboot && !ufb_->distinct) { set_error("ufboot_sample_top: neither -topboot nor -distinct_iter_top_boot is in force"); return MPF_E_STATE; }
  if (sample < 0 || sample >= ufb_->B || !n) { set_error("ufboot_sample_top: bad argument"); return MPF_E_INVALID; }
  const auto &top = ufb_->top[(size_t)sample];
  *n = (int)top.size();
  for (int i = 0; i < *n && i < cap; i++) {
    if (trees) trees[i] = top[(size_t)i].first;
    if (rell) rell[i] = top[(size_t)i].second;
  }
  if (threshold) *threshold = ufb_->top_thr[(size_t)sample];
  return MPF_OK;
}

bool Engine::ufb_topboot_offer(uint32_t b, int32_t rell, int64_t tree_index, bool newly_added)
{
  UfbState &u = *ufb_;
  auto &top = u.top[b];
  int32_t &thr = u.top_thr[b];
  if (!newly_added) return false;                                          // "if newly added" (:3560)
  const bool full = (int)top.size() == u.topboot;
  if (full && !(rell > thr)) return false;
  if (full) {                                                              // :3571-3578
    const int64_t d = top.back().first;
    top.pop_back();
    if (--u.refs[(size_t)d] == 0) u.store.erase(d);
  }
  size_t pos = 0;
  while (pos < top.size() && !(top[pos].second < rell)) pos++;            // keep the list sorted decreasingly (:3564-3569)
  top.insert(top.begin() + (long)pos, std::make_pair(tree_index, rell));
  if (full) thr = top[(size_t)u.topboot - 1].second;
  else if (!(thr < rell)) thr = rell;                                      // :3570
  u.refs[(size_t)tree_index]++;
  return true;
}

int Engine::ufboot_sample_trees(int sample, int64_t *out, int cap, int *n) const
{
  if (!ufb_) { set_error("no UFBoot tracker attached"); return MPF_E_STATE; }
  if (!ufb_->mulhits) { set_error("ufboot_sample_trees: the -mulhits rule is not in force"); return MPF_E_STATE; }
  if (sample < 0 || sample >= ufb_->B || !n) { set_error("ufboot_sample_trees: bad argument"); return MPF_E_INVALID; }
  const std::set<int64_t> &st = ufb_->hit_sets[(size_t)sample];
  *n = (int)st.size();
  int k = 0;
  for (int64_t t : st) { if (k >= cap || !out) break; out[k++] = t; }
  return MPF_OK;
}

// What the reference's printTree(WT_TAXON_ID | WT_SORT_TAXA) string stands for (iqtree.cpp:3508): a canonical form of the
// unrooted topology.  Here: the tree hung from tip 1, an inner node written as -1 followed by its two subtrees, the one
// with the smaller tip number first.
// "top cutoff_percent %" rule of the main loop (reference iqtree.cpp:1662-1676; treels_logl.size() > 1000)
double Engine::ufboot_next_cutoff(int percent) const
{
  if (!ufb_) return 0.0;
  if (ufb_->cut_btrees) return (double)*std::min_element(ufb_->boot_orig.begin(), ufb_->boot_orig.end());     // :1657-1660
  const std::vector<uint32_t> &t = ufb_->treels;
  if (t.size() <= 1000) return ufb_->logl_cutoff;
  std::vector<uint32_t> l(t);
  // (the reference indexes l[size * percent / 100] unclamped, iqtree.cpp:1666: one past the end at 100 %; here the
  //  worst tree's score is the cut-off in that case)
  const size_t k = std::min(l.size() - 1, l.size() * (size_t)percent / 100);
  std::nth_element(l.begin(), l.begin() + (long)k, l.end());   // k-th smallest length = k-th largest logl
  return -(double)l[k];
}

int Engine::ufboot_n_samples() const { return ufb_ ? ufb_->B : 0; }
int64_t Engine::ufboot_n_trees() const { return ufb_ ? (int64_t)ufb_->treels.size() : 0; }

int Engine::ufboot_tree_logl(double *out) const
{
  if (!ufb_) { set_error("no UFBoot tracker attached"); return MPF_E_STATE; }
  for (size_t i = 0; i < ufb_->treels.size(); i++) out[i] = -(double)ufb_->treels[i];
  return MPF_OK;
}

int Engine::ufboot_state(double *boot_logl, int32_t *boot_counts, int32_t *boot_trees) const
{
  if (!ufb_) { set_error("no UFBoot tracker attached"); return MPF_E_STATE; }
  for (int b = 0; b < ufb_->B; b++) {
    if (boot_logl) boot_logl[b] = ufb_->boot_score[(size_t)b] == UINT32_MAX ? -(double)LONG_MAX : -(double)ufb_->boot_score[(size_t)b];
    if (boot_counts) boot_counts[b] = ufb_->boot_counts[(size_t)b];
    if (boot_trees) boot_trees[b] = (int32_t)ufb_->boot_trees[(size_t)b];
  }
  return MPF_OK;
}

int Engine::ufboot_tree(int64_t tree_index, int32_t *back) const
{
  if (!ufb_) { set_error("no UFBoot tracker attached"); return MPF_E_STATE; }
  auto it = ufb_->store.find(tree_index);
  if (it == ufb_->store.end()) { set_error("tree is not referenced by any bootstrap sample"); return MPF_E_INVALID; }
  std::memcpy(back, it->second.data(), 3 * (size_t)(2 * n_ - 1) * sizeof(int32_t));
  return MPF_OK;
}

// Books that another search chain of the SAME run keeps (iteration-parallel -bb: several engines / ranks run different iterations
// of IQTree::doTreeSearch and meet every few iterations, as the reference's MPI branches do, README.md:71-78): a sample for which
// the other chain holds a strictly shorter tree takes that tree over -- the rule saveCurrentTree applies to a strictly better
// tree (iqtree.cpp:3686, :3710-3720: boot_logl, boot_counts = 1 then counted once, boot_trees), with the tree entering
// treels_logl under its length on the original alignment like any booked tree.  Equal lengths keep the holder (no draw: the
// chains' streams are their own).  Default update rule, unsharded tracker, between two climbs.
int Engine::ufboot_adopt(int n_upd, const int32_t *sample, const uint32_t *score, const int32_t *tree_of, int n_trees, const int32_t *backs,
                         const uint32_t *lengths, int32_t *n_taken)
{
  if (!ufb_) { set_error("no UFBoot tracker attached"); return MPF_E_STATE; }
  UfbState &u = *ufb_;
  if (u.mulhits || u.topboot || u.distinct || u.store_trees || u.exchange || !u.ids_identity) {
    set_error("mpf_ufboot_adopt: default update rule on an unsharded tracker only");
    return MPF_E_UNSUPPORTED;
  }
  if (n_upd < 0 || n_trees < 0 || (n_upd && (!sample || !score || !tree_of)) || (n_trees && (!backs || !lengths))) { set_error("mpf_ufboot_adopt: bad argument"); return MPF_E_INVALID; }
  ufb_drain_log();
  const size_t nrec = 3 * (size_t)(2 * n_ - 1);
  std::vector<int64_t> idx((size_t)n_trees, -1);
  std::vector<int32_t> bk(nrec);
  std::string key;
  int taken = 0;
  for (int k = 0; k < n_upd; k++) {
    const int b = sample[k], t = tree_of[k];
    if (b < 0 || b >= u.B || t < 0 || t >= n_trees) { set_error("mpf_ufboot_adopt: index out of range"); return MPF_E_INVALID; }
    if (score[k] >= u.boot_score[(size_t)b]) continue;
    if (idx[(size_t)t] < 0) {
      std::memcpy(bk.data(), backs + (size_t)t * nrec, nrec * sizeof(int32_t));
      for (int v = 1; v <= 2 * n_ - 2; v++)
        for (int sl = 0; sl < (v <= n_ ? 1 : 3); sl++) {
          const int r = 3 * v + sl, bb = bk[(size_t)r];
          if (bb < 3 || bb >= (int)nrec || bk[(size_t)bb] != r) { set_error("mpf_ufboot_adopt: inconsistent back links"); return MPF_E_INVALID; }
        }
      canonical_topology(bk, key);
      auto ins = u.topo_index.emplace(key, (int64_t)u.treels.size());
      if (ins.second) { u.treels.push_back(lengths[t]); }
      idx[(size_t)t] = ins.first->second;
      if (u.refs.size() <= (size_t)idx[(size_t)t]) u.refs.resize((size_t)idx[(size_t)t] + 1 + u.refs.size() / 2, 0);
      if (!u.store.count(idx[(size_t)t])) u.store.emplace(idx[(size_t)t], bk);
    }
    const int64_t ti = idx[(size_t)t];
    u.boot_score[(size_t)b] = score[k];
    u.boot_counts[(size_t)b] = 2;                              // = 1 on the strict improvement, counted once more as an equal (:3710, :3728-3730)
    if (u.cut_btrees) u.boot_orig[(size_t)b] = -(int32_t)lengths[t];
    int64_t &bt = u.boot_trees[(size_t)b];
    if (bt != ti) {
      u.refs[(size_t)ti]++;
      if (bt >= 0 && --u.refs[(size_t)bt] == 0) u.store.erase(bt);
      bt = ti;
    }
    taken++;
  }
  for (int t = 0; t < n_trees; t++)                            // (a topology stored for nothing: no sample took it)
    if (idx[(size_t)t] >= 0 && u.refs[(size_t)idx[(size_t)t]] == 0) u.store.erase(idx[(size_t)t]);
  if (n_taken) *n_taken = taken;
  return MPF_OK;
}

int Engine::ufboot_counters(uint64_t *draws, uint64_t *events, uint64_t *gemm_rows, double *gemm_ms) const
{
  if (!ufb_) { set_error("no UFBoot tracker attached"); return MPF_E_STATE; }
  if (draws) *draws = ufb_->draws;
  if (events) *events = ufb_->events;
  if (gemm_rows) *gemm_rows = ufb_->gemm_rows;
  if (gemm_ms) *gemm_ms = ufb_->gemm_ms;
  return MPF_OK;
}

int Engine::ufb_reserve_scan(size_t n_idx)
{
  UfbState &u = *ufb_;
  const size_t rows = (size_t)round_up((int)std::max<size_t>(n_idx, 1), kUfbRowTile);
  UCHK(u.masks.reserve(rows * (size_t)g_.Wp));
  UCHK(u.info.reserve(rows));
  return MPF_OK;
}

// R_T[b] = sum_ptn w_b[ptn] * pattern_pars(current tree)[ptn]: the n-1 joins of a rooted traversal, one mask each
int Engine::ufb_current_tree_reps()
{
  UfbState &u = *ufb_;
  if (!views_valid_) { int rc = update_views(); if (rc) return rc; }
  std::vector<EvOp> ops;
  ops.push_back(EvOp{slot(start_), slot(back_[start_]), 0, 0});
  std::vector<int> stack;
  stack.push_back(back_[start_]);
  while (!stack.empty()) {
    const int r = stack.back();
    stack.pop_back();
    if (tip(r)) continue;
    const int a = back_[nx(r)], b = back_[nx(nx(r))];
    ops.push_back(EvOp{slot(a), slot(b), 0, 0});
    stack.push_back(a);
    stack.push_back(b);
  }
  const int rows = (int)ops.size(), rows_p = round_up(rows, kUfbRowTile);
  UCHK(d_evops_.reserve(ops.size()));
  UCHK(u.jmasks.reserve((size_t)rows_p * (size_t)g_.Wp));
  UCHK(u.C.reserve((size_t)rows_p * (size_t)u.Bp));
  UCHK(hipMemcpyAsync(d_evops_.p, ops.data(), ops.size() * sizeof(EvOp), hipMemcpyHostToDevice, st_));
  UCHK(hipMemsetAsync(u.jmasks.p + (size_t)rows * g_.Wp, 0, (size_t)(rows_p - rows) * g_.Wp * sizeof(uint32_t), st_));
  UCHK(launch_join_masks(st_, g_, d_vec_, d_evops_.p, rows, u.jmasks.p));
  for (int pl = 0; pl < u.planes; pl++)
    UCHK(launch_bitgemm(st_, u.jmasks.p, rows_p, g_.Wp, u.wt.p + (size_t)pl * u.plane_bytes, u.Bp, u.C.p, 1 << (7 * pl), pl > 0));
  UCHK(launch_colsum(st_, u.C.p, rows, u.Bp, u.rt.p));
  UCHK(hipStreamSynchronize(st_));     // ops (host vector) must outlive the copy
  u.rt_valid = true;
  return MPF_OK;
}

// topologies of the trees accepted during the scan of one prune node that are still some sample's boot tree
void Engine::ufb_flush_pending(const ScanPlan &pl)
{
  UfbState &u = *ufb_;
  for (const UfbState::Pending &pe : u.pending) {
    if (u.refs[(size_t)pe.tree_index] <= 0) continue;
    if (pe.cand == 0xFFFFFFFFu) {                    // the current tree itself (booked in front of the candidates)
      if (!u.store.count(pe.tree_index)) { u.store.emplace(pe.tree_index, back_); u.stored++; }
      continue;
    }
    ufb_store_tree(pe.tree_index, pe.cand < (uint32_t)pl.n_p ? pl.rec : back_[pl.rec], candidate_record(pl, (size_t)pe.cand));
  }
  u.pending.clear();
}

// the tentatively inserted topology of one insertion test: subtree p re-inserted into branch q
void Engine::ufb_candidate_topology(int p, int q, std::vector<int32_t> &bk) const
{
  bk = back_;
  auto hk = [&](int a, int b) { bk[(size_t)a] = b; bk[(size_t)b] = a; };
  const int a = bk[(size_t)nx(p)], b = bk[(size_t)nx(nx(p))];
  hk(a, b);
  const int r = bk[(size_t)q];
  hk(nx(p), q);
  hk(nx(nx(p)), r);
}

void Engine::ufb_store_tree(int64_t tree_index, int p, int q)
{
  UfbState &u = *ufb_;
  if (u.store.count(tree_index)) return;
  std::vector<int32_t> bk;
  ufb_candidate_topology(p, q, bk);
  u.store.emplace(tree_index, std::move(bk));
  u.stored++;
}

int Engine::ufb_stage_small(const std::vector<ScanPlan> &plans, int count)
{
  UfbState &u = *ufb_;
  uint32_t n_idx = 0, n_parts = 0, n_self = 0;
  for (int j = 0; j < count; j++) {
    const ScanPlan &pl = plans[(size_t)j];
    if (pl.self_idx >= 0) { n_self++; n_idx = std::max(n_idx, (uint32_t)pl.self_idx + 1u); }
    for (int pi = 0; pi < pl.n_parts; pi++) {
      n_idx = std::max(n_idx, pl.part_off[pi] + (uint32_t)pl.part_cnt[pi] + 1u);
      n_parts = std::max(n_parts, (uint32_t)pl.part_desc[pi] + 1u);
    }
  }
  // layout: thr[n_parts] | home[n_parts] | plan_end[n_parts] | best[Bp] | self[n_self]
  const size_t o_self = (size_t)3 * n_parts + (size_t)u.Bp, words = o_self + n_self;
  UCHK(u.h_small.reserve(words + 4));
  uint32_t *sm = u.h_small.p;
  std::memset(sm, 0, words * sizeof(uint32_t));                 // (padding columns of best: 0 -> never an event)
  uint32_t at = (uint32_t)o_self;
  for (int j = 0; j < count; j++) {
    const ScanPlan &pl = plans[(size_t)j];
    if (pl.self_idx >= 0) sm[at++] = (uint32_t)pl.self_idx;
    uint32_t end = pl.self_idx >= 0 ? (uint32_t)pl.self_idx + 1u : 0u;     // end of this prune node's index range
    for (int pi = 0; pi < pl.n_parts; pi++) end = std::max(end, pl.part_off[pi] + (uint32_t)pl.part_cnt[pi] + 1u);
    for (int pi = 0; pi < pl.n_parts; pi++) {
      const uint32_t d = (uint32_t)pl.part_desc[pi];
      sm[d] = UINT32_MAX;                                        // no cut-off in force: every candidate takes part
      sm[n_parts + d] = pl.part_off[pi] + (uint32_t)pl.part_cnt[pi];
      sm[2 * n_parts + d] = end;
    }
  }
  for (int c2 = 0; c2 < u.Bl; c2++) sm[(size_t)3 * n_parts + (size_t)c2] = ufb_event_bound((uint32_t)u.ids[(size_t)c2]);
  u.st_n_idx = n_idx; u.st_n_parts = n_parts; u.st_n_self = n_self; u.st_o_self = (uint32_t)o_self; u.st_words = (uint32_t)words;
  u.st_valid = true;
  u.st_dev = nullptr;
  return MPF_OK;
}

// eight consecutive samples that ALL tie with the current tree (the rule of a move-less sweep: 2e6 such bookings at C3): their
// draws in one vector step (lcg_block.hpp), thresholds gathered from the reciprocal table, counts + 1.  Returns the mask of the
// accepted ones, or -1 with nothing touched when the eight are not all ties (or the table is too short): the caller takes them
// one by one.
__attribute__((target("avx512f,avx512dq,avx512vl,avx2"))) static int self_tie8(const int32_t *rt, const uint32_t *bs, int32_t *cnt, const double *inv,
                                                                              size_t inv_n, uint64_t &state, const Lcg64Jump8 &jump)
{
  const __m256i r = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(rt));
  const __m256i b = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(bs));
  if (_mm256_cmpeq_epi32_mask(r, b) != 0xFF) return -1;
  const __m256i k = _mm256_add_epi32(_mm256_loadu_si256(reinterpret_cast<const __m256i *>(cnt)), _mm256_set1_epi32(1));
  if (inv_n > 0x7FFFFFFFu || _mm256_cmpgt_epi32_mask(k, _mm256_set1_epi32((int)inv_n - 1)) || _mm256_cmplt_epi32_mask(k, _mm256_set1_epi32(1))) return -1;
  const __m512d thr = _mm512_i32gather_pd(k, inv, 8);
  __m512d v;
  state = lcg64_draw8(state, jump, &v);
  const int acc = (int)_mm512_cmp_pd_mask(v, thr, _CMP_LE_OQ);
  _mm256_storeu_si256(reinterpret_cast<__m256i *>(cnt), k);
  return acc;
}

void Engine::ufb_self_default(const int32_t *rt, int64_t tree_index, int32_t cur_plan, bool &log_open, uint64_t &n_draws, SelfMoot *moot)
{
  UfbState &u = *ufb_;
  uint32_t *bsv = u.boot_score.data();
  int32_t *cnt = u.boot_counts.data();
  const int32_t *ids = u.ids.data();
  std::vector<double> &inv = u.inv;
  if (inv.size() < 2) { inv.assign(2, 0.0); inv[1] = 1.0; }
  const bool callback = rand_fn_ != nullptr;
  uint64_t st = rng_.state, draws = 0;
  static const Lcg64Jump8 jump;
  const bool by_eight = !callback && u.ids_identity && lcg64_have_avx512();
  // Samples whose tree ALREADY is the current topology and whose best score the current tree ties (SelfMoot): the draw is taken --
  // the stream moves on, boot_counts rises -- but its outcome decides nothing (an acceptance would point the sample at the tree it
  // points at).  A move-less sweep of the tree every sample holds is 2n - 2 visits of nothing else: one jump of the generator.
  uint8_t *mf = (moot && !callback && u.ids_identity) ? moot->flag.data() : nullptr;
  if (mf && moot->n_le < 0) {
    // how many samples the current tree reaches at all (its REPS <= their best): while it is none -- the visits of a tree well
    // above every sample's best -- a booking touches nothing (recounted whenever a best score has fallen)
    int k = 0;
    for (int c2 = 0; c2 < u.Bl; c2++) k += (uint32_t)rt[c2] <= bsv[c2];
    moot->n_le = k;
  }
  if (mf && moot->n_le == 0) return;
  if (mf && moot->n_set == u.Bl) {
    for (int c2 = 0; c2 < u.Bl; c2++) cnt[c2]++;
    rng_.state = st * moot->jump_a + moot->jump_c;
    n_draws += (uint64_t)u.Bl;
    return;
  }
  for (int c2 = 0; c2 < u.Bl; c2++) {
    if (mf && c2 + 8 <= u.Bl) {
      uint64_t f8;
      std::memcpy(&f8, mf + c2, 8);
      if (f8 == 0x0101010101010101ULL) {
        for (int k = 0; k < 8; k++) cnt[c2 + k]++;
        st = st * jump.a[7] + jump.c[7];
        draws += 8;
        c2 += 7;
        continue;
      }
    }
    if (by_eight && c2 + 8 <= u.Bl) {
      int acc = self_tie8(rt + c2, bsv + c2, cnt + c2, inv.data(), inv.size(), st, jump);
      if (acc >= 0) {
        draws += 8;
        for (; acc; acc &= acc - 1) {
          const int b8 = c2 + __builtin_ctz((unsigned)acc);
          u.log.push_back(UfbState::LogEntry{(uint32_t)b8, 0xFFFFFFFFu, tree_index, cur_plan});
          log_open = true;
          if (u.cut_btrees) u.boot_orig[(size_t)b8] = u.cur_logl_now;                         // :3716-3718
          if (mf && !mf[b8]) { mf[b8] = 1; moot->n_set++; }              // (it ties and now points at the current tree)
        }
        c2 += 7;
        continue;
      }
    }
    const uint32_t b = (uint32_t)ids[c2], s = (uint32_t)rt[c2], bs = bsv[b];
    if (s > bs) continue;
    bool accept = true;                                                 // rell > boot_logl + epsilon (:3686)
    if (s == bs) {                                                      // a tie: one draw against 1 / (boot_counts + 1) (:3687-3688)
      draws++;
      const size_t k = (size_t)cnt[b] + 1;
      if (k >= inv.size()) { const size_t from = inv.size(); inv.resize(std::max(k + 1, 2 * from)); for (size_t i = from; i < inv.size(); i++) inv[i] = 1.0 / (double)i; }
      double r;
      if (callback) r = rand_fn_(rand_arg_);
      else { st = st * 0x27bb2ee687b0b0fdULL + 3037000493ULL; r = (double)st * 5.4210108624275222e-20; }     // (TieRng::next)
      accept = (mf && mf[c2]) ? false : r <= inv[k];                    // (moot: the acceptance would change nothing -- not logged)
    }
    if (accept) {
      u.log.push_back(UfbState::LogEntry{b, 0xFFFFFFFFu, tree_index, cur_plan});
      log_open = true;
      if (u.cut_btrees) u.boot_orig[b] = u.cur_logl_now;                                  // :3716-3718
      if (s < bs) { cnt[b] = 1; bsv[b] = s; }                           // :3710-3719
      if (mf && !mf[c2]) { mf[c2] = 1; moot->n_set++; }
    }
    cnt[b]++;                                                           // :3728-3730 (s equals the sample's best here in either case)
  }
  if (!callback) rng_.state = st;
  n_draws += draws;
}

// The deferred half of the default update rule (iqtree.cpp:3689-3707, :3720): the tree "string" of a booked tree -- looked up
// once, at its first acceptance --, boot_trees[b], the reference counts and the topologies to keep.  Runs against the topology
// the log was written under (the climb may have moved on by one move since).
void Engine::ufb_drain_log()
{
  UfbState &u = *ufb_;
  if (u.log.empty()) return;
  DrainScratch &sc = drain_scratch_;
  ufb_drain(u.log, u.log_back, u.log_epoch, *u.log_plans, sc);
  u.lookups += sc.lookups; u.stored += sc.stored; u.t_lookup += sc.t_lookup;
  sc.lookups = sc.stored = 0;
  sc.t_lookup = 0;
  u.log.clear();
}

// pllOptimizeSprParsimony's sweep loop (reference sprparsimony.cpp:3295-3316) with perSiteScores = 1, i.e. with
// saveCurrentTree after every insertion test.  Same speculative batching as Engine::spr_sweeps.
int Engine::spr_sweeps_ufboot(int mintrav, int maxtrav, uint32_t randomMP, uint32_t *final_score)
{
  UfbState &u = *ufb_;
  if (ufb_fast_ && ufb_pipe_ && !u.ratchet && !u.store_trees && !u.mulhits && !u.distinct && !u.topboot && u.logl_cutoff == 0.0 &&
      host_poll_ && !timing_ && !check_counts_)
    return spr_sweeps_ufboot_pipe(mintrav, maxtrav, randomMP, final_score);
  uint32_t startMP = randomMP;
  unsigned iter_hits = 1;
  const int total = 2 * n_ - 2;
  // ---- the quiet stretch of a climb under a cut-off.  Nothing above the cut-off reaches the bookkeeping (iqtree.cpp:3343: the
  // filter sits in front of everything saveCurrentTree does, and without -storetrees nothing else looks at such a tree), so as
  // long as neither the current tree nor any insertion test of the visit at hand gives a tree of at most the largest admissible
  // length, the tracked climb IS the plain one -- same draws, same moves -- and runs as one: in k_climb while moves are dense,
  // as whole-chip cost-only batches otherwise (Engine::spr_sweeps_run).  That is most of a later search iteration: the
  // perturbed tree starts far above the cut-off (top 10 % of the saved trees) and climbs back towards it.  The first visit
  // with an admissible insertion test moves to that tree for certain (it is strictly shorter than the current one), and from
  // there on every visit books the current tree: one hand-over per climb, in front of that visit.
  // Re-weighted (ratchet) climbs are filtered by the length booked LAST (iqtree.cpp:3283-3295): if the start tree's length on the
  // original alignment fails the cut-off, the first booking closes the gate and nothing of the climb is booked at all.
  int resume_i = 0;                                // > 0: the sweep under way continues at this visit (startMP, iter_hits as handed over)
  if (ufb_quiet_ && u.logl_cutoff != 0.0 && !u.store_trees && !u.snk) {
    const double lim0 = -u.logl_cutoff + 1e-4;
    const uint32_t mp_max0 = lim0 <= 0.0 ? 0u : (uint32_t)std::ceil(lim0) - 1u;
    bool whole = lim0 <= 0.0;                      // nothing can pass
    if (u.ratchet && !whole) {
      if (!u.rt_valid) { int rc = ufb_current_tree_reps(); if (rc) return rc; }
      UCHK(u.h_col.reserve(4));
      UCHK(hipMemcpyAsync(u.h_col.p, u.rt.p + u.Bl, sizeof(int32_t), hipMemcpyDeviceToHost, st_));
      UCHK(hipStreamSynchronize(st_));
      u.rt_orig = (uint32_t)u.h_col.p[0];
      whole = u.rt_orig > mp_max0;
    }
    if (whole || !u.ratchet) {
      SweepCursor c;
      c.randomMP = randomMP;
      // (the batch policy remembers the move-less sweeps the climb before ended with; this one starts from a perturbed tree, far
      //  above the cut-off: moves are dense again -- start as an engine without a history does, in the persistent kernel)
      gap_est_ = -1.0;
      since_move_ = 0;
      int rc = spr_sweeps_run(mintrav, maxtrav, c, whole ? 0u : mp_max0);
      if (rc) return rc;
      u.rt_valid = false;                          // (R_T is not carried through the stretch: made from scratch at the hand-over)
      ufb_stat_quiet_++;
      if (u.ratchet) { u.stale_len = u.rt_orig; u.gate_closed = true; }
      if (!c.stopped) {
        if (final_score) *final_score = c.randomMP;
        return MPF_OK;
      }
      randomMP = c.randomMP;
      startMP = c.startMP;
      iter_hits = c.iter_hits;
      resume_i = c.i;
    }
  }
  // (an early return may leave a chained batch in flight: wait for it and start over from the topology, as the pipelined climb does)
  struct Abort {
    Engine *e;
    bool ok = false;
    ~Abort()
    {
      if (ok) return;
      (void)hipStreamSynchronize(e->st_);
      e->walk_async_ = false;
      e->n_walk_ = 0;
      e->walk_out_ = 0;
      e->cnt_copy_pending_ = false;
      e->pending_scores_ = false;
      e->invalidate_all();
      if (e->ufb_) { e->ufb_->log.clear(); e->ufb_->rt_valid = false; }
    }
  } abort_guard{this};
  std::vector<ScanPlan> plans_buf[2];                // (the deferred log of a batch names its plans while the next batch is planned)
  int plans_cur = 0;
  const uint32_t *out = nullptr;
  if (u.exchange) {
    // sample-sharded run: every rank must cut the climb into the same batches -- start from a fixed batch policy
    // state instead of this engine's own history
    gap_est_ = -1.0;
    since_move_ = 0;
  }
  // (under a cut-off every batch is two waits -- the product is compacted between them -- and the moves that are left lie hundreds of
  //  prune nodes apart: a batch that restarts at a handful of prune nodes behind every move pays the waits six times over)
  const int batch_floor = (u.logl_cutoff != 0.0 && !u.exchange) ? std::min(total, ufb_cut_batch_) : 1;
  int batch = std::max(first_batch(), batch_floor);
  std::vector<UfbEvent> events, ev_tmp;
  std::vector<uint32_t> ev_count;
  std::vector<uint32_t> small, sel_rows, crow, self_list;
  std::vector<int32_t> mh_bk;                      // -mulhits: a candidate's topology and its canonical form
  std::string mh_key;
  bool have_C = false;
  uint32_t exchange_tag = 0;
  if (!u.rt_valid) { int rc = ufb_current_tree_reps(); if (rc) return rc; }
  const bool ratchet = u.ratchet;
  const bool store_trees = u.store_trees;          // -storetrees (iqtree.cpp:3302-3346)
  const bool host_self = !u.exchange;              // (sample-sharded: R_T lives in pieces on the ranks, the events are exchanged anyway)
  // default update rule: the part of an acceptance that no draw and no search decision depends on is logged and worked off
  // beside the next batch's device work (ufb_drain_log)
  const bool defer = ufb_fast_ && !ratchet && !store_trees && !u.mulhits && !u.distinct && !u.topboot;
  bool log_open = false;                           // acceptances logged since the last end-of-prune-node mark
  const int oc = u.Bl;                             // the column of the original pattern frequencies
  auto read_rt_orig = [&]() -> int {
    UCHK(u.h_col.reserve(4));
    UCHK(hipMemcpyAsync(u.h_col.p, u.rt.p + oc, sizeof(int32_t), hipMemcpyDeviceToHost, st_));
    UCHK(hipStreamSynchronize(st_));
    u.rt_orig = (uint32_t)u.h_col.p[0];
    return MPF_OK;
  };
  if (ratchet) {
    // what the IQ-TREE kernel left in _pattern_pars before this climb: the start tree (optimizeAllBranches ->
    // computeParsimony on the perturbed alignment, iqtree.cpp:1712-1714); its REPS against original_sample is the
    // start tree's length on the original alignment
    int rc = read_rt_orig();
    if (rc) return rc;
    u.stale_len = u.rt_orig;
    u.gate_closed = false;
  }
  SelfMoot moot;
  bool moot_on = false;
  std::vector<int32_t> lcol;                       // ratchet: the original-frequency column of the product, per mask row
  uint32_t sw_mp_max = 0;
  do {
    int i = 1;
    if (resume_i > 0) { i = resume_i; resume_i = 0; }          // (the sweep the quiet stretch handed over: order and startMP stand)
    else { startMP = randomMP; node_rectifier(); }
    // (for UfbState::quiet_topo: is this a COMPLETE sweep of one topology without a single candidate event?)
    bool sw_full = i == 1, sw_moved = false;
    uint64_t sw_events = 0;
    while (i <= total && !visits_out()) {
      const int hi = visits_cap(i, std::min(total, i + batch - 1));
      double t0 = now_ms();
      const double tr_scan0 = u.t_scan, tr_dev0 = u.t_dev;      // (MPF_UFB_TRACE=1: one line per batch)
      ufb_stat_batches_++;
      plans_cur ^= 1;
      std::vector<ScanPlan> &plans = plans_buf[plans_cur];
      // cut-off filter (reference iqtree.cpp:3343): a candidate is saved iff  -mp > logl_cutoff - 1e-4
      const bool have_cut = u.logl_cutoff != 0.0;
      // one dispatch chain per batch: without a cut-off nothing the host would read from the scan decides what is
      // multiplied, so the product and the extraction are enqueued right behind the scan and the host waits ONCE
      scan_masks_ = true;
      ufb_async_ = ufb_fast_ && !have_cut && !ratchet && !store_trees;
      int rc = scan_batch(plans, nodep_.data() + i, hi - i + 1, mintrav, maxtrav, &out);
      scan_masks_ = false;
      ufb_async_ = false;
      if (rc) return rc;
      const bool chained = walk_async_;            // the scan is in flight, `out` is not there yet
      if (!chained) ufb_drain_log();               // (a chained batch works the log off while the device runs)
      double t1 = now_ms();
      u.t_scan += t1 - t0;
      u.batches++;
      const int np = hi - i + 1;
      // ---- the first prune node with a strictly better candidate ends the batch for certain: nothing behind it
      //      needs REPS (a tie may end it earlier; then the tail of the product is simply not used)
      int jstar = np - 1;
      for (int j = 0; j < np && !chained; j++) {
        const ScanPlan &pl = plans[(size_t)j];
        uint32_t m = UINT32_MAX;
        for (int pi = 0; pi < pl.n_parts; pi++)
          for (int k = 0; k < pl.part_cnt[pi]; k++) m = std::min(m, out[pl.part_off[pi] + (uint32_t)k]);
        if (m != UINT32_MAX && pl.base + m < randomMP) { jstar = j; break; }
      }
      const double lim = -u.logl_cutoff + 1e-4;
      const uint32_t mp_max = have_cut ? (lim <= 0.0 ? 0u : (uint32_t)std::ceil(lim) - 1u) : UINT32_MAX;
      const bool none_pass = have_cut && lim <= 0.0;
      // ---- device: REPS of every candidate of plans [0, jstar], then the (candidate, sample) events
      uint32_t n_idx = 0, n_parts = 0;
      self_list.clear();
      for (int j = 0; j <= jstar; j++) {
        const ScanPlan &pl = plans[(size_t)j];
        if (pl.self_idx >= 0) { self_list.push_back((uint32_t)pl.self_idx); n_idx = std::max(n_idx, (uint32_t)pl.self_idx + 1u); }
        for (int pi = 0; pi < pl.n_parts; pi++) {
          n_idx = std::max(n_idx, pl.part_off[pi] + (uint32_t)pl.part_cnt[pi] + 1u);
          n_parts = std::max(n_parts, (uint32_t)pl.part_desc[pi] + 1u);
        }
      }
      uint32_t n_ev = 0, n_eager = 0;
      t0 = now_ms();
      u.t_prep += t0 - t1;
      // with a cut-off only the saved candidates (and the home rows of their parts) are multiplied: this lists their
      // mask rows (`sel_rows`), crow maps a scan output index to its row of C
      const uint2 *hinfo = u.h_info.p;             // (a chained batch fills it -- and may move it -- later)
      // (re-weighted climbs are filtered by the length booked last, not by the candidate's own: every row is multiplied,
      //  until a booked tree fails the cut-off -- from then on nothing of this climb is booked)
      // (-storetrees: a topology met before is booked again when its length improved, whatever the cut-off says -- every row
      //  is multiplied and the replay decides)
      const bool skip_product = store_trees ? false : ratchet ? (u.gate_closed || none_pass || (have_cut && u.stale_len > mp_max)) : none_pass;
      const bool compact = have_cut && !none_pass && !ratchet && !store_trees;
      uint32_t n_rows = n_idx;
      // A topology whose complete move-less sweep has produced NO candidate event before, under a cut-off at least as wide: the
      // samples' best scores only ever fall and the admissible set only shrinks, so none of its insertion tests can reach any
      // sample now either -- nothing to multiply, nothing to extract (UfbState::quiet_topo; the bookings themselves -- treels,
      // the current tree's visits, every draw -- go on as ever).  Searches return to their local optima again and again.
      bool memo = false;
      if (compact && defer && host_self && ufb_memo_ && !u.quiet_topo.empty()) {
        if (u.self_key_epoch != (uint64_t)topo_epoch_) { canonical_topology(back_, u.self_key); u.self_key_epoch = (uint64_t)topo_epoch_; }
        const auto it = u.quiet_topo.find(u.quiet_key(mintrav, maxtrav, n_));
        memo = it != u.quiet_topo.end() && mp_max <= it->second;
      }
      if (memo) { n_rows = 0; u.memo_batches++; }
      if (compact && !memo) {
        sel_rows.clear();
        crow.assign((size_t)n_idx, 0xFFFFFFFFu);
        for (int j = 0; j <= jstar; j++) {
          const ScanPlan &pl = plans[(size_t)j];
          if (mp_max < pl.base) continue;
          const uint32_t lim_cost = mp_max - pl.base;
          for (int pi = 0; pi < pl.n_parts; pi++) {
            bool any = false;
            for (int k = 0; k < pl.part_cnt[pi]; k++) {
              const uint32_t idx = pl.part_off[pi] + (uint32_t)k;
              if (out[idx] > lim_cost) continue;
              crow[idx] = (uint32_t)sel_rows.size();
              sel_rows.push_back(hinfo[idx].x);
              any = true;
            }
            if (any) {
              const uint32_t hidx = pl.part_off[pi] + (uint32_t)pl.part_cnt[pi];
              crow[hidx] = (uint32_t)sel_rows.size();
              sel_rows.push_back(hidx);
            }
          }
        }
        n_rows = (uint32_t)sel_rows.size();
      }
      have_C = false;
      bool ran_events = false;
      // the current tree is booked in front of every prune node's candidates (sprparsimony.cpp:2285-2289) with its own
      // length (= randomMP): it takes part unless that length fails the cut-off (ratchet climbs: decided in the replay)
      const bool self_pass = !self_list.empty() && !skip_product && (ratchet || store_trees || randomMP <= mp_max);
      if (chained && !(n_idx > 0 && !skip_product && !compact)) { set_error("online UFBoot: chained batch without a product"); return MPF_E_STATE; }
      if (memo) {
        if (self_pass) {                               // R_T for the host's walk over the current tree's bookings
          UCHK(u.h_rt.reserve((size_t)u.Bp));
          UCHK(hipMemcpyAsync(u.h_rt.p, u.rt.p, (size_t)u.Bl * sizeof(int32_t), hipMemcpyDeviceToHost, st_));
          UCHK(hipStreamSynchronize(st_));
        }
        ran_events = true;
        events.clear();
        t1 = now_ms();
        u.t_dev += t1 - t0;
      } else if (n_idx > 0 && !skip_product && (n_rows > 0 || self_pass)) {
        const int rows_p = round_up((int)std::max<uint32_t>(n_rows, 1u), chained ? ufb_row_padding((int)n_rows, u.Bp) : kUfbRowTile);
        // (a batch whose prune nodes have no insertion test at all -- a five-taxon tree at radius 1 -- still books the current
        //  tree at every visit: the scan launch that normally provides the mask / info buffers was skipped)
        { int rc2 = ufb_reserve_scan(n_idx); if (rc2) return rc2; }
        // staging: thr[n_parts] | home[n_parts] | best[Bp] | crow[n_idx] | sel[rows_p] | self[n_self]
        const size_t o_crow = (size_t)2 * n_parts + (size_t)u.Bp, o_sel = o_crow + (compact ? (size_t)n_idx : 0);
        const size_t o_self = o_sel + (compact ? (size_t)rows_p : 0);
        const size_t o_cnt = o_self + self_list.size();          // the event counter: a zero word of this upload (no memset dispatch)
        UCHK(u.C.reserve((size_t)rows_p * (size_t)u.Bp));
        const uint32_t nch = ufb_chunks(n_idx);
        UCHK(u.cmin.reserve((size_t)nch * (size_t)u.Bp));
        UCHK(u.pre.reserve((size_t)nch * (size_t)u.Bp));
        if (u.ev.cap == 0) { const size_t c0 = (size_t)std::min<int64_t>(ufb_event_cap_, 1 << 18); UCHK(u.ev.reserve(c0)); UCHK(u.h_ev.reserve(c0)); }
        const uint32_t *d_thr = nullptr, *d_home = nullptr, *d_best = nullptr, *d_crow = nullptr, *d_sel = nullptr, *d_self = nullptr;
        if (chained) {
          // the staging block normally went up with the refresh's own upload (Engine::scan_batch); a batch on valid views has none
          if (!u.st_valid) { int rc2 = ufb_stage_small(plans, np); if (rc2) return rc2; }
          if (u.st_n_idx != n_idx || u.st_n_parts != n_parts || u.st_n_self != self_list.size()) { set_error("online UFBoot: staged block out of step with the batch"); return MPF_E_STATE; }
          const uint32_t *dsm = u.st_dev;
          if (!dsm) {
            UCHK(u.thr.reserve((size_t)u.st_words + 4));
            UCHK(hipMemcpyAsync(u.thr.p, u.h_small.p, (size_t)u.st_words * sizeof(uint32_t), hipMemcpyHostToDevice, st_));
            dsm = u.thr.p;
          }
          d_thr = dsm; d_home = dsm + n_parts; d_best = dsm + 3 * n_parts; d_self = dsm + u.st_o_self;      // (dsm + 2 n_parts: the parts' prune-node ends)
        } else {
        small.assign(o_cnt + 1, 0u);
        std::copy(self_list.begin(), self_list.end(), small.begin() + (long)o_self);
        for (int j = 0; j <= jstar; j++) {
          const ScanPlan &pl = plans[(size_t)j];
          for (int pi = 0; pi < pl.n_parts; pi++) {
            const uint32_t d = (uint32_t)pl.part_desc[pi];
            small[d] = (!have_cut || ratchet || store_trees) ? UINT32_MAX : (mp_max >= pl.base ? mp_max - pl.base + 1u : 0u);   // max cost + 1
            small[n_parts + d] = pl.part_off[pi] + (uint32_t)pl.part_cnt[pi];
          }
        }
        for (int c2 = 0; c2 < u.Bl; c2++) small[(size_t)2 * n_parts + (size_t)c2] = ufb_event_bound((uint32_t)u.ids[(size_t)c2]);
        if (compact) {
          std::memcpy(small.data() + o_crow, crow.data(), (size_t)n_idx * sizeof(uint32_t));
          std::memcpy(small.data() + o_sel, sel_rows.data(), sel_rows.size() * sizeof(uint32_t));      // padding rows multiply mask row 0
        }
        UCHK(u.h_small.reserve(small.size() + 4));
        std::memcpy(u.h_small.p, small.data(), small.size() * sizeof(uint32_t));
        UCHK(u.thr.reserve(small.size() + 4));
        UCHK(hipMemcpyAsync(u.thr.p, u.h_small.p, small.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st_));
        d_thr = u.thr.p; d_home = u.thr.p + n_parts; d_best = u.thr.p + 2 * n_parts;
        d_crow = compact ? u.thr.p + o_crow : nullptr; d_sel = compact ? u.thr.p + o_sel : nullptr;
        d_self = u.thr.p + o_self;
        }
        if (chained) {
          // ---- prep (C <- 0, counter <- 0, the current tree's slots) -> product -> extraction, whose last workgroup writes the
          //      events, the scan's costs (with the refresh's mutation counts), info and R_T into the host's pinned buffers
          uint32_t *d_evcount = d_done_.p + 48, *d_fin = d_done_.p + 32;
          UCHK(launch_ufb_prep(st_, u.C.p, (size_t)rows_p * (size_t)u.Bp, u.info.p, d_self, (uint32_t)self_list.size(),
                               (self_pass && !host_self) ? 0xFFFFFFFEu : 0xFFFFFFFFu, d_evcount));
          if (timing_) UCHK(hipEventRecord(ev2_, st_));
          for (int pl = 0; pl < u.planes; pl++)
            UCHK(launch_bitgemm(st_, u.masks.p, rows_p, g_.Wp, u.wt.p + (size_t)pl * u.plane_bytes, u.Bp, u.C.p, 1 << (7 * pl), 1));
          u.gemm_rows += (uint64_t)rows_p;
          have_C = true;
          ran_events = true;
          if (timing_) UCHK(hipEventRecord(ev3_, st_));
          n_eager = (uint32_t)std::min<size_t>(u.ev.cap, 4096);
          UCHK(u.h_ev.reserve((size_t)n_eager));
          UCHK(u.h_flag.reserve(4));
          UCHK(u.h_info.reserve(walk_async_nout_));
          UfbPublishArgs pa;
          if (cnt_copy_pending_) { pa.src[0] = d_cnt(); pa.dst[0] = h_cnt(); pa.words[0] = (uint32_t)(out_off() + walk_async_nout_); }
          else { pa.src[0] = d_out(); pa.dst[0] = h_out(); pa.words[0] = (uint32_t)walk_async_nout_; }
          cnt_copy_pending_ = false;
          pa.src[1] = reinterpret_cast<const uint32_t *>(u.info.p);
          pa.dst[1] = reinterpret_cast<uint32_t *>(u.h_info.p);
          pa.words[1] = (uint32_t)(2 * walk_async_nout_);
          if (host_self && self_pass) {                // R_T for the host's own walk over the current tree's bookings
            UCHK(u.h_rt.reserve((size_t)u.Bp));
            pa.src[2] = reinterpret_cast<const uint32_t *>(u.rt.p);
            pa.dst[2] = reinterpret_cast<uint32_t *>(u.h_rt.p);
            pa.words[2] = (uint32_t)u.Bl;
          }
          pa.h_ev = u.h_ev.p;
          pa.h_ev_cap = n_eager;
          pa.h_flag = u.h_flag.p;
          pa.done = d_fin;
          __atomic_store_n(u.h_flag.p + 1, 0u, __ATOMIC_RELAXED);
          UCHK(launch_ufb_events_publish(st_, u.info.p, d_out(), d_thr, d_home, nullptr, u.C.p, u.Bp, u.Bl, u.rt.p, d_best, n_idx, u.cmin.p, u.pre.p,
                                         u.ev.p, (uint32_t)u.ev.cap, d_evcount, (u.topboot || u.distinct) ? 1 : 0, pa));
          { const double td = now_ms(); ufb_drain_log(); u.t_defer += now_ms() - td; }
          if (!(host_poll_ && !timing_ && wait_host_flag(u.h_flag.p + 1))) {
            UCHK(hipStreamSynchronize(st_));
            if (__atomic_load_n(u.h_flag.p + 1, __ATOMIC_ACQUIRE) != 1u) { set_error("online UFBoot: the extraction kernel did not publish its results"); return MPF_E_STATE; }
          }
          n_ev = u.h_flag.p[0];
          { int rc2 = run_walks_finish(plans, &out); if (rc2) return rc2; }
          hinfo = u.h_info.p;
          if (timing_) {
            UCHK(hipStreamSynchronize(st_));
            float ms = 0;
            if (hipEventElapsedTime(&ms, ev2_, ev3_) == hipSuccess) u.gemm_ms += ms;
          }
          if (n_ev > u.ev.cap) {
            // more events than room: grow and extract again (the product is still there)
            UCHK(u.ev.reserve((size_t)n_ev));
            UCHK(hipMemsetAsync(d_evcount, 0, sizeof(uint32_t), st_));
            UCHK(launch_ufb_events(st_, u.info.p, d_out(), d_thr, d_home, nullptr, u.C.p, u.Bp, u.Bl, u.rt.p, d_best, n_idx, u.cmin.p, u.pre.p,
                                   u.ev.p, (uint32_t)u.ev.cap, d_evcount, (u.topboot || u.distinct) ? 1 : 0));
            n_eager = 0;
          }
          if (n_ev > n_eager) {
            if (u.h_ev.cap < (size_t)n_ev) n_eager = 0;       // (a grown pinned buffer starts empty)
            UCHK(u.h_ev.reserve((size_t)n_ev));
            UCHK(hipMemcpyAsync(u.h_ev.p + n_eager, u.ev.p + n_eager, (size_t)(n_ev - n_eager) * sizeof(UfbEvent), hipMemcpyDeviceToHost, st_));
            UCHK(hipStreamSynchronize(st_));
          }
        } else {
        UCHK(launch_ufb_self(st_, u.info.p, d_self, (uint32_t)self_list.size(), (self_pass && !host_self) ? 0xFFFFFFFEu : 0xFFFFFFFFu));
        if (host_self && self_pass) {                  // R_T for the host's own walk over the current tree's bookings
          UCHK(u.h_rt.reserve((size_t)u.Bp));
          UCHK(hipMemcpyAsync(u.h_rt.p, u.rt.p, (size_t)u.Bl * sizeof(int32_t), hipMemcpyDeviceToHost, st_));   // synchronised with the event count below
        }
        if (timing_) UCHK(hipEventRecord(ev2_, st_));
        if (n_rows > 0) {
          for (int pl = 0; pl < u.planes; pl++)
            UCHK(launch_bitgemm(st_, u.masks.p, rows_p, g_.Wp, u.wt.p + (size_t)pl * u.plane_bytes, u.Bp, u.C.p, 1 << (7 * pl), pl > 0, d_sel));
          u.gemm_rows += (uint64_t)rows_p;
          have_C = true;
        }
        if (timing_) UCHK(hipEventRecord(ev3_, st_));
        ran_events = true;
        if (ratchet && have_C) {
          UCHK(u.d_col.reserve((size_t)rows_p));
          UCHK(u.h_col.reserve((size_t)rows_p));
          UCHK(launch_ufb_column(st_, u.C.p, u.Bp, oc, n_rows, u.d_col.p));
          UCHK(hipMemcpyAsync(u.h_col.p, u.d_col.p, (size_t)n_rows * sizeof(int32_t), hipMemcpyDeviceToHost, st_));   // synchronised with the event count below
        }
        uint32_t *d_evcount = u.thr.p + o_cnt;
        for (bool again = false;; again = true) {
          if (again) UCHK(hipMemsetAsync(d_evcount, 0, sizeof(uint32_t), st_));
          UCHK(launch_ufb_events(st_, u.info.p, d_out(), d_thr, d_home, d_crow, u.C.p, u.Bp, u.Bl, u.rt.p, d_best, n_idx, u.cmin.p, u.pre.p,
                                 u.ev.p, (uint32_t)u.ev.cap, d_evcount, (u.topboot || u.distinct || store_trees) ? 1 : 0));
          UCHK(hipMemcpyAsync(u.h_small.p, d_evcount, sizeof(uint32_t), hipMemcpyDeviceToHost, st_));
          // the first few thousand events ride along with their count: the batches of a climb need no second round trip
          n_eager = (uint32_t)std::min<size_t>(u.ev.cap, 4096);
          UCHK(u.h_ev.reserve((size_t)n_eager));
          UCHK(hipMemcpyAsync(u.h_ev.p, u.ev.p, (size_t)n_eager * sizeof(UfbEvent), hipMemcpyDeviceToHost, st_));
          UCHK(hipStreamSynchronize(st_));
          n_ev = u.h_small.p[0];
          if (n_ev <= u.ev.cap) break;
          UCHK(u.ev.reserve((size_t)n_ev));              // more events than room: grow and extract again
          UCHK(u.h_ev.reserve((size_t)n_ev));
        }
        if (timing_) {
          float ms = 0;
          if (hipEventElapsedTime(&ms, ev2_, ev3_) == hipSuccess) u.gemm_ms += ms;
        }
        if (n_ev > n_eager) {
          if (u.h_ev.cap < (size_t)n_ev) n_eager = 0;       // (a grown pinned buffer starts empty)
          UCHK(u.h_ev.reserve((size_t)n_ev));
          UCHK(hipMemcpyAsync(u.h_ev.p + n_eager, u.ev.p + n_eager, (size_t)(n_ev - n_eager) * sizeof(UfbEvent), hipMemcpyDeviceToHost, st_));
          UCHK(hipStreamSynchronize(st_));
        }
        }
        t1 = now_ms();
        u.t_dev += t1 - t0;
        if (ratchet && have_C) lcol.assign(u.h_col.p, u.h_col.p + n_rows);
        // a chained batch was multiplied and extracted as a whole: what lies behind the first prune node with a strictly better
        // candidate is never replayed (the batch ends there at the latest) -- dropped before the sort
        uint32_t idx_cut = n_idx;
        if (chained) {
          int js = np - 1;
          for (int j = 0; j < np; j++) {
            const ScanPlan &pl = plans[(size_t)j];
            uint32_t m = UINT32_MAX;
            for (int pi = 0; pi < pl.n_parts; pi++)
              for (int k = 0; k < pl.part_cnt[pi]; k++) m = std::min(m, out[pl.part_off[pi] + (uint32_t)k]);
            if (m != UINT32_MAX && pl.base + m < randomMP) { js = j; break; }
          }
          if (js < np - 1) {
            idx_cut = 0;
            for (int j = 0; j <= js; j++) {
              const ScanPlan &pl = plans[(size_t)j];
              if (pl.self_idx >= 0) idx_cut = std::max(idx_cut, (uint32_t)pl.self_idx + 1u);
              for (int pi = 0; pi < pl.n_parts; pi++) idx_cut = std::max(idx_cut, pl.part_off[pi] + (uint32_t)pl.part_cnt[pi] + 1u);
            }
          }
        }
        const bool fused_sort = !u.exchange && n_ev >= 512;
        if (fused_sort) {
          // straight from the pinned copy: local column -> sample of the run, ordered by sample (first counting pass)
          ev_count.assign((size_t)u.B + 1, 0u);
          const UfbEvent *src = u.h_ev.p;
          uint32_t kept = 0;
          for (uint32_t k = 0; k < n_ev; k++)
            if (src[k].idx < idx_cut) { ev_count[(size_t)u.ids[(size_t)src[k].b] + 1]++; kept++; }
          events.resize(kept);
          ev_tmp.resize(kept);
          for (size_t k = 1; k <= (size_t)u.B; k++) ev_count[k] += ev_count[k - 1];
          for (uint32_t k = 0; k < n_ev; k++) {
            UfbEvent e = src[k];
            if (e.idx >= idx_cut) continue;
            e.b = (uint32_t)u.ids[(size_t)e.b];
            ev_tmp[ev_count[e.b]++] = e;
          }
          n_ev = kept;
          // ... then by scan output index (second pass, stable)
          ev_count.assign((size_t)n_idx + 1, 0u);
          for (const UfbEvent &e : ev_tmp) ev_count[(size_t)e.idx + 1]++;
          for (size_t k = 1; k <= (size_t)n_idx; k++) ev_count[k] += ev_count[k - 1];
          for (const UfbEvent &e : ev_tmp) events[ev_count[e.idx]++] = e;
        } else {
          events.clear();
          for (uint32_t k = 0; k < n_ev; k++) {
            UfbEvent ev = u.h_ev.p[k];
            if (ev.idx >= idx_cut) continue;
            ev.b = (uint32_t)u.ids[(size_t)ev.b];                                  // local column -> sample of the run
            events.push_back(ev);
          }
          n_ev = (uint32_t)events.size();
        }
        if (u.exchange) {
          // sample-sharded run: every rank replays the events of all ranks (one all-gather per batch)
          static_assert(sizeof(UfbEvent) == sizeof(mpf_ufb_event), "event layouts must match");
          const mpf_ufb_event *all = nullptr;
          uint32_t n_all_ev = 0;
          if (u.exchange(u.exchange_arg, exchange_tag++, reinterpret_cast<const mpf_ufb_event *>(events.data()), (uint32_t)events.size(), &all, &n_all_ev) != 0) {
            set_error("online UFBoot: event exchange failed (ranks out of step?)");
            return MPF_E_STATE;
          }
          const UfbEvent *pa = reinterpret_cast<const UfbEvent *>(all);
          events.assign(pa, pa + n_all_ev);
        }
        if (!fused_sort) sort_events(events, ev_tmp, ev_count, n_idx, (uint32_t)u.B);
        u.events += n_ev;
        t0 = now_ms();
        u.t_sort += t0 - t1;
      } else {
        events.clear();
      }
      t0 = now_ms();
      const double tr_t0 = t0;
      // (SelfMoot, see ufb_self_default: which samples hold the current topology already and tie with it -- boot_trees is this
      //  thread's here, and the log of the batch before has been worked off)
      moot_on = false;
      if (defer && host_self && self_pass && ran_events && u.ids_identity && !rand_fn_ && ufb_moot_ && u.log.empty() && !u.cut_btrees) {
        // (a sample may have taken the current topology during a ratchet climb, under another logl than the tree's own: an acceptance
        //  now would change boot_tree_orig_logl -- no shortcut while that array is kept, i.e. under -cutoff_from_btrees; without the
        //  option the reference does not keep it (iqtree.cpp:3717) and neither does this tracker)
        if (u.self_key_epoch != (uint64_t)topo_epoch_) { canonical_topology(back_, u.self_key); u.self_key_epoch = (uint64_t)topo_epoch_; }
        const auto it = u.topo_index.find(u.self_key);
        moot.flag.assign((size_t)u.Bl + 8, 0);
        moot.n_set = 0;
        moot.n_le = -1;
        if (it != u.topo_index.end()) {
          const int64_t cur_res = it->second;
          for (int c2 = 0; c2 < u.Bl; c2++)
            if (u.boot_trees[(size_t)c2] == cur_res && (uint32_t)u.h_rt.p[c2] == u.boot_score[(size_t)c2]) { moot.flag[(size_t)c2] = 1; moot.n_set++; }
        }
        if (moot.jump_n != u.Bl) {
          moot.jump_n = u.Bl;
          moot.jump_a = lcg64_skip(1, (uint64_t)u.Bl) - lcg64_skip(0, (uint64_t)u.Bl);     // A^Bl
          moot.jump_c = lcg64_skip(0, (uint64_t)u.Bl);
        }
        moot_on = true;
      }
      // ---- host replay in the reference's order
      size_t ep = 0;
      bool moved = false;
      int j = i;
      for (; j <= hi && !moved; j++) {
        const ScanPlan &pl = plans[(size_t)(j - i)];
        const int32_t cur_plan = (int32_t)(j - i);
        if (tie_mode_ == MPF_TIE_RANDOM) {
          insert_rec_ = remove_rec_ = -1;
          hits_ = 1;
        }
        long sel = -1;
        uint32_t sel_idx = 0, sel_home = 0;
        size_t c = 0;
        // the update rule of one booked tree (iqtree.cpp:3684-3731) over the samples whose events name output index idx
        // treels.find(tree_str) / treels[tree_str] = tree_index (iqtree.cpp:3500-3514, :3689-3707): a topology that some
        // sample accepted before keeps the index of that first tree
        auto topology_key = [&](uint32_t cand_code) -> const std::string & {
          if (cand_code == 0xFFFFFFFFu) {
            if (u.self_key_epoch != (uint64_t)topo_epoch_) { canonical_topology(back_, u.self_key); u.self_key_epoch = (uint64_t)topo_epoch_; }
            return u.self_key;
          }
          ufb_candidate_topology(cand_code < (uint32_t)pl.n_p ? pl.rec : back_[pl.rec], candidate_record(pl, (size_t)cand_code), mh_bk);
          canonical_topology(mh_bk, mh_key);
          return mh_key;
        };
        auto lookup_topology = [&](int64_t tree_index, uint32_t cand_code) -> int64_t {
          const double tl = now_ms();
          const int64_t ti = u.topo_index.emplace(topology_key(cand_code), tree_index).first->second;
          u.t_lookup += now_ms() - tl;
          u.lookups++;
          return ti;
        };
        // the update rule of one booked tree for one sample (b, score s): shared by the events the device extracted and by the
        // current tree's own bookings, which the host walks through itself
        const UfbDeferCtx dctx{defer, cur_plan, &log_open, moot_on ? &moot : nullptr};
        auto one_event = [&](const uint32_t b, const uint32_t s, int64_t &tree_index, bool &looked_up, const uint32_t cand_code) {
          ufb_one_event(b, s, tree_index, looked_up, cand_code, lookup_topology, dctx);      // (host/ufboot_common.hpp)
        };
        auto replay_events = [&](uint32_t idx, int64_t tree_index, uint32_t cand_code) {
          while (ep < events.size() && events[ep].idx < idx) ep++;
          bool looked_up = store_trees;              // (-storetrees: tree_str is set at the top, no lookup per sample)
          for (; ep < events.size() && events[ep].idx == idx; ep++) one_event(events[ep].b, events[ep].s, tree_index, looked_up, cand_code);
        };
        // the current tree scores R_T[b] for every sample: no device events for its slots (they would be B per prune-node visit,
        // 2.0e6 per move-less C3 sweep, all to be copied and ordered) -- the host has R_T and offers it to every sample in order
        auto replay_self = [&](int64_t tree_index) {
          if (defer) { ufb_self_default(u.h_rt.p, tree_index, cur_plan, log_open, u.draws, moot_on ? &moot : nullptr); return; }
          bool looked_up = store_trees;
          for (int c2 = 0; c2 < u.Bl; c2++) one_event((uint32_t)u.ids[(size_t)c2], (uint32_t)u.h_rt.p[c2], tree_index, looked_up, 0xFFFFFFFFu);
        };
        // one tree arriving at saveCurrentTree with length cur_len: its index in treels_logl, or -1 when nothing is booked.
        // Default: the cut-off test, then a new index (iqtree.cpp:3343-3348).  -storetrees: looked up by topology first
        // (:3302-3341); one met before is skipped unless the length improved on the recorded one, and then it goes on under
        // its old index without the cut-off test.
        auto book_tree = [&](uint32_t cur_len, bool passes_cut, uint32_t cand_code) -> int64_t {
          return ufb_book_tree(cur_len, passes_cut, cand_code, store_trees, topology_key);
        };
        if (pl.self_idx >= 0) {
          // rearrangeParsimony's evaluateParsimony(p) + pllSaveCurrentTreeSprParsimony (sprparsimony.cpp:2285-2289): the
          // current tree, length randomMP, once per prune node and before any of its insertion tests
          bool pass;
          if (!ratchet) pass = !none_pass && randomMP <= mp_max;
          else {
            pass = (store_trees || !u.gate_closed) && ran_events && !none_pass && u.stale_len <= mp_max;
            if (!pass) u.gate_closed = true;
          }
          u.cur_logl_now = -(int32_t)(ratchet ? u.stale_len : randomMP);
          const int64_t tree_index = book_tree(ratchet ? u.stale_len : randomMP, pass, 0xFFFFFFFFu);
          if (tree_index >= 0) {
            if (host_self) replay_self(tree_index); else replay_events((uint32_t)pl.self_idx, tree_index, 0xFFFFFFFFu);
            if (ratchet) u.stale_len = u.rt_orig;        // _pattern_pars now holds the current tree
          }
        }
        for (int pi = 0; pi < pl.n_parts; pi++) {
          const uint32_t home = pl.part_off[pi] + (uint32_t)pl.part_cnt[pi];
          for (int k = 0; k < pl.part_cnt[pi]; k++, c++) {
            if (have_cut && !ratchet && !store_trees) {
              // under a cut-off most insertion tests change nothing: not booked (above the cut-off) and longer than the best tree so
              // far (no counter, no draw) -- on to the next one that is either (best_ only falls while the block is read)
              const uint32_t lim_len = std::max(none_pass ? 0u : mp_max, best_);
              const int k2 = lim_len >= pl.base ? first_le(out + pl.part_off[pi], k, pl.part_cnt[pi], lim_len - pl.base) : pl.part_cnt[pi];
              c += (size_t)(k2 - k);
              k = k2;
              if (k >= pl.part_cnt[pi]) break;
            }
            const uint32_t idx = pl.part_off[pi] + (uint32_t)k;
            const uint32_t mp = pl.base + out[idx];
            // saveCurrentTree(-mp) (reference sprparsimony.cpp:2163-2166), before the SPR tie rule
            bool pass;
            if (!ratchet) pass = !none_pass && mp <= mp_max;
            else {
              // iqtree.cpp:3283-3295 then :3343: the filter sees the length booked last; a tree that fails leaves
              // _pattern_pars as it is, so every later candidate of the climb fails too (-storetrees: unless a known
              // topology gets in past the cut-off)
              pass = (store_trees || !u.gate_closed) && have_C && !none_pass && u.stale_len <= mp_max;     // (have_C: a ratchet batch with candidates always has its product)
              if (!pass) u.gate_closed = true;
            }
            u.cur_logl_now = -(int32_t)(ratchet ? u.stale_len : mp);
            const int64_t tree_index = book_tree(ratchet ? u.stale_len : mp, pass, (uint32_t)c);          // iqtree.cpp:3302-3348
            if (tree_index >= 0) {
              replay_events(idx, tree_index, (uint32_t)c);
              // pllComputePatternParsimony (:3365) has now refreshed _pattern_pars for THIS candidate: its length on the
              // original alignment is what the next call will see
              if (ratchet) u.stale_len = (uint32_t)((int64_t)u.rt_orig - (int64_t)lcol[(size_t)home] + (int64_t)lcol[(size_t)hinfo[idx].x]);
            }
            // testInsertParsimony's tie rule (reference :2168-2176 / fastDNAparsimony.c:1224-1229)
            if (tie_mode_ == MPF_TIE_RANDOM) {
              if (mp < best_) hits_ = 1;
              else if (mp == best_) hits_++;
              if (mp < best_ || (mp == best_ && tie_draw() <= 1.0 / (double)hits_)) { best_ = mp; sel = (long)c; sel_idx = idx; sel_home = home; }
            } else if (mp < best_) {
              best_ = mp; sel = (long)c; sel_idx = idx; sel_home = home;
            }
          }
        }
        if (sel >= 0) {
          insert_rec_ = candidate_record(pl, (size_t)sel);
          remove_rec_ = sel < pl.n_p ? pl.rec : back_[pl.rec];
        }
        if (!defer) ufb_flush_pending(pl);
        else if (log_open) { u.log.push_back(UfbState::LogEntry{0xFFFFFFFFu, 0u, 0, cur_plan}); log_open = false; }
        bool accept;
        if (tie_mode_ == MPF_TIE_RANDOM) {
          if (best_ == randomMP) iter_hits++;
          if (best_ < randomMP) iter_hits = 1;
          accept = (best_ < randomMP || (best_ == randomMP && tie_draw() <= 1.0 / (double)iter_hits)) &&
                   remove_rec_ >= 0 && insert_rec_ >= 0;
        } else {
          accept = best_ < randomMP;
        }
        if (accept) {
          if (sel < 0) { set_error("online UFBoot: accepted move without a candidate of this prune node"); return MPF_E_STATE; }
          // the accepted candidate becomes the current tree: R_T += C[cand] - C[home]
          {
            uint32_t rc_ = 0xFFFFFFFFu, rh_ = 0xFFFFFFFFu;
            if (have_C && sel_idx < n_idx) {
              if (compact) { rc_ = crow[sel_idx]; rh_ = crow[sel_home]; }
              else { rc_ = hinfo[sel_idx].x; rh_ = sel_home; }
            }
            if (rc_ != 0xFFFFFFFFu && rh_ != 0xFFFFFFFFu) {
              UCHK(launch_rt_update(st_, u.rt.p, u.C.p, u.Bp, rc_, rh_));
            } else {
              // its rows were not part of the product (outside the saved set): multiply just these two mask rows
              UCHK(u.sel2.reserve((size_t)kUfbRowTile));
              UCHK(u.C2.reserve((size_t)kUfbRowTile * (size_t)u.Bp));
              UCHK(u.h_small.reserve((size_t)kUfbRowTile + 4));
              for (int z = 0; z < kUfbRowTile; z++) u.h_small.p[z] = sel_home;
              u.h_small.p[0] = hinfo[sel_idx].x;
              UCHK(hipMemcpyAsync(u.sel2.p, u.h_small.p, (size_t)kUfbRowTile * sizeof(uint32_t), hipMemcpyHostToDevice, st_));
              for (int pl2 = 0; pl2 < u.planes; pl2++)
                UCHK(launch_bitgemm(st_, u.masks.p, kUfbRowTile, g_.Wp, u.wt.p + (size_t)pl2 * u.plane_bytes, u.Bp, u.C2.p, 1 << (7 * pl2), pl2 > 0, u.sel2.p));
              UCHK(launch_rt_update(st_, u.rt.p, u.C2.p, u.Bp, 0u, 1u));
              UCHK(hipStreamSynchronize(st_));       // h_small is reused by the next batch
            }
          }
          if (ratchet) { int rc2 = read_rt_orig(); if (rc2) return rc2; }
          moves_.push_back(Move{remove_rec_, insert_rec_, best_});
          if (!u.log.empty()) { u.log_back = back_; u.log_epoch = topo_epoch_; u.log_plans = &plans; }   // the tree the log speaks of
          apply_move(remove_rec_, insert_rec_);
          randomMP = best_;
          moved = true;
        }
      }
      if (!moved && !u.log.empty()) { u.log_back = back_; u.log_epoch = topo_epoch_; u.log_plans = &plans; }
      if (ufb_trace_env())
        std::fprintf(stderr, "[ufb-batch] i %d np %d used %d n_idx %u rows %u events %u moved %d len %u mp_max %u | last scan %.3f dev %.3f replay %.3f ms\n", i, np, j - i, n_idx,
                     n_rows, n_ev, (int)moved, randomMP, mp_max, u.t_scan - tr_scan0, u.t_dev - tr_dev0, now_ms() - tr_t0);
      batch = std::max(next_batch(batch, moved, j - i, total), batch_floor);
      visits_done_ += j - i;
      i = j;
      sw_events += n_ev;
      sw_moved = sw_moved || moved;
      sw_mp_max = mp_max;
      u.t_replay += now_ms() - t0;
    }
    if (sw_full && !sw_moved && sw_events == 0 && i > total && defer && host_self && ufb_memo_) {
      if (u.self_key_epoch != (uint64_t)topo_epoch_) { canonical_topology(back_, u.self_key); u.self_key_epoch = (uint64_t)topo_epoch_; }
      uint32_t &v = u.quiet_topo[u.quiet_key(mintrav, maxtrav, n_)];
      v = std::max(v, sw_mp_max);
    }
  } while (randomMP < startMP && !visits_out());
  ufb_drain_log();
  climb_finished(total);
  if (u.exchange) {
    // closing handshake: a rank that took another path would be in the middle of a batch here
    const mpf_ufb_event *all = nullptr;
    uint32_t n_all_ev = 0;
    if (u.exchange(u.exchange_arg, 0xFFFFFFFFu, nullptr, 0, &all, &n_all_ev) != 0) { set_error("online UFBoot: ranks out of step at the end of the climb"); return MPF_E_STATE; }
  }
  if (final_score) *final_score = randomMP;
  abort_guard.ok = true;
  return MPF_OK;
}

}  // namespace mpf
