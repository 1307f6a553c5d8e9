// ufboot_refine.cpp -- batched bootstrap refinement (mpf_ufboot_refine_sweep) on both engines; the tie stream's skip-ahead
#include "ufboot_common.hpp"

namespace mpf {

// ---- batched bootstrap refinement ------------------------------------------------------------------------------------------
// IQTree::optimizeBootTrees' default branch (iqtree.cpp:2797-2862) takes the samples one at a time: re-weight the alignment with
// boot_samples_pars[b] (modifyPatternFreq :2520), rebuild the parsimony structures (on_opt_btree, sprparsimony.cpp:3253), read the
// sample's tree and run ONE pllOptimizeSprParsimony from it (:2837).  The online phase leaves most samples on few distinct trees
// (one, where the data are decisive), and a tree that is already SPR-optimal under a sample's weights comes back unchanged after
// one sweep without an accepted move.  Fitch state sets do not depend on the weights -- only the counts do -- so that first sweep
// is computed for ALL samples that share the current tree at once: one masked scan (the 1-bit "no common state" mask of every
// insertion test), one binary x int8 product against the samples' weights on the matrix cores (length of candidate c under
// sample b = R_T[b] - C[home][b] + C[c][b]), and the extraction of the (candidate, sample) pairs whose length reaches the
// sample's running best (k_ufb_events with the bound R_T[b]) -- the only pairs testInsertParsimony's rule (:2168-2176) can act on.
// Each sample's sweep is then replayed on the host from ITS events with ITS tie stream: hits / draws inside a prune node, the
// sweep's accept rule with bestIterationScoreHits behind every prune node (:3306-3311; one draw per visit while nothing is
// better, skipped ahead in closed form over visits without events).  stable[b] = the sweep accepts no move: the climb's result
// is the current tree with length scores[b], exactly what the solo call returns.  A sample whose sweep does accept a move
// (an improvement, or a drawn move to an equally long tree) is NOT advanced here: the caller runs its climb alone.
// state * A^k + c * (A^k - 1) / (A - 1) mod 2^64 by doubling: k draws of the tie stream at once
uint64_t lcg64_skip(uint64_t state, uint64_t k)
{
  uint64_t a = 0x27bb2ee687b0b0fdULL, c = 3037000493ULL, acc_a = 1, acc_c = 0;
  while (k) {
    if (k & 1) { acc_a *= a; acc_c = acc_c * a + c; }
    c = (a + 1) * c;
    a *= a;
    k >>= 1;
  }
  return acc_a * state + acc_c;
}

int Engine::ufboot_refine_sweep(int maxtrav, const int32_t *tie_seeds, uint32_t *scores, uint8_t *stable, int32_t *first_move_visit)
{
  if (!ufb_) { set_error("refine sweep: no samples attached (mpf_ufboot_attach)"); return MPF_E_STATE; }
  UfbState &u = *ufb_;
  if (!have_tree_ || ntips_ != n_) { set_error("refine sweep: no complete tree set"); return MPF_E_STATE; }
  if (u.suspended || u.ratchet) { set_error("refine sweep: the attach-time weights must be in force"); return MPF_E_STATE; }
  if (tie_mode_ != MPF_TIE_RANDOM) { set_error("refine sweep: mpboot's random tie rule only (MPF_TIE_RANDOM)"); return MPF_E_UNSUPPORTED; }
  const int total = 2 * n_ - 2;
  struct Ev { uint32_t col, visit, ord, s; };
  std::vector<Ev> all;
  std::vector<uint32_t> rt((size_t)u.Bl);
  // every sample's sweep, replayed from its own events (both engines)
  auto replay = [&]() {
    std::sort(all.begin(), all.end(), [](const Ev &x, const Ev &y) { return x.col != y.col ? x.col < y.col : x.visit != y.visit ? x.visit < y.visit : x.ord < y.ord; });
    size_t ep = 0;
    for (int c = 0; c < u.Bl; c++) {
      const int b = u.ids[(size_t)c];
      const uint32_t randomMP = rt[(size_t)c];
      uint32_t best = randomMP;
      TieRng rng;
      rng.seed(tie_seeds ? tie_seeds[b] : b);
      uint64_t iter_hits = 1;
      uint32_t done_visits = 0;                             // visits [0, done_visits) of the sweep are behind us
      bool moved = false;
      uint32_t move_visit = 0;
      while (ep < all.size() && all[ep].col == (uint32_t)c && !moved) {
        const uint32_t v = all[ep].visit;
        // visits without an event: bestParsimony == randomMP, one draw each, nothing selected (:3306-3311)
        const uint64_t k = (uint64_t)(v - done_visits);
        rng.state = lcg64_skip(rng.state, k);
        iter_hits += k;
        // this visit: testInsertParsimony's rule over its events (:2168-2176), bestTreeScoreHits = 1 at its start
        uint64_t hits = 1;
        bool sel = false;
        for (; ep < all.size() && all[ep].col == (uint32_t)c && all[ep].visit == v; ep++) {
          const uint32_t mp = all[ep].s;
          if (mp > best) continue;
          if (mp < best) hits = 1;
          else hits++;
          bool take = mp < best;
          if (!take) take = rng.next() <= 1.0 / (double)hits;
          if (take) { best = mp; sel = true; }
        }
        bool accept = best < randomMP;
        if (!accept) {                                      // best == randomMP
          iter_hits++;
          accept = rng.next() <= 1.0 / (double)iter_hits;
        }
        if (accept && sel) { moved = true; move_visit = v; }
        done_visits = v + 1u;
      }
      while (ep < all.size() && all[ep].col == (uint32_t)c) ep++;
      if (scores) scores[b] = randomMP;
      if (stable) stable[b] = moved ? 0 : 1;
      if (first_move_visit) first_move_visit[b] = moved ? (int32_t)move_visit + 1 : 0;     // 1-based index into nodep[], 0 = none
    }
  };
  if (u.snk) {
    // ---- the weighted (-cost) engine: the scans write every tentative tree's per-pattern lengths (k_snk_scan), bit planes of them
    // times the sample weights on the matrix cores = every sample's length of every tentative tree; the current tree's row is
    // multiplied along and is the samples' bound (the evaluate of :3277 under sample b)
    node_rectifier();
    const uint32_t npat = (uint32_t)g_.Wp;
    std::vector<ScanPlan> plans;
    const uint32_t *out = nullptr;
    std::vector<uint2> hinfo;
    std::vector<uint32_t> small;
    // (a row is one tentative tree's per-pattern lengths, 2 bytes per pattern, ~60 rows per prune node at radius 6: chunks of
    //  at most ~2 GB of rows -- plus their bit planes -- whatever the option says)
    const int chunk = std::max(1, std::min(refine_chunk_, (int)std::max<uint64_t>(1, 2000000000ull / ((uint64_t)npat * 2ull * 64ull))));
    bool have_rt = false;
    for (int i = 1; i <= total; i += chunk) {
      const int hi = std::min(total, i + chunk - 1), np = hi - i + 1;
      UCHK(u.vmax.reserve(4));
      UCHK(hipMemsetAsync(u.vmax.p, 0, sizeof(uint32_t), st_));
      scan_vals_ = true;
      int rc = scan_batch(plans, nodep_.data() + i, np, 1, maxtrav, &out);
      scan_vals_ = false;
      if (rc) return rc;
      const uint32_t n_idx = vals_rows_, R = n_idx;
      UCHK(u.vals.reserve(((size_t)n_idx + 1) * npat));
      UCHK(u.h_vmax.reserve(4));
      if (asym_) UCHK(launch_sankoff_pattern(st_, g_, d_vec_, slot(back_[start_]), slot(start_), u.vals.p + (size_t)R * npat, u.vmax.p));
      else UCHK(launch_sankoff_pattern(st_, g_, d_vec_, slot(start_), slot(back_[start_]), u.vals.p + (size_t)R * npat, u.vmax.p));
      UCHK(hipMemcpyAsync(u.h_vmax.p, u.vmax.p, sizeof(uint32_t), hipMemcpyDeviceToHost, st_));
      UCHK(hipStreamSynchronize(st_));
      int K = 1;
      while (K < 16 && (u.h_vmax.p[0] >> K)) K++;
      const uint32_t rows = n_idx + 1;
      const int rows_p = round_up((int)rows, kUfbRowTile);
      const size_t plane_words = (size_t)rows_p * (size_t)u.Wp_s;
      UCHK(u.bitp.reserve((size_t)K * plane_words));
      UCHK(hipMemsetAsync(u.bitp.p, 0, (size_t)K * plane_words * sizeof(uint32_t), st_));
      for (uint32_t r0 = 0; r0 < rows; r0 += 32768u)
        UCHK(launch_vals_planes(st_, u.vals.p + (size_t)r0 * npat, std::min(32768u, rows - r0), npat, K, u.bitp.p + (size_t)r0 * u.Wp_s, (uint32_t)rows_p,
                                (uint32_t)u.Wp_s));
      UCHK(u.C.reserve((size_t)rows_p * (size_t)u.Bp));
      bool first = true;
      for (int k = 0; k < K; k++)
        for (int pl = 0; pl < u.planes; pl++) {
          UCHK(launch_bitgemm(st_, u.bitp.p + (size_t)k * plane_words, rows_p, u.Wp_s, u.wt.p + (size_t)pl * u.plane_bytes, u.Bp, u.C.p,
                              (1 << k) << (7 * pl), first ? 0 : 1));
          first = false;
        }
      u.gemm_rows += (uint64_t)rows_p * (uint64_t)K;
      UCHK(u.rt.reserve((size_t)u.Bp));
      UCHK(launch_colsum(st_, u.C.p + (size_t)R * u.Bp, 1, u.Bp, u.rt.p));
      if (!have_rt) {
        UCHK(u.h_rt.reserve((size_t)u.Bp));
        UCHK(hipMemcpyAsync(u.h_rt.p, u.rt.p, (size_t)u.Bl * sizeof(int32_t), hipMemcpyDeviceToHost, st_));
        UCHK(hipStreamSynchronize(st_));
        for (int c = 0; c < u.Bl; c++) rt[(size_t)c] = (uint32_t)u.h_rt.p[c];
        have_rt = true;
      }
      if (n_idx == 0) continue;
      // per output index: (row, prune node of the chunk) for the candidates; the current tree's slots take no part
      hinfo.assign((size_t)n_idx, make_uint2(0u, 0xFFFFFFFFu));
      std::vector<uint32_t> ends((size_t)np);
      uint32_t run = 0;
      for (int j = 0; j < np; j++) {
        const ScanPlan &pl = plans[(size_t)j];
        if (pl.self_idx >= 0) run = std::max(run, (uint32_t)pl.self_idx + 1u);
        for (const Candidate &cd : pl.cands) { hinfo[cd.out] = make_uint2(cd.out, (uint32_t)j); run = std::max(run, cd.out + 1u); }
        ends[(size_t)j] = run;
      }
      const uint32_t n_parts = (uint32_t)np;
      const size_t o_cnt = (size_t)2 * n_parts;
      small.assign(o_cnt + 1, 0u);
      for (uint32_t d = 0; d < n_parts; d++) { small[d] = UINT32_MAX; small[n_parts + d] = R; }
      UCHK(u.h_small.reserve(small.size() + 4));
      std::memcpy(u.h_small.p, small.data(), small.size() * sizeof(uint32_t));
      UCHK(u.thr.reserve(small.size() + 4));
      UCHK(u.info.reserve((size_t)n_idx));
      const uint32_t nch = ufb_chunks(n_idx);
      UCHK(u.cmin.reserve((size_t)nch * (size_t)u.Bp));
      UCHK(u.pre.reserve((size_t)nch * (size_t)u.Bp));
      if (u.ev.cap == 0) { UCHK(u.ev.reserve(1u << 18)); UCHK(u.h_ev.reserve(1u << 18)); }
      UCHK(hipMemcpyAsync(u.thr.p, u.h_small.p, small.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st_));
      UCHK(hipMemcpyAsync(u.info.p, hinfo.data(), (size_t)n_idx * sizeof(uint2), hipMemcpyHostToDevice, st_));
      uint32_t *d_evcount = u.thr.p + o_cnt;
      uint32_t n_ev = 0;
      for (bool again = false;; again = true) {
        if (again) UCHK(hipMemsetAsync(d_evcount, 0, sizeof(uint32_t), st_));
        UCHK(launch_ufb_events(st_, u.info.p, d_out(), u.thr.p, u.thr.p + n_parts, nullptr, u.C.p, u.Bp, u.Bl, u.rt.p,
                               reinterpret_cast<const uint32_t *>(u.rt.p), n_idx, u.cmin.p, u.pre.p, u.ev.p, (uint32_t)u.ev.cap, d_evcount, 0));
        UCHK(hipMemcpyAsync(u.h_small.p, d_evcount, sizeof(uint32_t), hipMemcpyDeviceToHost, st_));
        UCHK(hipStreamSynchronize(st_));
        n_ev = u.h_small.p[0];
        if (n_ev <= u.ev.cap) break;
        UCHK(u.ev.reserve((size_t)n_ev));
        UCHK(u.h_ev.reserve((size_t)n_ev));
      }
      if (n_ev) {
        UCHK(u.h_ev.reserve((size_t)n_ev));
        UCHK(hipMemcpyAsync(u.h_ev.p, u.ev.p, (size_t)n_ev * sizeof(UfbEvent), hipMemcpyDeviceToHost, st_));
        UCHK(hipStreamSynchronize(st_));
        for (uint32_t k = 0; k < n_ev; k++) {
          const UfbEvent &e = u.h_ev.p[k];
          const uint32_t j = (uint32_t)(std::upper_bound(ends.begin(), ends.end(), e.idx) - ends.begin());
          all.push_back(Ev{e.b, (uint32_t)(i - 1) + j, e.idx, e.s});
        }
      }
    }
    replay();
    u.rt_valid = false;
    return MPF_OK;
  }
  if (scan_mode_ != 1) { set_error("refine sweep needs the device-walked scan (option scan_mode 1)"); return MPF_E_UNSUPPORTED; }
  node_rectifier();
  { int rc = ufb_current_tree_reps(); if (rc) return rc; }          // R_T[b]: what the evaluate of sprparsimony.cpp:3277 returns under sample b
  UCHK(u.h_rt.reserve((size_t)u.Bp));
  UCHK(hipMemcpyAsync(u.h_rt.p, u.rt.p, (size_t)u.Bl * sizeof(int32_t), hipMemcpyDeviceToHost, st_));
  UCHK(hipStreamSynchronize(st_));
  for (int c = 0; c < u.Bl; c++) rt[(size_t)c] = (uint32_t)u.h_rt.p[c];
  // ---- the whole sweep's insertion tests, with masks (chunks of prune nodes bound the mask and product buffers)
  std::vector<ScanPlan> plans;
  const uint32_t *out = nullptr;
  std::vector<uint32_t> small;
  const int chunk = std::max(1, refine_chunk_);
  for (int i = 1; i <= total; i += chunk) {
    const int hi = std::min(total, i + chunk - 1), np = hi - i + 1;
    scan_masks_ = true;
    int rc = scan_batch(plans, nodep_.data() + i, np, 1, maxtrav, &out);
    scan_masks_ = false;
    if (rc) return rc;
    uint32_t n_idx = 0, n_parts = 0;
    std::vector<uint32_t> self_list;
    for (int j = 0; j < np; j++) {
      const ScanPlan &pl = plans[(size_t)j];
      if (pl.self_idx >= 0) { self_list.push_back((uint32_t)pl.self_idx); n_idx = std::max(n_idx, (uint32_t)pl.self_idx + 1u); }
      for (int pi = 0; pi < pl.n_parts; pi++) {
        n_idx = std::max(n_idx, pl.part_off[pi] + (uint32_t)pl.part_cnt[pi] + 1u);
        n_parts = std::max(n_parts, (uint32_t)pl.part_desc[pi] + 1u);
      }
    }
    if (n_parts == 0) continue;                         // (no insertion test in this chunk)
    const int rows_p = round_up((int)std::max<uint32_t>(n_idx, 1u), kUfbRowTile);
    { int rc2 = ufb_reserve_scan(n_idx); if (rc2) return rc2; }
    // staging: thr[n_parts] | home[n_parts] | self[n_self] | event count
    const size_t o_self = (size_t)2 * n_parts, o_cnt = o_self + self_list.size();
    small.assign(o_cnt + 1, 0u);
    std::copy(self_list.begin(), self_list.end(), small.begin() + (long)o_self);
    for (int j = 0; j < np; j++) {
      const ScanPlan &pl = plans[(size_t)j];
      for (int pi = 0; pi < pl.n_parts; pi++) {
        const uint32_t d = (uint32_t)pl.part_desc[pi];
        small[d] = UINT32_MAX;                          // no cut-off: every insertion test counts
        small[n_parts + d] = pl.part_off[pi] + (uint32_t)pl.part_cnt[pi];
      }
    }
    UCHK(u.h_small.reserve(small.size() + 4));
    std::memcpy(u.h_small.p, small.data(), small.size() * sizeof(uint32_t));
    UCHK(u.thr.reserve(small.size() + 4));
    UCHK(u.C.reserve((size_t)rows_p * (size_t)u.Bp));
    const uint32_t nch = ufb_chunks(n_idx);
    UCHK(u.cmin.reserve((size_t)nch * (size_t)u.Bp));
    UCHK(u.pre.reserve((size_t)nch * (size_t)u.Bp));
    if (u.ev.cap == 0) { UCHK(u.ev.reserve(1u << 18)); UCHK(u.h_ev.reserve(1u << 18)); }
    UCHK(hipMemcpyAsync(u.thr.p, u.h_small.p, small.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st_));
    // (the current tree's own slots take no part: its length under sample b IS the bound)
    UCHK(launch_ufb_self(st_, u.info.p, u.thr.p + o_self, (uint32_t)self_list.size(), 0xFFFFFFFFu));
    if (timing_) UCHK(hipEventRecord(ev2_, st_));
    for (int pl = 0; pl < u.planes; pl++)
      UCHK(launch_bitgemm(st_, u.masks.p, rows_p, g_.Wp, u.wt.p + (size_t)pl * u.plane_bytes, u.Bp, u.C.p, 1 << (7 * pl), pl > 0, nullptr));
    u.gemm_rows += (uint64_t)rows_p;
    if (timing_) UCHK(hipEventRecord(ev3_, st_));
    uint32_t *d_evcount = u.thr.p + o_cnt;
    uint32_t n_ev = 0;
    for (bool again = false;; again = true) {
      if (again) UCHK(hipMemsetAsync(d_evcount, 0, sizeof(uint32_t), st_));
      // the bound of sample b: the current tree's length under its weights, R_T[b] -- a running minimum from there on
      UCHK(launch_ufb_events(st_, u.info.p, d_out(), u.thr.p, u.thr.p + n_parts, nullptr, u.C.p, u.Bp, u.Bl, u.rt.p,
                             reinterpret_cast<const uint32_t *>(u.rt.p), n_idx, u.cmin.p, u.pre.p, u.ev.p, (uint32_t)u.ev.cap, d_evcount, 0));
      UCHK(hipMemcpyAsync(u.h_small.p, d_evcount, sizeof(uint32_t), hipMemcpyDeviceToHost, st_));
      UCHK(hipStreamSynchronize(st_));
      n_ev = u.h_small.p[0];
      if (n_ev <= u.ev.cap) break;
      UCHK(u.ev.reserve((size_t)n_ev));
      UCHK(u.h_ev.reserve((size_t)n_ev));
    }
    if (timing_) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, ev2_, ev3_) == hipSuccess) u.gemm_ms += ms;
    }
    if (n_ev) {
      UCHK(u.h_ev.reserve((size_t)n_ev));
      UCHK(hipMemcpyAsync(u.h_ev.p, u.ev.p, (size_t)n_ev * sizeof(UfbEvent), hipMemcpyDeviceToHost, st_));
      UCHK(hipStreamSynchronize(st_));
      // output index -> (visit of the sweep, order inside the visit): the indices of one prune node's parts rise in the
      // reference's candidate order
      std::vector<uint32_t> ends((size_t)np);            // end (exclusive) of prune node j's index range
      uint32_t run = 0;
      for (int j = 0; j < np; j++) {
        const ScanPlan &pl = plans[(size_t)j];
        if (pl.self_idx >= 0) run = std::max(run, (uint32_t)pl.self_idx + 1u);
        for (int pi = 0; pi < pl.n_parts; pi++) run = std::max(run, pl.part_off[pi] + (uint32_t)pl.part_cnt[pi] + 1u);
        ends[(size_t)j] = run;
      }
      for (uint32_t k = 0; k < n_ev; k++) {
        const UfbEvent &e = u.h_ev.p[k];
        const uint32_t j = (uint32_t)(std::upper_bound(ends.begin(), ends.end(), e.idx) - ends.begin());
        all.push_back(Ev{e.b, (uint32_t)(i - 1) + j, e.idx, e.s});
      }
    }
  }
  replay();
  u.rt_valid = true;
  return MPF_OK;
}

}  // namespace mpf
