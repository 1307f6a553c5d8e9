// ufboot_snk.cpp -- the tracked climb on the weighted (Sankoff, -cost) engine
#include "ufboot_common.hpp"

namespace mpf {

// The same loop on the weighted (Sankoff) engine.  pllComputePatternParsimony dispatches to
// pllComputeSankoffPatternParsimony there (sprparsimony.cpp:3341-3355): the per-pattern lengths of the tentative tree are the
// minima the evaluate has just taken.  The scan writes them for every insertion test (k_snk_scan, 16 bits each), k_vals_planes
// slices them into bit planes and REPS = sum_k 2^k (plane k x weights) on the matrix cores; the current tree's own row is
// multiplied along with every batch and serves as the "home" row of the event formula, so that a candidate's score is its own
// product row and the current tree's slots score R_T.  Scans are host-planned (as every weighted scan), everything else --
// cut-off filter, ratchet rule, update rules, replay order -- is the code path of spr_sweeps_ufboot.
int Engine::spr_sweeps_ufboot_snk(int mintrav, int maxtrav, uint32_t randomMP, uint32_t *final_score)
{
  UfbState &u = *ufb_;
  if (u.exchange) {
    // sample-sharded run: every rank must cut the climb into the same batches -- start from a fixed batch policy state
    gap_est_ = -1.0;
    since_move_ = 0;
  }
  uint32_t exchange_tag = 0;
  uint32_t startMP;
  unsigned iter_hits = 1;
  const int total = 2 * n_ - 2;
  const uint32_t npat = (uint32_t)g_.Wp;
  std::vector<ScanPlan> plans;
  const uint32_t *out = nullptr;
  int batch = first_batch();
  std::vector<UfbEvent> events, ev_tmp;
  std::vector<uint32_t> ev_count, small;
  std::vector<uint2> hinfo;
  std::vector<int32_t> mh_bk, lcol;
  std::string mh_key;
  const bool ratchet = u.ratchet;
  const bool store_trees = u.store_trees;          // -storetrees (iqtree.cpp:3302-3346)
  // (an asymmetric matrix gives the CURRENT tree another length and another row at every prune node's visit -- it is evaluated at
  //  that node's edge, evaluateParsimony(p) of :2285 --: the scan writes both to the visit's slot, scan_batch)
  const bool asym = asym_;
  const bool host_self = !u.exchange && !asym;     // (sample-sharded: R_T lives in pieces on the ranks, the current tree's bookings come as device events)
  const int oc = u.Bl;
  if (ratchet) u.gate_closed = false;
  bool stale_init = false;                         // ratchet: _pattern_pars of the climb's start tree, known after the first product
  do {
    startMP = randomMP;
    node_rectifier();
    int i = 1;
    while (i <= total) {
      const int hi = std::min(total, i + batch - 1);
      const int np = hi - i + 1;
      UCHK(u.vmax.reserve(4));
      UCHK(hipMemsetAsync(u.vmax.p, 0, sizeof(uint32_t), st_));      // (a batch without any insertion test launches no scan)
      scan_vals_ = true;
      int rc = scan_batch(plans, nodep_.data() + i, np, mintrav, maxtrav, &out);
      scan_vals_ = false;
      if (rc) return rc;
      u.batches++;
      const uint32_t n_idx = vals_rows_;           // output indices: a slot for the current tree in front of every prune node's candidates
      const uint32_t R = n_idx;                    // the current tree's row of the product
      int jstar = np - 1;
      for (int j = 0; j < np; j++) {
        const ScanPlan &pl = plans[(size_t)j];
        uint32_t m = UINT32_MAX;
        for (const Candidate &cd : pl.cands) m = std::min(m, out[cd.out]);
        if (m < randomMP) { jstar = j; break; }
      }
      const bool have_cut = u.logl_cutoff != 0.0;
      const double lim = -u.logl_cutoff + 1e-4;
      const uint32_t mp_max = have_cut ? (lim <= 0.0 ? 0u : (uint32_t)std::ceil(lim) - 1u) : UINT32_MAX;
      const bool none_pass = have_cut && lim <= 0.0;
      const bool skip_product = store_trees ? false : ratchet ? (u.gate_closed || none_pass) : none_pass;
      bool have_C = false;
      events.clear();
      if (!skip_product) {
        // ---- the current tree's row, the bit planes, the product
        UCHK(u.vals.reserve(((size_t)n_idx + 1) * npat));
        UCHK(u.h_vmax.reserve(4));
        if (asym) UCHK(launch_sankoff_pattern(st_, g_, d_vec_, slot(back_[start_]), slot(start_), u.vals.p + (size_t)R * npat, u.vmax.p));     // (left = far end, as tree_length)
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
        for (uint32_t r0 = 0; r0 < rows; r0 += 32768u)                  // (grid.y limit)
          UCHK(launch_vals_planes(st_, u.vals.p + (size_t)r0 * npat, std::min(32768u, rows - r0), npat, K, u.bitp.p + (size_t)r0 * u.Wp_s, (uint32_t)rows_p,
                                  (uint32_t)u.Wp_s));
        UCHK(u.C.reserve((size_t)rows_p * (size_t)u.Bp));
        if (timing_) UCHK(hipEventRecord(ev2_, st_));
        bool first = true;
        for (int k = 0; k < K; k++)
          for (int pl = 0; pl < u.planes; pl++) {
            UCHK(launch_bitgemm(st_, u.bitp.p + (size_t)k * plane_words, rows_p, u.Wp_s, u.wt.p + (size_t)pl * u.plane_bytes, u.Bp, u.C.p,
                                (1 << k) << (7 * pl), first ? 0 : 1));
            first = false;
          }
        if (timing_) UCHK(hipEventRecord(ev3_, st_));
        u.gemm_rows += (uint64_t)rows_p * (uint64_t)K;
        UCHK(u.rt.reserve((size_t)u.Bp));
        UCHK(launch_colsum(st_, u.C.p + (size_t)R * u.Bp, 1, u.Bp, u.rt.p));       // R_T = the current tree's own row
        UCHK(u.h_rt.reserve((size_t)u.Bp));
        UCHK(hipMemcpyAsync(u.h_rt.p, u.rt.p, (size_t)u.Bl * sizeof(int32_t), hipMemcpyDeviceToHost, st_));    // synchronised below
        have_C = true;
        // ---- per output index: (row, part) for candidates of plans [0, jstar], the current tree's slots, everything else off
        uint32_t n_parts = 0;
        hinfo.assign((size_t)n_idx, make_uint2(0u, 0xFFFFFFFFu));
        const bool self_pass = ratchet || store_trees || randomMP <= mp_max;
        for (int j = 0; j <= jstar; j++) {
          const ScanPlan &pl = plans[(size_t)j];
          // (asym: the visit's slot is a row of its own -- it takes part like a candidate, its cost the length at that edge)
          if (pl.self_idx >= 0) hinfo[(size_t)pl.self_idx] = asym ? make_uint2((uint32_t)pl.self_idx, (uint32_t)j)
                                                                   : make_uint2(0u, (self_pass && !host_self) ? 0xFFFFFFFEu : 0xFFFFFFFFu);
          for (const Candidate &cd : pl.cands) hinfo[cd.out] = make_uint2(cd.out, (uint32_t)j);
          n_parts = (uint32_t)j + 1u;
        }
        // staging: thr[n_parts] | home[n_parts] | best[Bp]
        const size_t o_cnt = (size_t)2 * n_parts + (size_t)u.Bp;   // the event counter: a zero word of this upload
        small.assign(o_cnt + 1, 0u);
        for (uint32_t d = 0; d < n_parts; d++) {
          small[d] = (!have_cut || ratchet || store_trees) ? UINT32_MAX : mp_max + 1u;            // max cost + 1 (costs are full lengths here)
          small[n_parts + d] = R;
        }
        for (int c2 = 0; c2 < u.Bl; c2++) small[(size_t)2 * n_parts + (size_t)c2] = ufb_event_bound((uint32_t)u.ids[(size_t)c2]);
        UCHK(u.h_small.reserve(small.size() + 4));
        std::memcpy(u.h_small.p, small.data(), small.size() * sizeof(uint32_t));
        UCHK(u.thr.reserve(small.size() + 4));
        UCHK(u.info.reserve((size_t)std::max<uint32_t>(n_idx, 1u)));
        const uint32_t nch = ufb_chunks(std::max<uint32_t>(n_idx, 1u));
        UCHK(u.cmin.reserve((size_t)nch * (size_t)u.Bp));
        UCHK(u.pre.reserve((size_t)nch * (size_t)u.Bp));
        UCHK(u.evcount.reserve(4));
        if (u.ev.cap == 0) { UCHK(u.ev.reserve(1u << 18)); UCHK(u.h_ev.reserve(1u << 18)); }
        UCHK(hipMemcpyAsync(u.thr.p, u.h_small.p, small.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st_));
        if (n_idx) UCHK(hipMemcpyAsync(u.info.p, hinfo.data(), (size_t)n_idx * sizeof(uint2), hipMemcpyHostToDevice, st_));
        const uint32_t *d_thr = u.thr.p, *d_home = u.thr.p + n_parts, *d_best = u.thr.p + 2 * n_parts;
        if (ratchet) {
          UCHK(u.d_col.reserve((size_t)rows_p));
          UCHK(u.h_col.reserve((size_t)rows_p));
          UCHK(launch_ufb_column(st_, u.C.p, u.Bp, oc, rows, u.d_col.p));
          UCHK(hipMemcpyAsync(u.h_col.p, u.d_col.p, (size_t)rows * sizeof(int32_t), hipMemcpyDeviceToHost, st_));
        }
        uint32_t n_ev = 0, n_eager = 0;
        uint32_t *d_evcount = u.thr.p + o_cnt;
        for (bool again = false; n_idx; again = true) {
          if (again) UCHK(hipMemsetAsync(d_evcount, 0, sizeof(uint32_t), st_));
          UCHK(launch_ufb_events(st_, u.info.p, d_out(), d_thr, d_home, nullptr, u.C.p, u.Bp, u.Bl, u.rt.p, d_best, n_idx, u.cmin.p, u.pre.p,
                                 u.ev.p, (uint32_t)u.ev.cap, d_evcount, (u.topboot || u.distinct || store_trees) ? 1 : 0));
          UCHK(hipMemcpyAsync(u.h_small.p, d_evcount, sizeof(uint32_t), hipMemcpyDeviceToHost, st_));
          n_eager = (uint32_t)std::min<size_t>(u.ev.cap, 4096);     // (as in spr_sweeps_ufboot: one round trip for a small batch)
          UCHK(u.h_ev.reserve((size_t)n_eager));
          UCHK(hipMemcpyAsync(u.h_ev.p, u.ev.p, (size_t)n_eager * sizeof(UfbEvent), hipMemcpyDeviceToHost, st_));
          UCHK(hipStreamSynchronize(st_));
          n_ev = u.h_small.p[0];
          if (n_ev <= u.ev.cap) break;
          UCHK(u.ev.reserve((size_t)n_ev));
          UCHK(u.h_ev.reserve((size_t)n_ev));
        }
        if (!n_idx) UCHK(hipStreamSynchronize(st_));
        if (timing_) {
          float ms = 0;
          if (hipEventElapsedTime(&ms, ev2_, ev3_) == hipSuccess) u.gemm_ms += ms;
        }
        if (n_ev > n_eager) {
          if (u.h_ev.cap < (size_t)n_ev) n_eager = 0;
          UCHK(u.h_ev.reserve((size_t)n_ev));
          UCHK(hipMemcpyAsync(u.h_ev.p + n_eager, u.ev.p + n_eager, (size_t)(n_ev - n_eager) * sizeof(UfbEvent), hipMemcpyDeviceToHost, st_));
          UCHK(hipStreamSynchronize(st_));
        }
        events.assign(u.h_ev.p, u.h_ev.p + n_ev);
        for (UfbEvent &ev : events) ev.b = (uint32_t)u.ids[(size_t)ev.b];
        if (u.exchange) {
          // sample-sharded run: every rank replays the events of all ranks (one all-gather per batch)
          const mpf_ufb_event *all = nullptr;
          uint32_t n_all_ev = 0;
          if (u.exchange(u.exchange_arg, exchange_tag++, reinterpret_cast<const mpf_ufb_event *>(events.data()), (uint32_t)events.size(), &all, &n_all_ev) != 0) {
            set_error("online UFBoot: event exchange failed (ranks out of step?)");
            return MPF_E_STATE;
          }
          const UfbEvent *pa = reinterpret_cast<const UfbEvent *>(all);
          events.assign(pa, pa + n_all_ev);
        }
        sort_events(events, ev_tmp, ev_count, std::max<uint32_t>(n_idx, 1u), (uint32_t)u.B);
        u.events += n_ev;
        if (ratchet) {
          lcol.assign(u.h_col.p, u.h_col.p + rows);
          u.rt_orig = (uint32_t)lcol[(size_t)R];
          if (!stale_init) { u.stale_len = u.rt_orig; stale_init = true; }       // what the IQ-TREE kernel left for the start tree
        }
      }
      // ---- host replay in the reference's order
      size_t ep = 0;
      bool moved = false;
      int j = i;
      for (; j <= hi && !moved; j++) {
        const ScanPlan &pl = plans[(size_t)(j - i)];
        if (tie_mode_ == MPF_TIE_RANDOM) {
          insert_rec_ = remove_rec_ = -1;
          hits_ = 1;
        }
        long sel = -1;
        auto topology_key = [&](uint32_t cand_code) -> const std::string & {
          if (cand_code == 0xFFFFFFFFu) {
            if (u.self_key_epoch != (uint64_t)topo_epoch_) { canonical_topology(back_, u.self_key); u.self_key_epoch = (uint64_t)topo_epoch_; }
            return u.self_key;
          }
          ufb_candidate_topology(cand_code < (uint32_t)pl.n_p ? pl.rec : back_[pl.rec], pl.cands[(size_t)cand_code].q, mh_bk);
          canonical_topology(mh_bk, mh_key);
          return mh_key;
        };
        auto cand_topology_key = [&](uint32_t cand_code, int64_t tree_index) -> int64_t {
          return u.topo_index.emplace(topology_key(cand_code), tree_index).first->second;
        };
        // the update rule of one booked tree for one sample (b, score s): shared by the events the device extracted and by the
        // current tree's own bookings, which the host walks through itself
        const UfbDeferCtx dctx{};                   // (no deferred mode on this engine)
        auto one_event = [&](const uint32_t b, const uint32_t s, int64_t &tree_index, bool &looked_up, const uint32_t cand_code) {
          ufb_one_event(b, s, tree_index, looked_up, cand_code, [&](int64_t ti, uint32_t cc) { return cand_topology_key(cc, ti); }, dctx);
        };
        auto replay_events = [&](uint32_t idx, int64_t tree_index, uint32_t cand_code) {
          while (ep < events.size() && events[ep].idx < idx) ep++;
          bool looked_up = store_trees;              // (-storetrees: tree_str is set at the top, no lookup per sample)
          for (; ep < events.size() && events[ep].idx == idx; ep++) one_event(events[ep].b, events[ep].s, tree_index, looked_up, cand_code);
        };
        // the current tree scores R_T[b] for every sample: no device events for its slots (they would be B per prune-node visit,
        // 2.0e6 per move-less C3 sweep, all to be copied and ordered) -- the host has R_T and offers it to every sample in order
        auto replay_self = [&](int64_t tree_index) {
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
          const uint32_t self_len = asym ? out[(size_t)pl.self_idx] : randomMP;       // (:2285: mp of evaluateParsimony(p))
          bool pass;
          if (!ratchet) pass = !none_pass && self_len <= mp_max;
          else {
            pass = (store_trees || !u.gate_closed) && have_C && !none_pass && u.stale_len <= mp_max;
            if (!pass) u.gate_closed = true;
          }
          u.cur_logl_now = -(int32_t)(ratchet ? u.stale_len : self_len);
          const int64_t tree_index = book_tree(ratchet ? u.stale_len : self_len, pass, 0xFFFFFFFFu);
          if (tree_index >= 0) {
            if (host_self) replay_self(tree_index); else replay_events((uint32_t)pl.self_idx, tree_index, 0xFFFFFFFFu);
            if (ratchet) u.stale_len = asym ? (uint32_t)lcol[(size_t)pl.self_idx] : u.rt_orig;
          }
        }
        for (size_t c = 0; c < pl.cands.size(); c++) {
          const uint32_t idx = pl.cands[c].out;
          const uint32_t mp = out[idx];
          bool pass;
          if (!ratchet) pass = !none_pass && mp <= mp_max;
          else {
            pass = (store_trees || !u.gate_closed) && have_C && !none_pass && u.stale_len <= mp_max;
            if (!pass) u.gate_closed = true;
          }
          u.cur_logl_now = -(int32_t)(ratchet ? u.stale_len : mp);
          const int64_t tree_index = book_tree(ratchet ? u.stale_len : mp, pass, (uint32_t)c);
          if (tree_index >= 0) {
            replay_events(idx, tree_index, (uint32_t)c);
            if (ratchet) u.stale_len = (uint32_t)lcol[(size_t)idx];
          }
          if (tie_mode_ == MPF_TIE_RANDOM) {
            if (mp < best_) hits_ = 1;
            else if (mp == best_) hits_++;
            if (mp < best_ || (mp == best_ && tie_draw() <= 1.0 / (double)hits_)) { best_ = mp; sel = (long)c; }
          } else if (mp < best_) {
            best_ = mp; sel = (long)c;
          }
        }
        if (sel >= 0) {
          insert_rec_ = pl.cands[(size_t)sel].q;
          remove_rec_ = sel < pl.n_p ? pl.rec : back_[pl.rec];
        }
        // topologies of the trees accepted during this prune node's scan that some sample still points to
        for (const UfbState::Pending &pe : u.pending) {
          if (u.refs[(size_t)pe.tree_index] <= 0) continue;
          if (pe.cand == 0xFFFFFFFFu) {
            if (!u.store.count(pe.tree_index)) { u.store.emplace(pe.tree_index, back_); u.stored++; }
            continue;
          }
          ufb_store_tree(pe.tree_index, pe.cand < (uint32_t)pl.n_p ? pl.rec : back_[pl.rec], pl.cands[(size_t)pe.cand].q);
        }
        u.pending.clear();
        bool accept;
        if (tie_mode_ == MPF_TIE_RANDOM) {
          if (best_ == randomMP) iter_hits++;
          if (best_ < randomMP) iter_hits = 1;
          accept = (best_ < randomMP || (best_ == randomMP && tie_draw() <= 1.0 / (double)iter_hits)) && remove_rec_ >= 0 && insert_rec_ >= 0;
        } else {
          accept = best_ < randomMP;
        }
        if (accept) {
          if (sel < 0) { set_error("online UFBoot: accepted move without a candidate of this prune node"); return MPF_E_STATE; }
          moves_.push_back(Move{remove_rec_, insert_rec_, best_});
          apply_move(remove_rec_, insert_rec_);
          randomMP = best_;
          moved = true;
        }
      }
      batch = next_batch(batch, moved, j - i, total);
      i = j;
    }
  } while (randomMP < startMP);
  climb_finished(total);
  u.rt_valid = false;
  if (u.exchange) {
    // closing handshake: a rank that took another path would be in the middle of a batch here
    const mpf_ufb_event *all = nullptr;
    uint32_t n_all_ev = 0;
    if (u.exchange(u.exchange_arg, 0xFFFFFFFFu, nullptr, 0, &all, &n_all_ev) != 0) { set_error("online UFBoot: ranks out of step at the end of the climb"); return MPF_E_STATE; }
  }
  if (final_score) *final_score = randomMP;
  return MPF_OK;
}

}  // namespace mpf
