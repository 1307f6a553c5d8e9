// ufboot_pipe.cpp -- the tracked climb (-bb) as a two-stage pipeline: decisions from the scan's costs, the replay beside the next batch, the deferred log on a second host thread (ufb_books.hpp)
#include "ufboot_common.hpp"

namespace mpf {

// ---- the tracker's climb as a two-stage pipeline ----------------------------------------------------------------
// Default update rule, no cut-off in force (the first climb of a run -- where the time goes).
// The bookkeeping needs the product and the event extraction of a batch; the SEARCH mostly does not: whenever the costs alone
// settle what the sweep does next -- every prune node of the batch strictly worse than the current tree, or the first one that
// is better has ONE cheapest candidate (draws among dearer ties are overridden by it, sprparsimony.cpp:2168-2176, and a strictly
// better tree is accepted without a draw, :3306-3311) -- the move is applied and the next batch planned and launched as soon as
// the scan's results are on the host, while the device still multiplies and extracts the batch in front.  The replay of that
// batch (all draws in the reference's order, one stream) then runs beside the next batch's device work and must arrive at the
// very decision taken early (checked).  Where a draw decides (ties with the current tree, two cheapest candidates) the batch is
// taken as before: replay first, then move.
int Engine::spr_sweeps_ufboot_pipe(int mintrav, int maxtrav, uint32_t randomMP, uint32_t *final_score)
{
  UfbState &u = *ufb_;
  uint32_t startMP;
  unsigned iter_hits = 1;
  const int total = 2 * n_ - 2;
  struct Batch {
    int i = 0, hi = 0, np = 0, par = 0;
    std::vector<ScanPlan> plans;
    uint32_t n_idx = 0, n_parts = 0, n_self = 0, n_eager = 0;
    bool device = false;                           // false: nothing to scan (no insertion test in the whole batch)
    bool early_bounds = false;                     // launched before the replay of the batch in front
    std::vector<uint32_t> out;                     // the scan's costs and info, taken off the shared pinned buffers
    std::vector<uint2> info;
  } ring[3];
  struct Decision { bool certain = false, moved = false; int j = 0; long sel = -1; uint32_t score = 0, sel_idx = 0, sel_home = 0; int ins = -1, rem = -1; };
  int cur = 0;
  bool prelaunched = false;
  bool moved_once = false;
  int pipe_ahead = 0;
  if (u.exchange) {
    // sample-sharded run: every rank must cut the climb into the same batches -- start from a fixed batch policy state
    gap_est_ = -1.0;
    since_move_ = 0;
  }
  const bool host_self = !u.exchange;              // (sample-sharded: R_T lives in pieces on the ranks, the events are exchanged anyway)
  uint32_t exchange_tag = 0;
  int batch = first_batch();
  std::vector<UfbEvent> events, ev_tmp;
  std::vector<uint32_t> ev_count;
  std::vector<int32_t> snap_back;
  int32_t snap_epoch = 0;
  if (!u.rt_valid) { int rc = ufb_current_tree_reps(); if (rc) return rc; }
  for (int k = 0; k < 2; k++) {
    UCHK(u.p_flag_s[k].reserve(4));
    UCHK(u.p_flag_e[k].reserve(4));
    UCHK(u.p_rt[k].reserve((size_t)u.Bp));
  }
  // every event a batch can produce while the next one may already be in flight fits the device buffer AND the pinned one the
  // extraction kernel writes as it emits (a copy of "the rest" would queue up behind the next batch, which reuses the device buffer)
  if (u.ev.cap < (size_t)ufb_event_cap_) UCHK(u.ev.reserve((size_t)ufb_event_cap_));
  for (int k = 0; k < 2; k++) UCHK(u.p_ev[k].reserve(u.ev.cap));
  uint32_t *d_evcount = d_done_.p + 48, *d_fin = d_done_.p + 32, *d_cut = d_done_.p + 56;
  bool log_open = false;
  // the deferred log of every batch goes to a second host thread (option ufb_thread): it owns the tracker's deferred state
  // (topology map, boot_trees, reference counts, stored topologies) for the length of this climb and works on copies of the
  // topology and the plans, this thread never looks at that state before the worker has been joined
  using Worker = books::LogWorker<ScanPlan>;     // (host/ufb_books.hpp: the same code tests/cpu/ufb_books_test.cpp runs under the thread sanitizer)
  using Job = Worker::Job;
  Worker worker;
  worker.n_taxa = n_;
  worker.d = &u;
  const bool use_worker = ufb_thread_ != 0;
  // MPF_UFB_RECORD=<path>: what this climb hands its worker goes to a file as well (appended: one climb after the other), with
  // the deferred state in front of and behind it -- the stream tests/cpu/ufb_books_test.cpp replays without a GPU
  struct Recording {
    std::FILE *f = nullptr;
    ~Recording() { if (f) std::fclose(f); }
  } recording;
  if (const char *path = use_worker ? std::getenv("MPF_UFB_RECORD") : nullptr) {
    std::FILE *probe = std::fopen(path, "rb");
    const bool fresh = probe == nullptr;
    if (probe) std::fclose(probe);
    recording.f = std::fopen(path, "ab");
    if (recording.f) {
      if (fresh) { books::rec::put(recording.f, "UFBREC3", 8); books::rec::put1<int32_t>(recording.f, n_); }
      books::rec::write_state(recording.f, 'D', u, u.treels.size());
    }
  }
  // any return but the last one leaves launches in flight and a batch half consumed: wait for the device, forget what the
  // engine believes about the views and the pending scan, so that the next call starts from the topology alone
  struct Abort {
    Engine *e;
    Worker *w;
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
      // (the log worker booked trees without growing the reference counts: they follow treels on EVERY way out, or a later
      //  climb on the other paths -- which push both in lockstep -- would index past their end)
      w->finish();
      if (e->ufb_) {
        UfbState &u = *e->ufb_;
        u.log.clear();
        u.rt_valid = false;
        u.lookups += w->sc.lookups; u.stored += w->sc.stored; u.t_lookup += w->sc.t_lookup;
        w->sc.lookups = w->sc.stored = 0;
        w->sc.t_lookup = 0;
        if (u.refs.size() < u.treels.size()) u.refs.resize(u.treels.size(), 0);
      }
    }
  } abort_guard{this, &worker};
  uint64_t n_draws = 0;                            // (added to the tracker's counter at the end: its word shares a cache line with the worker's)

  // plan + enqueue the whole chain of the batch [i, i + b): refresh, masked scan, mid (C <- 0, self slots, scan results to the host),
  // product, extraction (events and R_T to the host)
  // (clamp: the current tree has been offered to every sample -- true once a move of this climb has been accepted --, so the
  //  extraction may start every sample's bound at R_T when the bounds it was given are one replay old)
  auto launch = [&](Batch &B, int i, int b, bool early, bool clamp) -> int {
    const double t0 = now_ms();
    B.i = i;
    B.hi = std::min(total, i + b - 1);
    if (max_visits_ > 0) B.hi = std::max(i, (int)std::min<int64_t>(B.hi, (int64_t)i + (max_visits_ - visits_done_ - (early ? pipe_ahead : 0)) - 1));
    B.np = B.hi - i + 1;
    B.early_bounds = early;
    const uint32_t *out_unused = nullptr;
    scan_masks_ = true;
    ufb_async_ = true;
    int rc = scan_batch(B.plans, nodep_.data() + i, B.np, mintrav, maxtrav, &out_unused);
    scan_masks_ = false;
    ufb_async_ = false;
    if (rc) return rc;
    B.device = walk_async_;
    u.batches++;
    if (!B.device) {                               // run_walks has finished the (empty) scan itself
      B.n_idx = B.n_parts = 0;
      B.out.clear();
      B.info.clear();
      UCHK(hipMemcpyAsync(u.p_rt[B.par].p, u.rt.p, (size_t)u.Bl * sizeof(int32_t), hipMemcpyDeviceToHost, st_));
      UCHK(hipStreamSynchronize(st_));
      u.t_scan += now_ms() - t0;
      return MPF_OK;
    }
    if (!u.st_valid) { int rc2 = ufb_stage_small(B.plans, B.np); if (rc2) return rc2; }
    B.n_idx = u.st_n_idx; B.n_parts = u.st_n_parts; B.n_self = u.st_n_self;
    if (B.n_idx != (uint32_t)walk_async_nout_) { set_error("online UFBoot: staged block out of step with the scan"); return MPF_E_STATE; }
    const int rows_p = round_up((int)std::max<uint32_t>(B.n_idx, 1u), ufb_row_padding((int)B.n_idx, u.Bp));
    UCHK(u.C.reserve((size_t)rows_p * (size_t)u.Bp));
    const uint32_t nch = ufb_chunks(B.n_idx);
    UCHK(u.cmin.reserve((size_t)nch * (size_t)u.Bp));
    UCHK(u.pre.reserve((size_t)nch * (size_t)u.Bp));
    const uint32_t *dsm = u.st_dev;
    if (!dsm) {
      UCHK(u.thr.reserve((size_t)u.st_words + 4));
      UCHK(hipMemcpyAsync(u.thr.p, u.h_small.p, (size_t)u.st_words * sizeof(uint32_t), hipMemcpyHostToDevice, st_));
      dsm = u.thr.p;
    }
    const uint32_t *d_thr = dsm, *d_home = dsm + B.n_parts, *d_pend = dsm + 2 * B.n_parts, *d_best = dsm + 3 * B.n_parts, *d_self = dsm + u.st_o_self;
    const bool small_batch = B.n_idx <= kUfbEvents2Max;   // (the chunked kernels of a larger batch do not take the cut)
    UCHK(u.h_info.reserve(walk_async_nout_));
    UfbPublishArgs ps;                             // the scan's results
    if (cnt_copy_pending_) { ps.src[0] = d_cnt(); ps.dst[0] = h_cnt(); ps.words[0] = (uint32_t)(out_off() + walk_async_nout_); }
    else { ps.src[0] = d_out(); ps.dst[0] = h_out(); ps.words[0] = (uint32_t)walk_async_nout_; }
    cnt_copy_pending_ = false;
    ps.src[1] = reinterpret_cast<const uint32_t *>(u.info.p);
    ps.dst[1] = reinterpret_cast<uint32_t *>(u.h_info.p);
    ps.words[1] = (uint32_t)(2 * walk_async_nout_);
    ps.h_flag = u.p_flag_s[B.par].p;
    ps.done = d_fin;
    __atomic_store_n(u.p_flag_s[B.par].p + 1, 0u, __ATOMIC_RELAXED);
    UCHK(launch_ufb_mid(st_, u.C.p, (size_t)rows_p * (size_t)u.Bp, u.info.p, d_self, B.n_self, host_self ? 0xFFFFFFFFu : 0xFFFFFFFEu, d_evcount, ps,
                        d_out(), d_home, d_pend, B.n_idx, d_cut));
    for (int pl = 0; pl < u.planes; pl++)
      UCHK(launch_bitgemm(st_, u.masks.p, rows_p, g_.Wp, u.wt.p + (size_t)pl * u.plane_bytes, u.Bp, u.C.p, 1 << (7 * pl), 1, nullptr, small_batch ? d_cut : nullptr));
    u.gemm_rows += (uint64_t)rows_p;
    B.n_eager = (uint32_t)std::min<size_t>(u.p_ev[B.par].cap, 0xFFFFFFFFu);
    UfbPublishArgs pe;                             // the bookkeeping's inputs
    pe.src[0] = reinterpret_cast<const uint32_t *>(u.rt.p);
    pe.dst[0] = reinterpret_cast<uint32_t *>(u.p_rt[B.par].p);
    pe.words[0] = (uint32_t)u.Bl;
    pe.h_ev = u.p_ev[B.par].p;
    pe.h_ev_cap = B.n_eager;
    pe.h_flag = u.p_flag_e[B.par].p;
    pe.done = d_fin;
    __atomic_store_n(u.p_flag_e[B.par].p + 1, 0u, __ATOMIC_RELAXED);
    UCHK(launch_ufb_events_publish(st_, u.info.p, d_out(), d_thr, d_home, nullptr, u.C.p, u.Bp, u.Bl, u.rt.p, d_best, B.n_idx, u.cmin.p, u.pre.p,
                                   u.ev.p, (uint32_t)u.ev.cap, d_evcount, 0, pe, (early && clamp) ? 1 : 0, small_batch ? d_cut : nullptr));
    u.t_scan += now_ms() - t0;
    return MPF_OK;
  };
  auto wait_flag = [&](const uint32_t *flag) -> int {
    if (!wait_host_flag(flag)) {
      UCHK(hipStreamSynchronize(st_));
      if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != 1u) { set_error("online UFBoot: a batch's results were not published"); return MPF_E_STATE; }
    }
    return MPF_OK;
  };
  // what the sweep does with this batch as far as the costs alone say it (no draw taken, no state touched)
  auto decide = [&](const Batch &B) -> Decision {
    Decision d;
    uint32_t best = best_;
    for (int j = 0; j < B.np; j++) {
      const ScanPlan &pl = B.plans[(size_t)j];
      uint32_t m = UINT32_MAX, m_cnt = 0, m_idx = 0, m_home = 0;
      long m_c = -1, c = 0;
      for (int pi = 0; pi < pl.n_parts; pi++) {
        const uint32_t home = pl.part_off[pi] + (uint32_t)pl.part_cnt[pi];
        for (int k = 0; k < pl.part_cnt[pi]; k++, c++) {
          const uint32_t idx = pl.part_off[pi] + (uint32_t)k, mp = pl.base + B.out[idx];
          if (mp < m) { m = mp; m_cnt = 1; m_c = c; m_idx = idx; m_home = home; }
          else if (mp == m) m_cnt++;
        }
      }
      if (m_c < 0 || m > best) {                   // nothing reaches the best length: no candidate is selected, whatever is drawn
        if (best > randomMP) return d;             // (never: the best length is at most the current tree's)
        continue;
      }
      if (m == best) {
        if (tie_mode_ == MPF_TIE_RANDOM) return d; // a draw picks among the ties, another accepts the move or not
        if (best < randomMP) return d;             // (never, see above)
        continue;                                  // first-best rule: an equally long tree is not a candidate
      }
      if (!(m < randomMP)) return d;
      if (tie_mode_ == MPF_TIE_RANDOM && m_cnt > 1) return d;    // which of the cheapest candidates: a draw
      d.certain = d.moved = true;                  // (first-best rule: m_c is the first of the cheapest, as the rule takes it)
      d.j = j; d.sel = m_c; d.score = m; d.sel_idx = m_idx; d.sel_home = m_home;
      return d;
    }
    d.certain = true;
    d.moved = false;
    d.j = B.np;
    return d;
  };

  do {
    startMP = randomMP;
    node_rectifier();
    int i = 1;
    bool sw_moved = false;                         // (UfbState::quiet_topo: a complete sweep of one topology without a candidate event)
    uint64_t sw_events = 0;
    while (i <= total && !visits_out()) {
      Batch &B = ring[cur];
      if (!prelaunched) { B.par = cur & 1; int rc = launch(B, i, batch, false, false); if (rc) return rc; }
      prelaunched = false;
      double t0 = now_ms();
      const uint32_t *out = nullptr;
      if (B.device) {
        { int rc = wait_flag(u.p_flag_s[B.par].p + 1); if (rc) return rc; }
        { int rc = run_walks_finish(B.plans, &out); if (rc) return rc; }
        B.out.assign(out, out + B.n_idx);
        B.info.assign(u.h_info.p, u.h_info.p + B.n_idx);
      }
      out = B.out.data();
      const uint2 *hinfo = B.info.data();
      // ---- the search's decision from the costs, and the next batch on its way
      Decision d = decide(B);
      int next_i = i, next_batch_size = batch;
      const bool overflow_safe = (uint64_t)B.n_idx * (uint64_t)u.Bl <= (uint64_t)u.ev.cap;     // (no second extraction after C has been reused)
      bool early = ufb_pipe_ && d.certain && overflow_safe;
      ufb_stat_batches_++;
      if (early) {
        ufb_stat_early_++;
        if (d.moved) {
          const ScanPlan &pl = B.plans[(size_t)d.j];
          d.ins = candidate_record(pl, (size_t)d.sel);
          d.rem = d.sel < pl.n_p ? pl.rec : back_[pl.rec];
          if (B.device) UCHK(launch_rt_update(st_, u.rt.p, u.C.p, u.Bp, hinfo[d.sel_idx].x, d.sel_home));
          snap_back = back_;
          snap_epoch = topo_epoch_;
          moves_.push_back(Move{d.rem, d.ins, d.score});
          apply_move(d.rem, d.ins);
          next_i = i + d.j + 1;
          next_batch_size = next_batch(batch, true, d.j + 1, total);
        } else {
          next_i = B.hi + 1;
          next_batch_size = next_batch(batch, false, B.np, total);
        }
        pipe_ahead = next_i - i;                   // (visits of this batch that count before the look-ahead batch starts)
        if (next_i <= total && !(max_visits_ > 0 && visits_done_ + pipe_ahead >= max_visits_)) {
          const int nxt = (cur + 1) % 3;
          ring[nxt].par = B.par ^ 1;
          int rc = launch(ring[nxt], next_i, next_batch_size, true, moved_once || d.moved);
          if (rc) return rc;
          prelaunched = true;
        }
      }
      u.t_prep += now_ms() - t0;
      // ---- the bookkeeping of this batch: the log of the one before (beside the device), then events -> order -> replay
      { const double td = now_ms(); ufb_drain_log(); u.t_defer += now_ms() - td; }
      t0 = now_ms();
      uint32_t n_ev = 0;
      if (B.device) {
        { int rc = wait_flag(u.p_flag_e[B.par].p + 1); if (rc) return rc; }
        n_ev = u.p_flag_e[B.par].p[0];
        if (n_ev > u.ev.cap) {
          if (prelaunched) { set_error("online UFBoot: event buffer overflow behind a batch launched early"); return MPF_E_STATE; }
          const uint32_t *dsm = u.st_dev ? u.st_dev : u.thr.p;
          // (the first extraction ran under the device's cut -- rows behind the batch's certain end were not multiplied --, this one
          //  does not and may find more: extract until the count it reports fits the buffer it wrote to)
          for (;;) {
            UCHK(u.ev.reserve((size_t)n_ev));
            UCHK(hipMemsetAsync(d_evcount, 0, sizeof(uint32_t), st_));
            UCHK(launch_ufb_events(st_, u.info.p, d_out(), dsm, dsm + B.n_parts, nullptr, u.C.p, u.Bp, u.Bl, u.rt.p, dsm + 3 * B.n_parts, B.n_idx, u.cmin.p, u.pre.p,
                                   u.ev.p, (uint32_t)u.ev.cap, d_evcount, 0));      // (staging: thr | home | prune-node ends | best)
            UCHK(u.h_col.reserve(4));
            UCHK(hipMemcpyAsync(u.h_col.p, d_evcount, sizeof(uint32_t), hipMemcpyDeviceToHost, st_));
            // (nothing else is in flight: both pinned buffers follow the device buffer, which the overflow rule is stated in)
            UCHK(hipStreamSynchronize(st_));
            n_ev = (uint32_t)u.h_col.p[0];
            if (n_ev <= u.ev.cap) break;
          }
          B.n_eager = 0;
          for (int k = 0; k < 2; k++) UCHK(u.p_ev[k].reserve(u.ev.cap));
        }
        if (n_ev > B.n_eager) {
          if (prelaunched) { set_error("online UFBoot: events beyond the pinned buffer behind a batch launched early"); return MPF_E_STATE; }
          if (u.p_ev[B.par].cap < (size_t)n_ev) B.n_eager = 0;
          UCHK(u.p_ev[B.par].reserve((size_t)n_ev));
          UCHK(hipMemcpyAsync(u.p_ev[B.par].p + B.n_eager, u.ev.p + B.n_eager, (size_t)(n_ev - B.n_eager) * sizeof(UfbEvent), hipMemcpyDeviceToHost, st_));
          UCHK(hipStreamSynchronize(st_));
        }
      }
      double t1 = now_ms();
      u.t_dev += t1 - t0;
      // what lies behind the batch's certain end is never replayed: dropped before the sort
      uint32_t idx_cut = B.n_idx;
      {
        int js = B.np - 1;
        for (int j = 0; j < B.np; j++) {
          const ScanPlan &pl = B.plans[(size_t)j];
          uint32_t m = UINT32_MAX;
          for (int pi = 0; pi < pl.n_parts; pi++)
            for (int k = 0; k < pl.part_cnt[pi]; k++) m = std::min(m, out[pl.part_off[pi] + (uint32_t)k]);
          if (m != UINT32_MAX && pl.base + m < randomMP) { js = j; break; }
        }
        if (js < B.np - 1) {
          idx_cut = 0;
          for (int j = 0; j <= js; j++) {
            const ScanPlan &pl = B.plans[(size_t)j];
            if (pl.self_idx >= 0) idx_cut = std::max(idx_cut, (uint32_t)pl.self_idx + 1u);
            for (int pi = 0; pi < pl.n_parts; pi++) idx_cut = std::max(idx_cut, pl.part_off[pi] + (uint32_t)pl.part_cnt[pi] + 1u);
          }
        }
      }
      if (B.device && u.p_flag_s[B.par].p[2] != idx_cut) { set_error("online UFBoot: the device's end of the batch differs from the host's"); return MPF_E_STATE; }
      {
        const UfbEvent *src = u.p_ev[B.par].p;
        if (u.exchange) {
          // every rank replays the events of all ranks (one all-gather per batch; the cut is the same everywhere: same costs)
          events.clear();
          for (uint32_t k = 0; k < n_ev; k++) {
            UfbEvent e = src[k];
            if (e.idx >= idx_cut) continue;
            e.b = (uint32_t)u.ids[(size_t)e.b];
            events.push_back(e);
          }
          if (!B.device)                             // nothing was scanned: the current tree's own bookings, from R_T on the host
            for (int jj = 0; jj < B.np; jj++)
              if (B.plans[(size_t)jj].self_idx >= 0)
                for (int c2 = 0; c2 < u.Bl; c2++)
                  if ((uint32_t)u.p_rt[B.par].p[c2] <= u.boot_score[(size_t)u.ids[(size_t)c2]])
                    events.push_back(UfbEvent{(uint32_t)B.plans[(size_t)jj].self_idx, (uint32_t)u.ids[(size_t)c2], (uint32_t)u.p_rt[B.par].p[c2]});
          n_ev = (uint32_t)events.size();
          const mpf_ufb_event *all = nullptr;
          uint32_t n_all_ev = 0;
          if (u.exchange(u.exchange_arg, exchange_tag++, reinterpret_cast<const mpf_ufb_event *>(events.data()), (uint32_t)events.size(), &all, &n_all_ev) != 0) {
            set_error("online UFBoot: event exchange failed (ranks out of step?)");
            return MPF_E_STATE;
          }
          const UfbEvent *pa = reinterpret_cast<const UfbEvent *>(all);
          events.assign(pa, pa + n_all_ev);
          uint32_t n_keys = B.n_idx;
          for (int jj = 0; jj < B.np; jj++) n_keys = std::max(n_keys, (uint32_t)(B.plans[(size_t)jj].self_idx + 1));
          sort_events(events, ev_tmp, ev_count, n_keys, (uint32_t)u.B);
        } else if (n_ev >= 512) {
          ev_count.assign((size_t)u.B + 1, 0u);
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
          ev_count.assign((size_t)B.n_idx + 1, 0u);
          for (const UfbEvent &e : ev_tmp) ev_count[(size_t)e.idx + 1]++;
          for (size_t k = 1; k <= (size_t)B.n_idx; k++) ev_count[k] += ev_count[k - 1];
          for (const UfbEvent &e : ev_tmp) events[ev_count[e.idx]++] = e;
          n_ev = kept;
        } else {
          events.clear();
          for (uint32_t k = 0; k < n_ev; k++) {
            UfbEvent e = src[k];
            if (e.idx >= idx_cut) continue;
            e.b = (uint32_t)u.ids[(size_t)e.b];
            events.push_back(e);
          }
          n_ev = (uint32_t)events.size();
          sort_events(events, ev_tmp, ev_count, B.n_idx, (uint32_t)u.B);
        }
        u.events += n_ev;
      }
      t0 = now_ms();
      u.t_sort += t0 - t1;
      // ---- replay in the reference's order (Engine::spr_sweeps_ufboot's, reduced to the default rule without a cut-off)
      const int32_t *h_rt = u.p_rt[B.par].p;
      size_t ep = 0;
      bool moved = false;
      int j = i;
      for (; j <= B.hi && !moved; j++) {
        const ScanPlan &pl = B.plans[(size_t)(j - i)];
        const int32_t cur_plan = (int32_t)(j - i);
        if (tie_mode_ == MPF_TIE_RANDOM) {
          insert_rec_ = remove_rec_ = -1;
          hits_ = 1;
        }
        long sel = -1;
        uint32_t sel_idx = 0, sel_home = 0;
        size_t c = 0;
        auto one_event = [&](const uint32_t b, const uint32_t s, const int64_t tree_index, const uint32_t cand_code) {
          uint32_t &bs = u.boot_score[b];
          bool accept = false;
          if (s < bs) accept = true;                                    // rell > boot_logl + epsilon (iqtree.cpp:3686)
          else if (s == bs) {                                           // rell > boot_logl - epsilon: tie, draw (:3687-3688)
            n_draws++;
            accept = tie_draw() <= 1.0 / (double)(u.boot_counts[b] + 1);
          }
          if (accept) {
            u.log.push_back(UfbState::LogEntry{b, cand_code, tree_index, cur_plan});
            log_open = true;
            if (u.cut_btrees) u.boot_orig[b] = u.cur_logl_now;                          // :3716-3718
            if (s < bs) { u.boot_counts[b] = 1; bs = s; }              // :3710-3719
          }
          if (s == bs) u.boot_counts[b]++;                              // :3728-3730
        };
        auto book = [&](uint32_t len) -> int64_t {                      // iqtree.cpp:3343-3348 without a cut-off
          u.treels.push_back(len);
          if (!use_worker) u.refs.push_back(0);                         // (the worker sizes the reference counts itself)
          return (int64_t)u.treels.size() - 1;
        };
        if (pl.self_idx >= 0) {
          // the current tree, once per prune node and before its insertion tests (sprparsimony.cpp:2285-2289)
          u.cur_logl_now = -(int32_t)randomMP;
          const int64_t tree_index = book(randomMP);
          if (host_self) {
            ufb_self_default(h_rt, tree_index, cur_plan, log_open, n_draws);
          } else {
            const uint32_t idx = (uint32_t)pl.self_idx;
            while (ep < events.size() && events[ep].idx < idx) ep++;
            for (; ep < events.size() && events[ep].idx == idx; ep++) one_event(events[ep].b, events[ep].s, tree_index, 0xFFFFFFFFu);
          }
        }
        for (int pi = 0; pi < pl.n_parts; pi++) {
          const uint32_t home = pl.part_off[pi] + (uint32_t)pl.part_cnt[pi];
          for (int k = 0; k < pl.part_cnt[pi]; k++, c++) {
            const uint32_t idx = pl.part_off[pi] + (uint32_t)k;
            const uint32_t mp = pl.base + out[idx];
            u.cur_logl_now = -(int32_t)mp;
            const int64_t tree_index = book(mp);                        // saveCurrentTree(-mp), sprparsimony.cpp:2163-2166
            while (ep < events.size() && events[ep].idx < idx) ep++;
            for (; ep < events.size() && events[ep].idx == idx; ep++) one_event(events[ep].b, events[ep].s, tree_index, (uint32_t)c);
            if (tie_mode_ == MPF_TIE_RANDOM) {                          // testInsertParsimony's tie rule (:2168-2176)
              if (mp < best_) hits_ = 1;
              else if (mp == best_) hits_++;
              if (mp < best_ || (mp == best_ && tie_draw() <= 1.0 / (double)hits_)) { best_ = mp; sel = (long)c; sel_idx = idx; sel_home = home; }
            } else if (mp < best_) {
              best_ = mp; sel = (long)c; sel_idx = idx; sel_home = home;
            }
          }
        }
        const bool early_here = early && d.moved && (j - i) == d.j;
        if (sel >= 0) {
          if (early_here) { insert_rec_ = d.ins; remove_rec_ = d.rem; }          // (named before the move was applied)
          else if (early) { set_error("online UFBoot: the replay selects a candidate where the costs said none is"); return MPF_E_STATE; }
          else {
            insert_rec_ = candidate_record(pl, (size_t)sel);
            remove_rec_ = sel < pl.n_p ? pl.rec : back_[pl.rec];
          }
        }
        if (log_open) { u.log.push_back(UfbState::LogEntry{0xFFFFFFFFu, 0u, 0, cur_plan}); log_open = false; }
        bool accept;
        if (tie_mode_ == MPF_TIE_RANDOM) {                              // :3306-3311
          if (best_ == randomMP) iter_hits++;
          if (best_ < randomMP) iter_hits = 1;
          accept = (best_ < randomMP || (best_ == randomMP && tie_draw() <= 1.0 / (double)iter_hits)) &&
                   remove_rec_ >= 0 && insert_rec_ >= 0;
        } else {
          accept = best_ < randomMP;
        }
        if (early && accept != early_here) { set_error("online UFBoot: the replay's decision differs from the one taken from the costs"); return MPF_E_STATE; }
        if (accept) {
          if (sel < 0) { set_error("online UFBoot: accepted move without a candidate of this prune node"); return MPF_E_STATE; }
          if (early) {
            if (sel != d.sel || best_ != d.score) { set_error("online UFBoot: the replay's move differs from the one taken from the costs"); return MPF_E_STATE; }
          } else {
            if (B.device) UCHK(launch_rt_update(st_, u.rt.p, u.C.p, u.Bp, hinfo[sel_idx].x, sel_home));
            snap_back = back_;
            snap_epoch = topo_epoch_;
            moves_.push_back(Move{remove_rec_, insert_rec_, best_});
            apply_move(remove_rec_, insert_rec_);
          }
          randomMP = best_;
          moved = true;
          moved_once = true;
        }
      }
      if (early && !d.moved && moved) { set_error("online UFBoot: a move where the costs said none is possible"); return MPF_E_STATE; }
      // the log of this batch speaks of the tree in front of its move
      if (!u.log.empty() && use_worker) {
        Job *jb = worker.get();
        jb->log.swap(u.log);
        u.log.clear();
        if (moved) { jb->back.swap(snap_back); jb->epoch = snap_epoch; }
        else { jb->back = back_; jb->epoch = topo_epoch_; }
        jb->plans = B.plans;
        if (recording.f) books::rec::write_job<ScanPlan>(recording.f, jb->log, jb->back, jb->epoch, jb->plans);
        if (!worker.submit(jb)) {
          ufb_drain(jb->log, jb->back, jb->epoch, jb->plans, worker.sc);
          jb->log.clear();
          worker.spare.push_back(jb);
        }
      } else if (!u.log.empty()) {
        if (moved) { u.log_back.swap(snap_back); u.log_epoch = snap_epoch; }
        else { u.log_back = back_; u.log_epoch = topo_epoch_; }
        u.log_plans = &B.plans;
      }
      visits_done_ += j - i;
      if (early) { batch = next_batch_size; i = next_i; }
      else { batch = next_batch(batch, moved, j - i, total); i = j; }
      cur = (cur + 1) % 3;
      sw_events += n_ev;
      sw_moved = sw_moved || moved;
      u.t_replay += now_ms() - t0;
    }
    if (!sw_moved && sw_events == 0 && i > total && host_self && ufb_memo_) {       // (no cut-off in force here: every insertion test was multiplied)
      if (u.self_key_epoch != (uint64_t)topo_epoch_) { canonical_topology(back_, u.self_key); u.self_key_epoch = (uint64_t)topo_epoch_; }
      u.quiet_topo[u.quiet_key(mintrav, maxtrav, n_)] = UINT32_MAX;
    }
  } while (randomMP < startMP && !visits_out());
  ufb_drain_log();
  worker.finish();
  u.lookups += worker.sc.lookups; u.stored += worker.sc.stored; u.t_lookup += worker.sc.t_lookup;
  u.draws += n_draws;
  if (u.refs.size() < u.treels.size()) u.refs.resize(u.treels.size(), 0);
  if (recording.f) books::rec::write_state(recording.f, 'E', u, u.treels.size());
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
