// search.cpp -- the reference's host-side search loops, driven by batched device scans.
//
// What stays on the host is exactly what is sequential in the reference: the order in which
// prune nodes are visited, the accept / tie-break rules with their random draws, and the
// topology edits.  What the reference does one insertion test at a time (insertParsimony +
// evaluateParsimony per candidate) arrives here as arrays of scores computed by k_scan for a
// whole batch of prune nodes; the batch is speculative -- as soon as a move is accepted the
// remaining scans of the batch are discarded, the views are refreshed and scanning resumes at
// the next prune node, so the trajectory is the reference's.
#include <algorithm>
#include <climits>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "../csrc/engine.hpp"
#include "../csrc/grow.hpp"
#include "simd_util.hpp"

namespace mpf {

static bool sweep_trace_env() { static const bool on = std::getenv("MPF_UFB_TRACE") != nullptr; return on; }
static double sw_now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// restoreTreeRearrangeParsimony (reference sprparsimony.cpp:2379-2384, :2191-2205)
void Engine::apply_move(int p, int q)
{
  const int a = back_[nx(p)], b = back_[nx(nx(p))];
  hookup(a, b);                                  // removeNodeParsimony :2245-2257
  const int r = back_[q];
  hookup(nx(p), q);
  hookup(nx(nx(p)), r);
  // the vectors whose subtree contains an edited node are stale; everything else stays valid
  invalidate_node(num(a));
  invalidate_node(num(b));
  invalidate_node(num(p));
  invalidate_node(num(q));
  invalidate_node(num(r));
  stats.moves_applied++;
}

// the sweep loop shared by pllOptimizeSprParsimony (reference sprparsimony.cpp:3295-3316) and
// _pllMakeParsimonyTreeFast (:3185-3206); MPF_TIE_FIRST follows pllrepo/src/fastDNAparsimony.c:1919-1938
int Engine::spr_sweeps(int mintrav, int maxtrav, uint32_t randomMP, uint32_t *final_score)
{
  SweepCursor c;
  c.randomMP = randomMP;
  int rc = spr_sweeps_run(mintrav, maxtrav, c, 0u);
  if (rc) return rc;
  if (final_score) *final_score = c.randomMP;
  return MPF_OK;
}

// The same loop with a way out (stop_len != 0): it returns IN FRONT of the first prune-node visit that would test a tree of at
// most stop_len -- or at once when the current tree is that short --, nothing of that visit consumed (c.stopped, c.i = its index
// in the sweep under way, c.startMP / c.iter_hits the sweep's state).  With -bb and a logl_cutoff in force, trees above the
// cut-off never reach IQTree::saveCurrentTree's bookkeeping (iqtree.cpp:3343): up to that visit the tracked climb IS the plain one
// (Engine::spr_sweeps_ufboot takes over from there).
int Engine::spr_sweeps_run(int mintrav, int maxtrav, SweepCursor &cur, uint32_t stop_len)
{
  uint32_t startMP, randomMP = cur.randomMP;
  unsigned iter_hits = 1;                         // bestIterationScoreHits
  cur.stopped = false;
  auto stop_at = [&](int i, uint32_t start_mp) {
    cur.stopped = true; cur.i = i; cur.startMP = start_mp; cur.randomMP = randomMP; cur.iter_hits = iter_hits;
    return MPF_OK;
  };
  const int total = 2 * n_ - 2;
  // (batches of a few hundred prune nodes -- what is left of a sweep -- are cheaper on the whole-sweep path of scan_batch: every stale
  //  vector refreshed by the level kernel, no closure of the batch's roots on the host)
  struct SmallMax { int &ref; int saved; ~SmallMax() { ref = saved; } } small_guard{small_batch_max_, small_batch_max_};
  small_batch_max_ = std::min(small_batch_max_, 128);
  // (the engine's own plan storage: a whole-sweep batch on a topology whose sweep is still planned -- the closing sweep of
  //  the previous climb, the same tree under other weights -- reuses descriptors and device program, Engine::scan_batch)
  std::vector<ScanPlan> &plans = sweep_plans_;
  const uint32_t *out = nullptr;
  int batch = first_batch();
  // While accepted moves are dense every move depends on the one before it and a host-driven batch is one launch chain
  // + one synchronisation per move: those stretches run inside ONE persistent kernel per sweep (k_climb, csrc/climb.hip),
  // which hands back when moves get rare -- there the host's whole-chip batches are the better tool.  Same trajectory
  // either way (the kernel follows the same rules with the same tie stream); a host-supplied random_double() callback
  // keeps the loop on the host.
  const int mt_eff = std::min(maxtrav, ntips_ - 3);
  // (... and only where all of the kernel's workgroups fit the chip together, if need be on wider tiles: climb_fit_vw)
  bool dev_ok = climb_device_ > 0 && max_visits_ == 0 && !rand_fn_ && !sankoff_ && mintrav == 1 && ntips_ == n_ && scan_mode_ == 1 && climb_supported(g_, n_, mt_eff, climb_batch_bound(stop_len != 0)) &&
                climb_fit_vw() > 0;
  uint32_t sweep_moves = 0;
  bool first_sweep = true;
  do {
    startMP = randomMP;
    node_rectifier();
    int i = 1;
    // (auto mode: a sweep starts on the device if moves were dense lately -- an engine that has just refined nearly optimal
    //  trees starts on the host)
    bool dev = dev_ok && (climb_device_ >= 2 || (first_sweep ? (gap_est_ < 0 || gap_est_ < 64.0) : sweep_moves >= 16u));
    first_sweep = false;
    sweep_moves = 0;
    while (i <= total) {
      if (visits_out()) { cur.randomMP = randomMP; cur.iter_hits = iter_hits; return MPF_OK; }
      if (stop_len && randomMP <= stop_len) return stop_at(i, startMP);     // the current tree itself is booked from here on
      if (dev) {
        uint32_t reason = 0, nm = 0;
        const int i0 = i;
        climb_stop_len_ = stop_len;
        const double tr0 = sweep_trace_env() ? sw_now_ms() : 0.0;
        int rc = climb_segment(mt_eff, total, &i, &randomMP, &iter_hits, climb_device_ < 2, &reason, &nm);
        climb_stop_len_ = 0;
        if (rc) return rc;
        if (sweep_trace_env()) std::fprintf(stderr, "[sweep] k_climb i %d -> %d moves %u reason %u len %u | %.3f ms\n", i0, i, nm, reason, randomMP, sw_now_ms() - tr0);
        // (not resident in time -- somebody else holds the chip: the rest of this climb runs as host-driven batches; trying again
        //  behind every move would cost a 30 ms time-out each)
        if (reason == CLIMB_ABORT) { dev = false; dev_ok = false; continue; }
        sweep_moves += nm;
        if (nm) {
          const double g = (double)(i - i0) / (double)nm;
          gap_est_ = gap_est_ < 0 ? g : std::exp(0.7 * std::log(gap_est_ + 1.0) + 0.3 * std::log(g + 1.0)) - 1.0;
          since_move_ = 0;
        } else {
          since_move_ += i - i0;
        }
        if (reason == CLIMB_IDLE) { dev = false; batch = std::min(total, std::max(batch, 64)); }
        if (reason == CLIMB_CUTOFF) return stop_at(i, startMP);
        continue;
      }
      // what a batch costs is a step function of its size: up to 32 prune nodes (neighbourhoods cut into short parts, chained refresh,
      // the device-walked kernel, results polled) 45-110 us; anything larger pays the long parts' dependent chains and the level
      // refresh -- 0.25-0.4 ms for 64 .. 256 prune nodes, 0.22 ms for the whole rest of a sweep on the planned path.  So: small, or
      // all that is left of the sweep; and where moves have been lying dozens of prune nodes apart, 32 at once rather than 4, 8, 16
      if (batch > 32) batch = total;
      else if (gap_est_ >= 32.0) batch = 32;
      const int hi = visits_cap(i, std::min(total, i + batch - 1));
      const double tr0 = sweep_trace_env() ? sw_now_ms() : 0.0;
      const double tp0 = stats.host_plan_ms_total, tv0 = stats.host_views_ms_total, ts0 = stats.host_scan_ms_total;
      int rc = scan_batch(plans, nodep_.data() + i, hi - i + 1, mintrav, maxtrav, &out);
      if (rc) return rc;
      const double tr1 = sweep_trace_env() ? sw_now_ms() : 0.0;
      bool moved = false;
      int j = i;
      bool cut = false;
      for (; j <= hi && !moved; j++) {
        const ScanPlan &pl = plans[(size_t)(j - i)];
        if (stop_len && stop_len >= pl.base) {
          // (a visit with an insertion test the tracker would book: hand back in front of it)
          if (pl.walked) {
            for (int part = 0; part < pl.n_parts && !cut; part++)
              cut = first_le(out + pl.part_off[part], 0, pl.part_cnt[part], stop_len - pl.base) < pl.part_cnt[part];
          } else {
            for (size_t c = 0; c < (size_t)pl.n_total && !cut; c++) cut = pl.base + pl.cost(c, out) <= stop_len;
          }
          if (cut) break;
        }
        if (tie_mode_ == MPF_TIE_RANDOM) {
          insert_rec_ = remove_rec_ = -1;
          hits_ = 1;
        }
        // testInsertParsimony's bookkeeping (reference :2168-2176 / fastDNAparsimony.c:1224-1229);
        // the chosen candidate is remembered by index and only named (record q) if the move is applied
        long sel = -1;
        // a candidate worse than the best so far changes nothing (no counter, no draw): that is nearly all of them, so
        // the loop runs over the parts' output blocks with that test first
        auto offer = [&](uint32_t mp, long c) {
          if (mp > best_) return;
          if (tie_mode_ == MPF_TIE_RANDOM) {
            if (mp < best_) hits_ = 1;
            else hits_++;
            if (mp < best_ || tie_draw() <= 1.0 / (double)hits_) { best_ = mp; sel = c; }
          } else if (mp < best_) {
            best_ = mp; sel = c;
          }
        };
        if (pl.walked) {
          long c = 0;
          for (int part = 0; part < pl.n_parts; part++) {
            const uint32_t *o = out + pl.part_off[part];
            const int cnt = pl.part_cnt[part];
            // (only a candidate with cost <= best_ - base matters; best_ can only fall while the block is read)
            for (int k = 0; best_ >= pl.base && (k = first_le(o, k, cnt, best_ - pl.base)) < cnt; k++) offer(pl.base + o[k], c + k);
            c += cnt;
          }
        } else {
          const size_t nc = (size_t)pl.n_total;
          for (size_t c = 0; c < nc; c++) offer(pl.base + pl.cost(c, out), (long)c);
        }
        if (sel >= 0) {
          insert_rec_ = candidate_record(pl, (size_t)sel);
          remove_rec_ = sel < pl.n_p ? pl.rec : back_[pl.rec];
        }
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
          moves_.push_back(Move{remove_rec_, insert_rec_, best_});
          apply_move(remove_rec_, insert_rec_);
          randomMP = best_;
          moved = true;
          sweep_moves++;
        }
      }
      if (sweep_trace_env()) std::fprintf(stderr, "[sweep] host batch i %d np %d used %d moved %d cut %d len %u | scan %.3f (plan %.3f views %.3f scan %.3f) select %.3f ms\n", i, hi - i + 1, j - i, (int)moved, (int)cut, randomMP, tr1 - tr0,
                                         stats.host_plan_ms_total - tp0, stats.host_views_ms_total - tv0, stats.host_scan_ms_total - ts0, sw_now_ms() - tr1);
      batch = next_batch(batch, moved, j - i, total);
      visits_done_ += j - i;
      i = j;
      if (cut) return stop_at(i, startMP);
      // moves have become dense again behind a quiet stretch: the rest of the sweep goes back to the kernel -- in a sweep that HAS been
      // dense so far (a move per sixteen prune nodes).  The later sweeps of a climb have a dozen moves in two thousand prune nodes; two
      // of them close together sent the kernel in for nothing (measured on the search iterations of a -bb run: five such launches of
      // 0.6-0.9 ms for five moves in one climb)
      if (moved && dev_ok && gap_est_ >= 0 && gap_est_ < 24.0 && i <= total && (uint64_t)sweep_moves * 16u >= (uint64_t)i) dev = true;
    }
  } while (randomMP < startMP && !visits_out());
  climb_finished(total);
  cur.randomMP = randomMP;
  cur.iter_hits = iter_hits;
  return MPF_OK;
}

// pllOptimizeSprParsimony (reference sprparsimony.cpp:3244-3319)
int Engine::optimize_spr(int mintrav, int maxtrav, uint32_t *score)
{
  if (!have_tree_) { set_error("no tree set"); return MPF_E_STATE; }
  moves_.clear();
  node_rectifier();
  uint32_t len = 0;
  invalidate_vectors();                      // the full evaluate of :3277: every vector again, same topology
  int rc = tree_length(&len);
  if (rc) return rc;
  best_ = len;
  ntips_ = n_;
  insert_rec_ = remove_rec_ = -1;
  visits_done_ = 0;
  if (ufb_ && !ufb_->suspended && ufb_->snk) return spr_sweeps_ufboot_snk(mintrav, maxtrav, best_, score);
  if (ufb_ && !ufb_->suspended) {                 // perSiteScores = gbo_replicates > 0 (reference :3245)
    if (scan_mode_ != 1) { set_error("online UFBoot needs the device-walked scan (option scan_mode 1)"); return MPF_E_UNSUPPORTED; }
    ufb_->rt_valid = false;
    return spr_sweeps_ufboot(mintrav, maxtrav, best_, score);
  }
  return spr_sweeps(mintrav, maxtrav, best_, score);
}

// makePermutationFast + buildSimpleTree + the addition loop of _pllMakeParsimonyTreeFast
// (reference sprparsimony.cpp:2221-2242, :1955-1981, :3107-3181) with stepwiseAddition (:2977-3019)
// evaluated as one JOIN scan per added taxon.
int Engine::addition_phase(int64_t seed, uint32_t *best_per_step, int32_t *insert_per_step)
{
  const int n = n_;
  std::vector<int> perm((size_t)n + 2);
  randum_seed_ = seed;
  for (int i = 1; i <= n; i++) perm[i] = i;
  for (int i = 1; i <= n; i++) {
    const double d = randum(&randum_seed_);
    const int k = (int)((double)(n + 1 - i) * d);
    std::swap(perm[i], perm[i + k]);
  }
  ntips_ = 0;
  nextnode_ = n + 1;
  const int ip = perm[1], iq = perm[2], ir = perm[3];
  start_ = nodep_[std::min(ip, std::min(iq, ir))];
  ntips_ = 3;
  {
    const int p = nodep_[ip];
    hookup(p, nodep_[iq]);
    const int s = nodep_[nextnode_++];
    hookup(nodep_[ir], s);
    back_[nx(s)] = back_[nx(nx(s))] = -1;
    const int r = back_[p];
    hookup(nx(s), p);
    hookup(nx(nx(s)), r);
  }
  have_tree_ = true;
  invalidate_all();
  const int f = start_;
  hits_ = 1;
  std::vector<ScanPlan> plans(1);
  std::vector<uint32_t> out;
  std::vector<int> stack;
  std::vector<uint32_t> out_of(back_.size(), 0u);
  // length of the three-taxon tree; afterwards the length of the tree built so far is the best score of the
  // previous addition (Fitch/Sankoff length of the tree with the tip inserted IS that candidate's score)
  uint32_t len = 0;
  {
    int rc = tree_length(&len);
    if (rc) return rc;
  }
  // the whole loop below as ONE persistent kernel (k_grow, csrc/grow.hip) where it applies: same insertions, same draws
  if (grow_device_ && !rand_fn_ && !sankoff_ && grow_supported(g_, n_)) {
    bool done = false;
    int rc = grow_segment(perm, len, best_per_step, insert_per_step, &done);
    if (rc) return rc;
    if (done) return MPF_OK;
  }
  while (ntips_ < n) {
    best_ = (uint32_t)INT_MAX;
    const int nextsp = ++ntips_;
    const int p = nodep_[perm[nextsp]];
    const int q = nodep_[nextnode_++];
    back_[p] = q;
    back_[q] = p;
    // score the insertion on EVERY branch and apply the reference's descent cut (:3014) afterwards on the host: the scan
    // program (topology only) is built first and goes up with the refresh of the views of the tree built so far, the
    // refresh launch clears the program's outputs: one upload, two launches, one copy back, one synchronisation per taxon
    stack.clear();
    stack.push_back(back_[f]);
    while (!stack.empty()) {
      const int c = stack.back();
      stack.pop_back();
      ScanOp o;
      o.own = slot(c);
      o.sib = slot(back_[c]);
      o.meta = (uint32_t)SCAN_JOIN << 16;
      o.out = prog_out_;
      out_of[(size_t)c] = prog_out_++;
      prog_ops_.push_back(o);
      if (!tip(c)) {
        stack.push_back(back_[nx(nx(c))]);
        stack.push_back(back_[nx(c)]);
      }
    }
    // the joins are independent of each other: cut them into short programs so that the launch
    // has (candidates/4 x tiles) waves instead of one wave per tile
    for (size_t b = 0; b < prog_ops_.size(); b += 4) {
      ScanHdr h;
      h.op_begin = (uint32_t)b;
      h.op_end = (uint32_t)std::min(prog_ops_.size(), b + 4);
      h.s_slot = slot(p);
      h.pad = 1;                                   // stepwise addition: the new tip is the root side of every test (k_snk_scan, ASYM)
      prog_hdr_.push_back(h);
    }
    {
      hipError_t he = reserve_results(prog_out_);
      if (he != hipSuccess) { set_error(std::string("HIP: ") + hipGetErrorString(he)); return MPF_E_HIP; }
    }
    ride_[0].src = prog_ops_.data();
    ride_[0].bytes = prog_ops_.size() * sizeof(ScanOp);
    ride_[1].src = prog_hdr_.data();
    ride_[1].bytes = prog_hdr_.size() * sizeof(ScanHdr);
    zero_req_ptr_ = d_out();
    zero_req_words_ = clear_words(prog_out_);
    want_host_results_ = true;                     // counts and join costs land in the host's buffers: no copy-back dispatch
    int rc = schedule_views(nullptr);
    ride_[0].src = ride_[1].src = nullptr;
    zero_req_ptr_ = nullptr;
    zero_req_words_ = 0;
    if (!rc) rc = run_scans(plans, out);
    want_host_results_ = false;
    cnt_on_host_ = false;
    if (rc) return rc;
    // candidate branches in the reference's DFS order with its descent cut
    stack.clear();
    stack.push_back(back_[f]);
    while (!stack.empty()) {
      const int c = stack.back();
      stack.pop_back();
      const uint32_t mp = (sankoff_ ? 0u : len) + out[out_of[(size_t)c]];   // weighted: the join kernel returns the full length
      if (tie_mode_ == MPF_TIE_RANDOM) {
        if (mp < best_) hits_ = 1;
        else if (mp == best_) hits_++;
        if (mp < best_ || (mp == best_ && tie_draw() <= 1.0 / (double)hits_)) { best_ = mp; insert_rec_ = c; }
      } else if (mp < best_) { best_ = mp; insert_rec_ = c; }
      if (!tip(c) && sc_[c] > 0) {
        stack.push_back(back_[nx(nx(c))]);
        stack.push_back(back_[nx(c)]);
      }
    }
    len = best_;
    if (best_per_step) best_per_step[nextsp] = best_;
    if (insert_per_step) insert_per_step[nextsp] = insert_rec_;
    const int r = back_[insert_rec_];
    hookup(nx(q), insert_rec_);
    hookup(nx(nx(q)), r);
    // the new node's three vectors have never been valid, so invalidate_node will not count them
    if (n_invalid_ >= 0) n_invalid_ += 3;
    invalidate_node(num(q));
    invalidate_node(num(insert_rec_));
    invalidate_node(num(r));
  }
  return MPF_OK;
}

int Engine::stepwise_addition(int64_t seed, uint32_t *best_per_step, int32_t *insert_per_step, uint32_t *score)
{
  moves_.clear();
  int rc = addition_phase(seed, best_per_step, insert_per_step);
  if (rc) return rc;
  if (score) *score = best_;
  return MPF_OK;
}

// _pllComputeRandomizedStepwiseAdditionParsimonyTree (reference sprparsimony.cpp:3224-3235)
int Engine::make_parsimony_tree(int64_t seed, int spr_dist, uint32_t *score)
{
  moves_.clear();
  int rc = addition_phase(seed, nullptr, nullptr);
  if (rc) return rc;
  node_rectifier();
  uint32_t fin = best_;
  rc = spr_sweeps(1, spr_dist, best_, &fin);
  if (rc) return rc;
  if (score) *score = fin;
  return MPF_OK;
}

}  // namespace mpf
