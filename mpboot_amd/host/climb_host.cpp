// climb_host.cpp -- host side of the device-resident climb (csrc/climb.hip): hand the search state to k_climb, launch one
// sweep segment, replay the moves it reports onto the topology mirror.
//
// What crosses the boundary: the tree as back links in compact vector ids, the sweep's visiting order (nodep[] after
// nodeRectifierPars, reference sprparsimony.cpp:2046-2101, :3297), bestParsimony / randomMP / bestIterationScoreHits /
// bestTreeScoreHits / insertNode / removeNode and the state of the lcg64 tie stream (:2168-2176, :3306-3311).  The
// directional vectors stay where they are; the kernel treats them all as stale at launch and keeps its own per-tile
// subtree scores, the host forgets its validity bookkeeping afterwards -- both sides only ever trust what they computed.
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <thread>

#include "../csrc/engine.hpp"
#include "../csrc/grow.hpp"

namespace mpf {

#define HIPCHK(expr)                                                                              \
  do {                                                                                            \
    hipError_t e__ = (expr);                                                                      \
    if (e__ != hipSuccess) {                                                                      \
      set_error(std::string(#expr) + ": " + hipGetErrorString(e__) + " (" + __FILE__ + ":" +      \
                std::to_string(__LINE__) + ")");                                                  \
      return MPF_E_HIP;                                                                           \
    }                                                                                             \
  } while (0)

// k_climb's workgroups wait for each other, so all of them must be resident at once.  Engines on several host threads of one
// process share a device: launches are admitted only while their workgroups fit the chip together (a launch that still fails
// to become resident -- other processes -- gives up inside the kernel after 30 ms and the caller falls back to host batches).
namespace {
struct ClimbGate {
  std::mutex m;
  std::condition_variable cv;
  int used = 0, cus = 0;
};
ClimbGate g_gate[64];
struct GateHold {
  ClimbGate *g = nullptr;
  int n = 0;
  ~GateHold() { release(); }
  void release()
  {
    if (!g) return;
    { std::lock_guard<std::mutex> lk(g->m); g->used -= n; }
    g->cv.notify_all();
    g = nullptr;
  }
};
}  // namespace

// k_climb's workgroups wait for each other: a launch whose tiles cannot all be resident would spin until its start barrier times
// out, every time.  The tile width the launch will use: option "climb_tile", widened (DNA: 2, 4, 8 words per lane group) until the
// workgroups fit 85 % of the CUs (several per CU where the control state is small); 0 = this alignment is too long for the kernel.
int Engine::climb_fit_vw(bool one_workgroup)
{
  if (one_workgroup && g_.S == 4 && !climb_vw_set_) {
    // a climb as ONE workgroup (k_climb_many): its waves take whole tiles through the refresh and share out the scans of all tiles.
    // 64-word tiles first (eight waves; 128-word tiles leave room for four: measured at C3 97 against 81 climbs/s), then 128, 32, 16 --
    // whichever keeps the padding within 8 % of the narrowest's (C2, 313 words: 5 x 64; C3, 1563: 25 x 64)
    size_t least = ~(size_t)0;
    for (int vw = 1; vw <= 8; vw *= 2) least = std::min(least, (size_t)climb_tiles(g_, vw) * 16 * (size_t)vw);
    for (int vw : {4, 8, 2, 1}) {
      if (climb_lds_bytes(g_, n_, vw, many_batch_max(), true, vw == 4 && many_word_major_) > 160 * 1024) continue;
      if ((size_t)climb_tiles(g_, vw) * 16 * (size_t)vw * 100 <= least * 108) return vw;
    }
    return 0;
  }
  if (climb_cus_ <= 0) {
    int c = 0;
    if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev_) != hipSuccess || c <= 0) c = 256;
    climb_cus_ = c;
  }
  const int cap = std::max(1, climb_cus_ * 85 / 100);
  for (int vw = (g_.S == 4) ? std::max(1, climb_vw_) : 1; vw <= ((g_.S == 4) ? 8 : 1); vw *= 2) {
    const size_t lds = climb_lds_bytes(g_, n_, vw, climb_batch_bound(climb_stop_len_ != 0));
    if (lds > 160 * 1024) continue;
    const int per_cu = (int)std::max<size_t>(1, (160 * 1024) / std::max<size_t>(lds, 1));
    const int tiles = climb_tiles(g_, vw);
    const int wgs = climb_groups_ > 0 ? std::min(climb_groups_, tiles) : tiles;        // (fewer workgroups than tiles: each works through several)
    if ((wgs + per_cu - 1) / per_cu <= cap) return vw;
  }
  return 0;
}

// One launch of the climb kernel in three parts: climb_prepare lays the segment out (staging, parameters) and enqueues its uploads,
// the caller launches -- k_climb for this engine alone (climb_segment), or k_climb_many for a batch of engines (Engine::climb_many) --
// and copies the result block back, climb_harvest takes the moves over.
int Engine::climb_prepare(int maxtrav_eff, int total, int i, uint32_t randomMP, unsigned iter_hits, bool may_idle, int force_groups,
                          hipStream_t st, ClimbParams &p, int *vw_out, int *tiles_out, bool sweeps_inside, uint32_t start_mp)
{
  const int vw = climb_fit_vw(force_groups == 1);
  *vw_out = vw;
  if (vw <= 0) { set_error("device climb: the alignment's tiles do not fit the chip"); return MPF_E_STATE; }
  const int tiles = climb_tiles(g_, vw);
  *tiles_out = tiles;
  const size_t ns = nslots_;
  const size_t hdr_words = (sizeof(ClimbHeader) + 3) / 4;
  // (a launch that runs every sweep of a climb makes more moves than one sweep's: room for four sweeps' worth, then it hands back)
  // (option many_moves_cap, tests: a short list, so that launches end in the middle of sweeps the device started)
  const size_t max_moves = sweeps_inside ? (many_moves_cap_ > 0 ? std::min<size_t>((size_t)many_moves_cap_, 4 * (size_t)total) : 4 * (size_t)total) : (size_t)total;
  cd_.max_moves = (uint32_t)max_moves;
  const size_t out_words = hdr_words + 3 * max_moves;
  HIPCHK(cd_.bk.reserve(ns));
  HIPCHK(cd_.order.reserve((size_t)total));
  // (k_climb_many's word-major shape keeps a score per word: 64 per vector and tile)
  HIPCHK(cd_.sct.reserve((size_t)tiles * ns * ((force_groups == 1 || climb_word_major_) ? 64 : 16)));
  // exchange ring | per-XCD level-1 words | per-XCD workgroup counts
  const size_t gsum_words = 3 * (size_t)kClimbCap + 8 * 3 * (size_t)kClimbCap + 8;
  HIPCHK(cd_.gsum.reserve(gsum_words));
  HIPCHK(cd_.out.reserve(out_words));
  HIPCHK(cd_.h_bk.reserve(ns));
  HIPCHK(cd_.h_order.reserve((size_t)total));
  HIPCHK(cd_.h_out.reserve(out_words));
  for (size_t c = 0; c < ns; c++) {
    // (the store has room for node 2n - 1, which an unrooted tree does not use: such a record points at itself)
    const int b = back_[(size_t)rec_of((uint32_t)c)];
    cd_.h_bk.p[c] = b < 0 ? (uint16_t)c : (uint16_t)slot(b);
  }
  for (int k = 1; k <= total; k++) cd_.h_order.p[k - 1] = (uint16_t)slot(nodep_[(size_t)k]);
  ClimbHeader h;
  std::memset(&h, 0, sizeof(h));
  h.rng = rng_.state;
  h.hits = hits_;
  h.best = best_;
  h.randomMP = randomMP;
  h.iter_hits = iter_hits;
  h.pos = (uint32_t)i;
  h.insert_cid = insert_rec_ >= 0 ? (int32_t)slot(insert_rec_) : -1;
  h.remove_cid = remove_rec_ >= 0 ? (int32_t)slot(remove_rec_) : -1;
  h.since_move = 0;
  h.batch = 0;
  h.start_mp = start_mp;
  std::memcpy(cd_.h_out.p, &h, sizeof(h));
  p.vec = d_vec_;
  p.n = (uint32_t)n_;
  p.nslots = (uint32_t)ns;
  p.Wp = (uint32_t)g_.Wp;
  p.tiles = (uint32_t)tiles;
  p.total = (uint32_t)total;
  p.maxtrav = (uint32_t)maxtrav_eff;
  p.tie_mode = (uint32_t)tie_mode_;
  p.idle_limit = may_idle ? (uint32_t)climb_idle_ : 0u;
  p.max_moves = (uint32_t)max_moves;
  // (a climb that is ONE workgroup pays for every speculative prune node with its own arithmetic -- a step there is bound by its
  //  candidates, not by its round trips --: one prune node behind a move unless the option says otherwise; C3 153 -> 162 climbs/s)
  p.batch_min = (force_groups == 1 && !climb_batch_min_set_) ? 1u : (uint32_t)std::max(1, std::min(climb_batch_min_, 16));
  // (a climb under a stop length is a later iteration of a -bb search: it starts near an optimum, its moves are some 25-45 prune
  //  nodes apart -- sixteen prune nodes per step there)
  p.batch_max = (uint32_t)climb_batch_bound(climb_stop_len_ != 0);
  p.order = cd_.order.p;
  p.bk = cd_.bk.p;
  p.sct = cd_.sct.p;
  p.gsum = cd_.gsum.p;
  p.xsum = cd_.gsum.p + 3 * (size_t)kClimbCap;
  p.xcnt = reinterpret_cast<uint32_t *>(cd_.gsum.p + 27 * (size_t)kClimbCap);
  p.hdr = reinterpret_cast<ClimbHeader *>(cd_.out.p);
  p.moves = cd_.out.p + hdr_words;
  p.trace = nullptr;
  p.trace_cap = 0;
  p.beat = nullptr;
  p.near_q = (force_groups == 1 && !climb_near_q_set_) ? 1u : (uint32_t)climb_near_q_;       // (one workgroup per climb: speculation is paid for -- 164 -> 171 climbs/s at C3)
  p.fault = (uint32_t)climb_fault_;
  climb_fault_ = 0;                                  // (one launch)
  p.stop_len = climb_stop_len_;
  // option "climb_groups": 0 = a workgroup per tile; k = at most k workgroups, each working through its share of the tiles
  const int groups = force_groups > 0 ? std::min(force_groups, tiles) : climb_groups_ > 0 ? std::min(climb_groups_, tiles) : tiles;
  p.groups = (uint32_t)groups;
  p.sweeps_inside = sweeps_inside ? 1u : 0u;
  p.snap = nullptr;
  p.snap_r = nullptr;
  if (groups < tiles) {
    HIPCHK(cd_.snap.reserve((size_t)groups * (ns + ns / 4 + 1)));
    HIPCHK(cd_.snap_r.reserve((size_t)groups * ns));
    p.snap = cd_.snap.p;
    p.snap_r = cd_.snap_r.p;
  }
  if (climb_trace_) {
    HIPCHK(cd_.h_beat.reserve(32));
    std::memset(cd_.h_beat.p, 0, 32 * sizeof(uint32_t));
    p.beat = cd_.h_beat.p;
    const size_t cap = 1u << 16;
    HIPCHK(cd_.trace.reserve(8 * cap));
    p.trace = cd_.trace.p;
    p.trace_cap = (uint32_t)cap;
  }
  HIPCHK(hipMemcpyAsync(cd_.bk.p, cd_.h_bk.p, ns * sizeof(uint16_t), hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(cd_.order.p, cd_.h_order.p, (size_t)total * sizeof(uint16_t), hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(cd_.out.p, cd_.h_out.p, sizeof(h), hipMemcpyHostToDevice, st));
  HIPCHK(hipMemsetAsync(cd_.gsum.p, 0, gsum_words * sizeof(unsigned long long), st));
  shadow_ok_ = false;                              // (k_climb rewrites vectors in the row-major store only)
  return MPF_OK;
}

int Engine::climb_segment(int maxtrav_eff, int total, int *i, uint32_t *randomMP, unsigned *iter_hits, bool may_idle,
                          uint32_t *reason, uint32_t *n_moves)
{
  const auto t0 = std::chrono::steady_clock::now();
  ClimbParams p;
  int vw = 0, tiles = 0;
  {
    const int rc = climb_prepare(maxtrav_eff, total, *i, *randomMP, *iter_hits, may_idle, 0, st_, p, &vw, &tiles, false, 0);
    if (rc) return rc;
  }
  const int groups = (int)p.groups;
  const size_t hdr_words = (sizeof(ClimbHeader) + 3) / 4;
  const size_t out_words = hdr_words + 3 * (size_t)total;
  GateHold hold;
  {
    ClimbGate &g = g_gate[dev_ & 63];
    const size_t lds = climb_lds_bytes(g_, n_, vw, (int)p.batch_max, false, vw == 4 && climb_word_major_ && g_.S == 4);
    const int per_cu = (int)std::max<size_t>(1, (160 * 1024) / std::max<size_t>(lds, 1));
    std::unique_lock<std::mutex> lk(g.m);
    if (!g.cus) {
      int c = 0;
      if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev_) != hipSuccess || c <= 0) c = 256;
      g.cus = c;
    }
    // (counted in CU-equivalents of this launch's footprint: tiles / per_cu CUs)
    const int need = (groups + per_cu - 1) / per_cu;
    // (not every CU takes a workgroup of this size at every moment -- other queues' kernels come and go -- so admission
    //  stops at 85 % of the chip: a launch beyond that waits its turn here instead of timing out in the kernel)
    const int cap = std::max(1, g.cus * 85 / 100);
    g.cv.wait(lk, [&] { return g.used == 0 || g.used + need <= cap; });
    g.used += need;
    hold.g = &g;
    hold.n = need;
  }
  HIPCHK(launch_climb(st_, g_, vw, p, climb_word_major_));
  HIPCHK(hipMemcpyAsync(cd_.h_out.p, cd_.out.p, out_words * sizeof(uint32_t), hipMemcpyDeviceToHost, st_));
  {
    // a sweep segment takes milliseconds; a launch that is still running after many seconds will not come back (every wait
    // inside the kernel is bounded) -- report that instead of blocking the caller for ever
    const auto w0 = std::chrono::steady_clock::now();
    hipError_t q;
    long spins = 0;
    WaitScope waiting;
    while ((q = hipStreamQuery(st_)) == hipErrorNotReady) {
      // (a launch runs for milliseconds: the waiting thread sleeps between its looks -- a core per waiting engine is what a
      //  container's CPU quota cannot afford when many engines share the GPU, and the HIP runtime's own threads need theirs)
      { struct timespec ts = {0, 30000}; nanosleep(&ts, nullptr); }
      if ((++spins & 15) == 0) {
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count() > (climb_trace_ ? 4.0 : 20.0)) {
          std::string where;
          if (p.beat) for (int k = 0; k < 32; k++) where += " " + std::to_string(cd_.h_beat.p[k]);
          set_error("device climb: the launch did not finish within 20 s" + (where.empty() ? std::string() : " (step phase pos B ncand nops chains done err:" + where + ")"));
          // (the kernel may still be writing vectors: nothing this engine holds can be trusted any more)
          invalidate_all();
          broken_ = true;
          return MPF_E_STATE;
        }
        wait_pause();
      }
    }
    if (q != hipSuccess) { invalidate_all(); set_error(std::string("device climb: ") + hipGetErrorString(q)); return MPF_E_HIP; }
  }
  hold.release();
  return climb_harvest(total, tiles, t0, i, randomMP, iter_hits, reason, n_moves);
}

int Engine::climb_harvest(int total, int tiles, std::chrono::steady_clock::time_point t0, int *i, uint32_t *randomMP, unsigned *iter_hits,
                          uint32_t *reason, uint32_t *n_moves)
{
  const size_t hdr_words = (sizeof(ClimbHeader) + 3) / 4;
  ClimbHeader h;
  std::memcpy(&h, cd_.h_out.p, sizeof(h));
  *reason = h.reason;
  *n_moves = 0;
  stats.climb_launches++;
  if (h.reason == CLIMB_ABORT) {
    stats.climb_ms_total += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return MPF_OK;                                // nothing was changed: the caller goes on with host-driven batches
  }
  if (h.reason == CLIMB_ERROR && h.err == 1u) {
    // the launch's workgroups lost each other (a chip shared with other processes' persistent kernels: climb.hip, exchange): the
    // moves it may have made were never taken over -- topology mirror, tie stream and counters are as they were before the
    // launch --, only the vectors in HBM were rewritten.  Forget those, and let the caller run the segment as host-driven batches.
    invalidate_all();
    stats.climb_ms_total += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    *reason = CLIMB_ABORT;
    return MPF_OK;
  }
  if (h.reason == CLIMB_ERROR || h.reason == CLIMB_RUNNING || h.n_moves > cd_.max_moves) {
    invalidate_all();                             // (the kernel has rewritten vectors for topologies the mirror never saw)
    set_error("device climb: internal error " + std::to_string(h.err) + " (reason " + std::to_string(h.reason) + "; waiter " +
              std::to_string(h.pad2[0] & 0x7FFFFFFFu) + " of " + std::to_string(tiles) + " tiles, candidate " + std::to_string(h.pad2[1] & 0xFFFFu) + " of " +
              std::to_string(h.pad2[1] >> 16) + ", arrivals seen " + std::to_string(h.pad2[2] & 0xFFFFu) + ", exchange " + std::to_string(h.pad2[2] >> 16) +
              ", steps " + std::to_string(h.steps) + ")");
    return MPF_E_STATE;
  }
  if (climb_trace_) {
    const size_t nrec = std::min<size_t>(h.pad[0], 1u << 16);
    const size_t at = cd_.h_trace.size();
    cd_.h_trace.resize(at + 8 * nrec);
    if (nrec) HIPCHK(hipMemcpy(cd_.h_trace.data() + at, cd_.trace.p, 8 * nrec * sizeof(uint32_t), hipMemcpyDeviceToHost));
    cd_.trace_records += nrec;
  }
  // replay: restoreTreeRearrangeParsimony on the mirror, link by link as apply_move does
  const uint32_t *mv = cd_.h_out.p + hdr_words;
  for (uint32_t m = 0; m < h.n_moves; m++) {
    const int pr = rec_of(mv[3 * m]), qr = rec_of(mv[3 * m + 1]);
    moves_.push_back(Move{pr, qr, mv[3 * m + 2]});
    const int a = back_[nx(pr)], b = back_[nx(nx(pr))];
    hookup(a, b);
    const int r = back_[qr];
    hookup(nx(pr), qr);
    hookup(nx(nx(pr)), r);
    stats.moves_applied++;
  }
  if (h.sweeps) {
    // (ClimbParams::sweeps_inside) nodeRectifierPars ran on the device: nodep[] as its last run left it
    for (int k = n_ + 1; k <= total; k++) nodep_[(size_t)k] = rec_of((uint32_t)cd_.h_order.p[k - 1]);
  }
  many_.startMP = h.start_mp;
  invalidate_all();
  rng_.state = h.rng;
  hits_ = (unsigned long)h.hits;
  best_ = h.best;
  *randomMP = h.randomMP;
  *iter_hits = h.iter_hits;
  *i = (int)h.pos;
  insert_rec_ = h.insert_cid >= 0 ? rec_of((uint32_t)h.insert_cid) : -1;
  remove_rec_ = h.remove_cid >= 0 ? rec_of((uint32_t)h.remove_cid) : -1;
  *n_moves = h.n_moves;
  stats.insertion_tests += h.n_tests;
  stats.algorithmic_bytes += h.n_tests * 6u * (uint64_t)g_.S * (uint64_t)Wref_ * 4u;
  for (int k = 0; k < 7; k++) climb_phase_ticks_[k] += h.tph[k];
  for (int k = 8; k < 16; k++) climb_phase_ticks_[k] += h.tph[k];
  climb_phase_ticks_[7] = h.tph[7] * 100ull;     // (kHz of the last launch; read back as "us" / 100)
  climb_ctr_[0] += h.n_ops; climb_ctr_[1] += h.pad[1]; climb_ctr_[2] += h.pad[2]; climb_ctr_[3] += h.pad[3];
  stats.climb_steps += h.steps;
  stats.climb_nodes += h.n_scanned_nodes;
  stats.climb_moves += h.n_moves;
  stats.climb_ms_total += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return MPF_OK;
}

// ---- many independent climbs in one launch -----------------------------------------------------------------------------------
// pllOptimizeSprParsimony (reference sprparsimony.cpp:3244-3319) on n engines at once -- random restarts, the bootstrap samples'
// refinement climbs (iqtree.cpp:2797-2862): climbs that have nothing to do with each other.  One engine alone gets through its
// chain of dependent steps fastest on a workgroup per tile (98 at C3) -- and leaves most of the chip idle doing so; host threads
// with an engine each fill some of it (bench: concurrent_climbs), but every persistent launch holds a hardware queue and its CUs
// while it waits.  Here every climb is ONE workgroup that works through all tiles itself (ClimbParams::groups == 1: nothing crosses
// between workgroups) and through all its sweeps (ClimbParams::sweeps_inside: nodeRectifierPars on the device), and ONE launch of
// k_climb_many holds a workgroup per climb.  A launch ends when every climb is at its optimum or has filled its move list (four
// sweeps' worth); the host replays the moves on each engine's topology mirror and, for the rare climb that is not through, launches
// again from the middle of the sweep the device was in.  Each climb: the same moves, draws and final tree as its solo
// mpf_optimize_spr (tests/test_gpu_climb_many.py; at C3 size against the oracle-pinned climb: tests/test_gpu_configs.py).
int Engine::climb_many(Engine **engs, int n, int mintrav, int maxtrav, uint32_t *scores)
{
  if (n <= 0) return MPF_OK;
  if (!engs || !scores) { set_error("mpf_optimize_spr_many: null argument"); return MPF_E_INVALID; }
  std::vector<uint8_t> state((size_t)n, 1);         // 1 = a climb starts on this engine
  for (;;) {
    const int rc = climb_many_round(engs, n, mintrav, maxtrav, state.data(), scores);
    if (rc) return rc;
    bool any = false;
    for (int k = 0; k < n; k++) any = any || state[(size_t)k] == 2;
    if (!any) return MPF_OK;
  }
}

// One launch: state[k] in: 0 = engine k takes no part, 1 = its climb STARTS now (tree, weights, tie stream set as for mpf_optimize_spr),
// 2 = its climb goes on; out: 2 = not at its optimum yet, 0 = done (scores[k] = the final length).  A caller that has more climbs than
// engines hands a finished engine its next tree between two launches (mpboot_amd/bootstrap.py: refine_boot_trees).
int Engine::climb_many_round(Engine **engs, int n, int mintrav, int maxtrav, uint8_t *state, uint32_t *scores)
{
  if (n <= 0) return MPF_OK;
  if (!engs || !scores || !state) { set_error("mpf_optimize_spr_many: null argument"); return MPF_E_INVALID; }
  int first = -1;
  for (int k = 0; k < n; k++) if (state[k]) { first = k; break; }
  if (first < 0) return MPF_OK;
  Engine &e0 = *engs[first];
  const int vw0 = e0.climb_fit_vw(true);
  // a starting climb's preamble (the full evaluate of :3277) on its own engine; what does not fit the batch runs alone, here
  std::vector<int> starting;
  for (int k = 0; k < n; k++) {
    if (state[k] != 1) continue;
    Engine &e = *engs[k];
    if (!e.have_tree_) { set_error("no tree set"); return MPF_E_STATE; }
    const int mt_eff = std::min(maxtrav, e.n_ - 3);
    const bool fits = vw0 > 0 && e.dev_ == e0.dev_ && e.g_.S == e0.g_.S && !e.sankoff_ && !e.rand_fn_ && !(e.ufb_ && !e.ufb_->suspended) && mintrav == 1 &&
                      e.max_visits_ == 0 && e.scan_mode_ == 1 && e.climb_device_ > 0 && climb_supported(e.g_, e.n_, mt_eff, e.many_batch_max()) && e.climb_fit_vw(true) == vw0;
    if (!fits) {
      const int rc = e.optimize_spr(mintrav, maxtrav, &scores[k]);
      if (rc) return rc;
      state[k] = 0;
      continue;
    }
    starting.push_back(k);
  }
  {
    auto preamble = [&](int k) -> int {
      Engine &e = *engs[k];
      e.moves_.clear();
      e.node_rectifier();
      uint32_t len = 0;
      e.invalidate_vectors();
      const int rc = e.tree_length(&len);
      if (rc) return rc;
      e.best_ = len;
      e.ntips_ = e.n_;
      e.insert_rec_ = e.remove_rec_ = -1;
      e.visits_done_ = 0;
      e.many_ = ManyState{};
      e.many_.randomMP = len;
      return MPF_OK;
    };
    // (an evaluate per engine, each with its own stream and its own wait: a quarter of a millisecond one after the other --
    //  a third of the whole call at 512 climbs of C2 --, so a handful of threads share them)
    const int nthr = starting.size() >= 16 ? (int)std::min<size_t>(8, std::max(1u, std::thread::hardware_concurrency())) : 1;
    if (nthr <= 1) {
      for (int k : starting) { const int rc = preamble(k); if (rc) return rc; }
    } else {
      std::vector<std::thread> th;
      std::vector<int> rcs((size_t)nthr, MPF_OK);
      std::vector<std::string> errs((size_t)nthr);
      for (int t = 0; t < nthr; t++)
        th.emplace_back([&, t] {
          if (hipSetDevice(e0.dev_) != hipSuccess) { rcs[(size_t)t] = MPF_E_HIP; errs[(size_t)t] = "hipSetDevice failed"; return; }
          for (size_t i = (size_t)t; i < starting.size(); i += (size_t)nthr) {
            const int rc = preamble(starting[i]);
            if (rc) { rcs[(size_t)t] = rc; errs[(size_t)t] = last_error(); return; }
          }
        });
      for (auto &x : th) x.join();
      for (int t = 0; t < nthr; t++) if (rcs[(size_t)t]) { set_error(errs[(size_t)t]); return rcs[(size_t)t]; }
    }
    for (int k : starting) state[k] = 2;
  }
  HIPCHK(hipSetDevice(e0.dev_));
  ManyBufs &mb = e0.many_bufs_;
  HIPCHK(mb.d_params.reserve((size_t)n));
  HIPCHK(mb.h_params.reserve((size_t)n));
  std::vector<int> batch;
  const size_t hdr_words = (sizeof(ClimbHeader) + 3) / 4;
  uint32_t max_ns = 0;
  const auto t0 = std::chrono::steady_clock::now();
  for (int k = 0; k < n; k++) {
    if (state[k] != 2) continue;
    Engine &e = *engs[k];
    ManyState &s = e.many_;
    const int total = 2 * e.n_ - 2;
    if (!s.in_sweep) { s.startMP = s.randomMP; e.node_rectifier(); s.i = 1; s.in_sweep = true; }
    int vw = 0;
    const int rc = e.climb_prepare(std::min(maxtrav, e.n_ - 3), total, s.i, s.randomMP, s.iter_hits, false, 1, e0.st_, mb.h_params.p[batch.size()], &vw, &s.tiles,
                                   e0.many_sweeps_inside_, s.startMP);
    if (rc) return rc;
    max_ns = std::max(max_ns, (uint32_t)e.nslots_);
    batch.push_back(k);
  }
  if (batch.empty()) return MPF_OK;
  const auto t_prep = std::chrono::steady_clock::now();
  HIPCHK(hipMemcpyAsync(mb.d_params.p, mb.h_params.p, batch.size() * sizeof(ClimbParams), hipMemcpyHostToDevice, e0.st_));
  const bool wm = e0.g_.S == 4 && vw0 == 4 && e0.many_word_major_;
  size_t lds = 0;
  for (size_t b = 0; b < batch.size(); b++) {
    Engine &e = *engs[batch[b]];
    lds = std::max(lds, climb_lds_bytes(e.g_, e.n_, vw0, (int)mb.h_params.p[b].batch_max, true, wm));
  }
  HIPCHK(launch_climb_many(e0.st_, e0.g_, vw0, mb.d_params.p, (int)batch.size(), lds, wm));
  for (int k : batch) {
    Engine &e = *engs[k];
    const size_t out_words = hdr_words + 3 * (size_t)e.cd_.max_moves;
    HIPCHK(hipMemcpyAsync(e.cd_.h_out.p, e.cd_.out.p, out_words * sizeof(uint32_t), hipMemcpyDeviceToHost, e0.st_));
    if (e0.many_sweeps_inside_) HIPCHK(hipMemcpyAsync(e.cd_.h_order.p, e.cd_.order.p, (size_t)(2 * e.n_ - 2) * sizeof(uint16_t), hipMemcpyDeviceToHost, e0.st_));
  }
  {
    // (bounded like every wait on this kernel: climb_segment)
    const auto w0 = std::chrono::steady_clock::now();
    hipError_t q;
    long spins = 0;
    while ((q = hipStreamQuery(e0.st_)) == hipErrorNotReady) {
      { struct timespec ts = {0, 30000}; nanosleep(&ts, nullptr); }
      if ((++spins & 1023) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count() > 120.0) {
        for (int k : batch) { engs[k]->invalidate_all(); engs[k]->broken_ = true; }
        set_error("mpf_optimize_spr_many: the launch did not finish within 120 s");
        return MPF_E_STATE;
      }
    }
    if (q != hipSuccess) { for (int k : batch) engs[k]->invalidate_all(); set_error(std::string("mpf_optimize_spr_many: ") + hipGetErrorString(q)); return MPF_E_HIP; }
  }
  const auto t_dev = std::chrono::steady_clock::now();
  for (int k : batch) {
    Engine &e = *engs[k];
    ManyState &s = e.many_;
    const int total = 2 * e.n_ - 2;
    uint32_t reason = 0, nm = 0;
    int rc = e.climb_harvest(total, s.tiles, t0, &s.i, &s.randomMP, &s.iter_hits, &reason, &nm);
    if (!rc && reason == CLIMB_ABORT) { set_error("mpf_optimize_spr_many: a single-workgroup climb reported an abort"); rc = MPF_E_STATE; }
    if (rc) {
      // the call fails as a whole.  The launch has rewritten vectors of EVERY engine of the batch; those not taken over yet keep the
      // tree and the stream they had before it -- and must not trust a vector (their state says "goes on": the caller starts over)
      const std::string msg = last_error();
      for (int k2 : batch) if (engs[k2] != &e) engs[k2]->invalidate_all();
      set_error(msg);
      return rc;
    }
    if (s.i > total) {                              // the sweep is through (:3316)
      s.in_sweep = false;
      if (!(s.randomMP < s.startMP)) {
        e.climb_finished(total);
        scores[k] = s.randomMP;
        state[k] = 0;
      }
    }
  }
  if (getenv("MPF_MANY_TRACE")) {
    const auto t_end = std::chrono::steady_clock::now();
    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    std::fprintf(stderr, "[many] round of %zu climbs: prepare %.2f ms, launch + wait %.2f ms, harvest %.2f ms\n", batch.size(), ms(t0, t_prep), ms(t_prep, t_dev), ms(t_dev, t_end));
  }
  return MPF_OK;
}

// ---- device-resident stepwise addition (csrc/grow.hip) ---------------------------------------------------------------------
// The tree built so far goes down as a ROOTED tree hung from the start tip: per node its parent, its two children in the order
// stepwiseAddition visits them (q->next->back, then q->next->next->back, reference sprparsimony.cpp:3015-3016), pre-order position,
// subtree size, depth and the vector id of its "down" record.  The kernel adds the taxa perm[ntips_ + 1 ..] and reports, per
// step, the branch it chose and the length of the tree; the insertions are replayed here on the topology mirror exactly as
// Engine::addition_phase applies them (:3158-3171).  *done = false: the kernel could not be used or did not come back clean --
// nothing of the engine's state has been touched except the vectors, the caller's own loop builds the tree.
int Engine::grow_fit_vw() const
{
  if (g_.S != 4) return 1;
  if (grow_vw_ > 0) return grow_vw_;
  // (option grow_tile 0: the word-major copy, DNA below 2 GiB -- 64-word tiles, a lane = one word with its four states: half the
  //  vector instructions and contiguous kilobytes, but measured slower than the 64-word quad tiles at C3: 45 against 56 trees/s)
  if (g_.shoff && grow_vw_ == 0) return 0;
  for (int vw = 1; vw <= 8; vw *= 2)
    if (grow_tiles(g_, vw) <= 32) return vw;
  return 8;
}

int Engine::grow_segment(const std::vector<int> &perm, uint32_t len0, uint32_t *best_per_step, int32_t *insert_per_step, bool *done)
{
  *done = false;
  const int n = n_;
  const int steps = n - ntips_;
  if (steps <= 0) return MPF_OK;
  const int vw = grow_fit_vw();
  const int tiles = grow_tiles(g_, vw), waves = grow_waves(g_, vw);
  const size_t N2 = 2 * (size_t)n;
  const auto t0 = std::chrono::steady_clock::now();
  // ---- the rooted start tree
  std::vector<uint16_t> init(8 * N2, (uint16_t)kGrowNone);
  uint16_t *par = init.data(), *ch1 = par + N2, *ch2 = ch1 + N2, *pos = ch2 + N2, *sz = pos + N2, *dep = sz + N2, *dc = dep + N2, *ord = dc + N2;
  auto node_of = [&](int r) -> uint32_t { const int v = num(r); return v <= n ? (uint32_t)(v - 1) : (uint32_t)(n + (v - n - 1)); };
  const int f = start_;
  uint32_t m0 = 0;
  {
    struct Fr { int rec; uint32_t parent; uint32_t depth; };
    std::vector<Fr> st;
    st.push_back(Fr{back_[f], kGrowNone, 0u});
    while (!st.empty()) {
      const Fr fr = st.back();
      st.pop_back();
      const int c = fr.rec;
      const uint32_t u = node_of(c);
      par[u] = (uint16_t)fr.parent; dep[u] = (uint16_t)fr.depth; pos[u] = (uint16_t)m0; ord[m0++] = (uint16_t)u;
      dc[u] = (uint16_t)slot(c);
      sz[u] = 1;
      if (!tip(c)) {
        const int c1 = back_[nx(c)], c2 = back_[nx(nx(c))];
        ch1[u] = (uint16_t)node_of(c1); ch2[u] = (uint16_t)node_of(c2);
        st.push_back(Fr{c2, u, fr.depth + 1u});
        st.push_back(Fr{c1, u, fr.depth + 1u});
      }
    }
    for (int i = (int)m0 - 1; i >= 0; i--) {
      const uint32_t u = ord[i];
      if (ch1[u] != kGrowNone) sz[u] = (uint16_t)(1u + sz[ch1[u]] + sz[ch2[u]]);
    }
  }
  if (m0 != (uint32_t)(2 * ntips_ - 3)) { set_error("device addition: the start tree is not a binary tree below the start tip"); return MPF_E_STATE; }
  // per step: the tip, and the inner node the reference takes for it -- tr->nodep[nextnode++], whichever node and record that
  // is after earlier nodeRectifierPars calls; q->next->next becomes its down record
  std::vector<uint16_t> tips(3 * (size_t)steps);
  for (int s = 0; s < steps; s++) {
    tips[(size_t)s] = (uint16_t)slot(nodep_[(size_t)perm[(size_t)(ntips_ + 1 + s)]]);
    const int q = nodep_[(size_t)(nextnode_ + s)];
    if (tip(q)) { set_error("device addition: nodep names a tip where an inner node is due"); return MPF_E_STATE; }
    tips[(size_t)steps + (size_t)s] = (uint16_t)node_of(q);
    tips[2 * (size_t)steps + (size_t)s] = (uint16_t)slot(nx(nx(q)));
  }
  // ---- buffers
  const size_t hdr_words = (sizeof(GrowHeader) + 3) / 4, out_words = hdr_words + 2 * (size_t)steps;
  const size_t xstride = (size_t)n + N2 / 32 + 8;
  const size_t slot_words = grow_vec_words(g_, vw);
  HIPCHK(gd_.init.reserve(8 * N2 + 3 * (size_t)steps));
  HIPCHK(gd_.xrow.reserve(2 * (size_t)tiles * xstride));
  HIPCHK(gd_.park.reserve((size_t)tiles * ((size_t)waves * kGrowParkPart + kGrowParkSkel) * slot_words));
  HIPCHK(gd_.ucp.reserve((size_t)tiles * kGrowMaxParts * slot_words));
  HIPCHK(gd_.out.reserve(out_words));
  HIPCHK(gd_.h_init.reserve(8 * N2 + 3 * (size_t)steps));
  HIPCHK(gd_.h_out.reserve(out_words));
  std::memcpy(gd_.h_init.p, init.data(), 8 * N2 * sizeof(uint16_t));
  std::memcpy(gd_.h_init.p + 8 * N2, tips.data(), 3 * (size_t)steps * sizeof(uint16_t));
  GrowHeader h;
  std::memset(&h, 0, sizeof(h));
  h.rng = rng_.state;
  std::memcpy(gd_.h_out.p, &h, sizeof(h));
  GrowParams p;
  p.vec = vw == 0 ? d_vec_ + g_.shoff : d_vec_;
  p.n = (uint32_t)n; p.nslots = (uint32_t)nslots_; p.Wp = (uint32_t)g_.Wp; p.tiles = (uint32_t)tiles;
  p.m0 = m0; p.steps = (uint32_t)steps; p.tie_mode = (uint32_t)tie_mode_; p.len0 = len0;
  p.root_cid = slot(f); p.root_node = node_of(back_[f]);
  p.init = gd_.init.p; p.tips = gd_.init.p + 8 * N2; p.xnode = p.tips + steps; p.xdcid = p.tips + 2 * (size_t)steps;
  p.xrow = gd_.xrow.p; p.xstride = (uint32_t)xstride;
  p.park = gd_.park.p; p.ucp = gd_.ucp.p;
  p.hdr = reinterpret_cast<GrowHeader *>(gd_.out.p);
  p.out = gd_.out.p + hdr_words;
  p.fault = (uint32_t)grow_fault_;
  grow_fault_ = 0;
  GateHold hold;
  {
    ClimbGate &g = g_gate[dev_ & 63];
    const int per_cu = std::max(1, grow_blocks_per_cu(g_, n, vw));
    std::unique_lock<std::mutex> lk(g.m);
    if (!g.cus) {
      int c = 0;
      if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev_) != hipSuccess || c <= 0) c = 256;
      g.cus = c;
    }
    const int need = (tiles + per_cu - 1) / per_cu;
    const int cap = std::max(1, g.cus * 85 / 100);
    g.cv.wait(lk, [&] { return g.used == 0 || g.used + need <= cap; });
    g.used += need;
    hold.g = &g;
    hold.n = need;
  }
  HIPCHK(hipMemcpyAsync(gd_.init.p, gd_.h_init.p, (8 * N2 + 3 * (size_t)steps) * sizeof(uint16_t), hipMemcpyHostToDevice, st_));
  HIPCHK(hipMemcpyAsync(gd_.out.p, gd_.h_out.p, sizeof(h), hipMemcpyHostToDevice, st_));
  shadow_ok_ = false;                              // (k_grow writes vectors in the row-major store only)
  HIPCHK(launch_grow(st_, g_, vw, p));
  HIPCHK(hipMemcpyAsync(gd_.h_out.p, gd_.out.p, out_words * sizeof(uint32_t), hipMemcpyDeviceToHost, st_));
  {
    const auto w0 = std::chrono::steady_clock::now();
    hipError_t q;
    long spins = 0;
    WaitScope waiting;
    while ((q = hipStreamQuery(st_)) == hipErrorNotReady) {
      { struct timespec ts = {0, 50000}; nanosleep(&ts, nullptr); }      // (a tree takes tens of milliseconds: sleep between looks, as climb_segment)
      if ((++spins & 15) == 0) {
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count() > 30.0) {
          set_error("device addition: the launch did not finish within 30 s");
          invalidate_all();
          broken_ = true;
          return MPF_E_STATE;
        }
        wait_pause();
      }
    }
    if (q != hipSuccess) { invalidate_all(); set_error(std::string("device addition: ") + hipGetErrorString(q)); return MPF_E_HIP; }
  }
  hold.release();
  std::memcpy(&h, gd_.h_out.p, sizeof(h));
  grow_launches_++;
  grow_ms_total_ += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  invalidate_all();                                // (whatever the kernel did, only it knows which vectors are whose)
  if (h.reason != GROW_DONE || h.steps_done != (uint32_t)steps) {
    grow_last_err_ = (int)(h.reason * 100u + h.err);
    return MPF_OK;                                 // the caller's own loop builds the tree (same seed, same draws: nothing was taken over)
  }
  // ---- replay the insertions on the mirror (buildNewTip + the hookups of :3158-3171)
  const uint32_t *out = gd_.h_out.p + hdr_words;
  for (int s = 0; s < steps; s++) {
    const int nextsp = ++ntips_;
    const int pr = nodep_[(size_t)perm[(size_t)nextsp]];
    const int q = nodep_[(size_t)nextnode_++];
    back_[(size_t)pr] = q;
    back_[(size_t)q] = pr;
    const int ins = rec_of(out[2 * s]);
    const int r = back_[(size_t)ins];
    hookup(nx(q), ins);
    hookup(nx(nx(q)), r);
    best_ = out[2 * s + 1];
    insert_rec_ = ins;
    if (best_per_step) best_per_step[nextsp] = best_;
    if (insert_per_step) insert_per_step[nextsp] = ins;
    stats.insertion_tests += (uint64_t)(2 * (nextsp - 1) - 3);
  }
  rng_.state = h.rng;
  topo_epoch_++;
  kids_dirty_ = true;
  for (int k = 0; k < 7; k++) grow_phase_ticks_[k] += h.tph[k];
  grow_steps_ += (uint64_t)steps;
  *done = true;
  return MPF_OK;
}

}  // namespace mpf
