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
int Engine::climb_fit_vw()
{
  if (climb_cus_ <= 0) {
    int c = 0;
    if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev_) != hipSuccess || c <= 0) c = 256;
    climb_cus_ = c;
  }
  const int cap = std::max(1, climb_cus_ * 85 / 100);
  for (int vw = (g_.S == 4) ? std::max(1, climb_vw_) : 1; vw <= ((g_.S == 4) ? 8 : 1); vw *= 2) {
    const size_t lds = climb_lds_bytes(g_, n_, vw);
    if (lds > 160 * 1024) continue;
    const int per_cu = (int)std::max<size_t>(1, (160 * 1024) / std::max<size_t>(lds, 1));
    if ((climb_tiles(g_, vw) + per_cu - 1) / per_cu <= cap) return vw;
  }
  return 0;
}

int Engine::climb_segment(int maxtrav_eff, int total, int *i, uint32_t *randomMP, unsigned *iter_hits, bool may_idle,
                          uint32_t *reason, uint32_t *n_moves)
{
  const auto t0 = std::chrono::steady_clock::now();
  const int vw = climb_fit_vw();
  if (vw <= 0) { set_error("device climb: the alignment's tiles do not fit the chip"); return MPF_E_STATE; }
  const int tiles = climb_tiles(g_, vw);
  const size_t ns = nslots_;
  const size_t hdr_words = (sizeof(ClimbHeader) + 3) / 4;
  const size_t out_words = hdr_words + 3 * (size_t)total;
  HIPCHK(cd_.bk.reserve(ns));
  HIPCHK(cd_.order.reserve((size_t)total));
  HIPCHK(cd_.sct.reserve((size_t)tiles * ns * 16));
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
  h.randomMP = *randomMP;
  h.iter_hits = *iter_hits;
  h.pos = (uint32_t)*i;
  h.insert_cid = insert_rec_ >= 0 ? (int32_t)slot(insert_rec_) : -1;
  h.remove_cid = remove_rec_ >= 0 ? (int32_t)slot(remove_rec_) : -1;
  h.since_move = 0;
  h.batch = 0;
  std::memcpy(cd_.h_out.p, &h, sizeof(h));
  ClimbParams p;
  p.vec = d_vec_;
  p.n = (uint32_t)n_;
  p.nslots = (uint32_t)ns;
  p.Wp = (uint32_t)g_.Wp;
  p.tiles = (uint32_t)tiles;
  p.total = (uint32_t)total;
  p.maxtrav = (uint32_t)maxtrav_eff;
  p.tie_mode = (uint32_t)tie_mode_;
  p.idle_limit = may_idle ? (uint32_t)climb_idle_ : 0u;
  p.max_moves = (uint32_t)total;
  p.batch_min = (uint32_t)std::max(1, std::min(climb_batch_min_, 8));
  p.batch_max = (uint32_t)std::max((int)p.batch_min, std::min(climb_batch_max_, 8));
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
  p.fault = (uint32_t)climb_fault_;
  climb_fault_ = 0;                                  // (one launch)
  p.stop_len = climb_stop_len_;
  if (climb_trace_) {
    HIPCHK(cd_.h_beat.reserve(32));
    std::memset(cd_.h_beat.p, 0, 32 * sizeof(uint32_t));
    p.beat = cd_.h_beat.p;
    const size_t cap = 1u << 16;
    HIPCHK(cd_.trace.reserve(8 * cap));
    p.trace = cd_.trace.p;
    p.trace_cap = (uint32_t)cap;
  }
  GateHold hold;
  {
    ClimbGate &g = g_gate[dev_ & 63];
    const size_t lds = climb_lds_bytes(g_, n_, vw);
    const int per_cu = (int)std::max<size_t>(1, (160 * 1024) / std::max<size_t>(lds, 1));
    std::unique_lock<std::mutex> lk(g.m);
    if (!g.cus) {
      int c = 0;
      if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev_) != hipSuccess || c <= 0) c = 256;
      g.cus = c;
    }
    // (counted in CU-equivalents of this launch's footprint: tiles / per_cu CUs)
    const int need = (tiles + per_cu - 1) / per_cu;
    // (not every CU takes a workgroup of this size at every moment -- other queues' kernels come and go -- so admission
    //  stops at 85 % of the chip: a launch beyond that waits its turn here instead of timing out in the kernel)
    const int cap = std::max(1, g.cus * 85 / 100);
    g.cv.wait(lk, [&] { return g.used == 0 || g.used + need <= cap; });
    g.used += need;
    hold.g = &g;
    hold.n = need;
  }
  HIPCHK(hipMemcpyAsync(cd_.bk.p, cd_.h_bk.p, ns * sizeof(uint16_t), hipMemcpyHostToDevice, st_));
  HIPCHK(hipMemcpyAsync(cd_.order.p, cd_.h_order.p, (size_t)total * sizeof(uint16_t), hipMemcpyHostToDevice, st_));
  HIPCHK(hipMemcpyAsync(cd_.out.p, cd_.h_out.p, sizeof(h), hipMemcpyHostToDevice, st_));
  HIPCHK(hipMemsetAsync(cd_.gsum.p, 0, gsum_words * sizeof(unsigned long long), st_));
  shadow_ok_ = false;                              // (k_climb rewrites vectors in the row-major store only)
  HIPCHK(launch_climb(st_, g_, vw, p));
  HIPCHK(hipMemcpyAsync(cd_.h_out.p, cd_.out.p, out_words * sizeof(uint32_t), hipMemcpyDeviceToHost, st_));
  {
    // a sweep segment takes milliseconds; a launch that is still running after many seconds will not come back (every wait
    // inside the kernel is bounded) -- report that instead of blocking the caller for ever
    const auto w0 = std::chrono::steady_clock::now();
    hipError_t q;
    long spins = 0;
    while ((q = hipStreamQuery(st_)) == hipErrorNotReady) {
      if ((++spins & 63) == 0) {
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count() > (climb_trace_ ? 4.0 : 20.0)) {
          std::string where;
          if (p.beat) for (int k = 0; k < 32; k++) where += " " + std::to_string(cd_.h_beat.p[k]);
          set_error("device climb: the launch did not finish within 20 s" + (where.empty() ? std::string() : " (step phase pos B ncand nops chains done err:" + where + ")"));
          // (the kernel may still be writing vectors: nothing this engine holds can be trusted any more)
          invalidate_all();
          broken_ = true;
          return MPF_E_STATE;
        }
        std::this_thread::yield();
      }
    }
    if (q != hipSuccess) { invalidate_all(); set_error(std::string("device climb: ") + hipGetErrorString(q)); return MPF_E_HIP; }
  }
  hold.release();
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
  if (h.reason == CLIMB_ERROR || h.reason == CLIMB_RUNNING || h.n_moves > (uint32_t)total) {
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

}  // namespace mpf
