// engine.cpp -- device state, tip packing, directional views and SPR-scan programs.
#include "engine.hpp"
#include <cstdlib>
#include "../host/simd_util.hpp"

#include <sched.h>

#include <algorithm>
#include <atomic>
#include <thread>
#include <ctime>
#include <chrono>
#include <climits>
#include <cstdio>
#include <cstring>

namespace mpf {

static inline double now_ms()
{
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
struct ScopedMs {
  double &acc, t0;
  explicit ScopedMs(double &a) : acc(a), t0(now_ms()) {}
  ~ScopedMs() { acc += now_ms() - t0; }
};

static thread_local std::string g_err;
void set_error(const std::string &msg) { g_err = msg; }
const std::string &last_error() { return g_err; }

#define HIPCHK(expr)                                                                              \
  do {                                                                                            \
    hipError_t e__ = (expr);                                                                      \
    if (e__ != hipSuccess) {                                                                      \
      set_error(std::string(#expr) + ": " + hipGetErrorString(e__) + " (" + __FILE__ + ":" +      \
                std::to_string(__LINE__) + ")");                                                  \
      return MPF_E_HIP;                                                                           \
    }                                                                                             \
  } while (0)

Engine::~Engine()
{
  if (std::getenv("MPF_VIEWS_PROFILE"))
    std::fprintf(stderr, "[views] launches %llu ops %llu levels %llu\n", (unsigned long long)stats.view_launches, (unsigned long long)stats.newview_ops, (unsigned long long)dbg_levels_);
  if (d_codes_) (void)hipFree(d_codes_);
  if (d_vec_) (void)hipFree(d_vec_);
  if (d_tipslots_) (void)hipFree(d_tipslots_);
  if (ev0_) (void)hipEventDestroy(ev0_);
  if (ev1_) (void)hipEventDestroy(ev1_);
  if (ev2_) (void)hipEventDestroy(ev2_);
  if (ev3_) (void)hipEventDestroy(ev3_);
  if (ev4_) (void)hipEventDestroy(ev4_);
  if (st_) (void)hipStreamDestroy(st_);
}

int Engine::init(const mpf_config &cfg, const uint8_t *codes, const int32_t *weights, const uint32_t *cost)
{
  if (cfg.n_taxa < 4 || cfg.n_patterns < 1 || !codes || !weights) {
    set_error("mpf_engine_create: need n_taxa >= 4, n_patterns >= 1, codes and weights");
    return MPF_E_INVALID;
  }
  if (cfg.datatype != MPF_DNA && cfg.datatype != MPF_AA && cfg.datatype != MPF_BIN && cfg.datatype != MPF_GENERIC) {
    set_error("mpf_engine_create: data type must be MPF_DNA, MPF_AA, MPF_BIN or MPF_GENERIC");
    return MPF_E_UNSUPPORTED;
  }
  if (const char *hp = std::getenv("MPF_HOST_POLL")) host_poll_ = std::atoi(hp) ? 1 : 0;        // (experiments)
  if (const char *vt = std::getenv("MPF_VIEWS_TILE")) { const int t = std::atoi(vt); if (t == 32 || t == 16 || t == 8 || t == 4 || t == 0) g_.nv_tile = t; }
  if (const char *pc = std::getenv("MPF_PLAN_CACHE")) plan_cache_ = std::atoi(pc);     // (debugging: default of option "plan_cache"; bits: 1 keep topology state, 2 refresh schedule, 4 sweep plans)
  // (Engines on several host threads share a device through their own streams; the runtime maps streams onto 4 hardware queues
  //  per process unless GPU_MAX_HW_QUEUES says otherwise, and a persistent k_climb launch holds its queue for a whole sweep --
  //  8 climbs side by side: 8.7 climbs/s on 4 queues, 19.8 on 16, profiles/r3/concurrent_climbs.txt.  The variable belongs to
  //  whoever starts the process -- INTEGRATION.md, mpboot_amd/engine.py --: a library does not edit its host's environment.)
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    set_error("no HIP device available: libmpfitch has no CPU fallback");
    return MPF_E_NO_DEVICE;
  }
  if (cfg.device < 0 || cfg.device >= ndev) {
    set_error("mpf_engine_create: device ordinal out of range");
    return MPF_E_INVALID;
  }
  dev_ = cfg.device;
  HIPCHK(hipSetDevice(dev_));
  n_ = cfg.n_taxa;
  P_ = cfg.n_patterns;
  datatype_ = cfg.datatype;
  keep_all_ = cfg.keep_all_sites;
  // states of the reference / undetermined code (globalVariables.h pLengths) / rows the kernels work on
  sref_ = datatype_ == MPF_DNA ? 4 : datatype_ == MPF_AA ? 20 : datatype_ == MPF_BIN ? 2 : 32;
  und_ = datatype_ == MPF_DNA ? 15 : datatype_ == MPF_AA ? 22 : datatype_ == MPF_BIN ? 3 : 32;
  g_.S = (datatype_ == MPF_DNA || datatype_ == MPF_BIN) ? 4 : 20;
  // (full refresh, one word per lane: k_newview_wgq on tiles chosen from the row length, kernels.hip: newview_tile)
  if (const char *vp = std::getenv("MPF_VIEWS_PIPE")) g_.nv_pipe = std::atoi(vp) != 0;      // (experiments: defaults of "views_pipe" / "views_tile")
  const int und = und_;
  for (size_t i = 0; i < (size_t)n_ * P_; i++) {
    if (codes[i] > und || ((datatype_ == MPF_DNA || datatype_ == MPF_BIN) && codes[i] == 0)) {
      set_error("mpf_engine_create: tip code outside the PLL alphabet");   // reference: assert(bitVector[nucleotide] > 0)
      return MPF_E_INVALID;
    }
  }
  codes_.assign(codes, codes + (size_t)n_ * P_);
  if (datatype_ == MPF_GENERIC) {
    bool used[32] = {false};
    // Multistate data with at most 20 symbols in use runs on the 20-state kernels: parsimony does not care what a state is
    // called, so the symbols in use are renumbered 0, 1, 2, ... in order of their codes (row gmap_[c] of a vector is the
    // reference's row c; unused rows do not exist).  More symbols -- or a cost matrix, under which a state that no tip has can
    // still be the cheapest label of an inner node (a Steiner point of the metric), so that the matrix's 32 states cannot be
    // renumbered -- take the 32-state kernels with the reference's own numbering (its `default:` / `case 32` branches,
    // sprparsimony.cpp:824-869, :1164-1203, :571-573).
    for (uint8_t c : codes_) if (c < 32) used[c] = true;
    int k = 0;
    for (int c = 0; c < 32; c++) gmap_[c] = used[c] ? k++ : -1;
    if (k > 20 || cost) {
      g_.S = 32;
      for (int c = 0; c < 32; c++) gmap_[c] = c;
    } else {
      for (uint8_t &c : codes_) if (c < 32) c = (uint8_t)gmap_[c];
    }
  }
  if (cost) {
    // ParsTree::loadCostMatrixFile's triangle-inequality closure (reference parstree.cpp:74-80).  A matrix that is not
    // symmetric afterwards makes the length of a tree depend on where it is rooted; the reference roots every evaluation
    // at the edge it is handed, and so do the kernels (asym_: the root side of a test goes through the transposed matrix)
    const int S = g_.S;
    if (sref_ == S) {
      cost_.assign(cost, cost + S * S);
    } else {
      // binary in the 4-state / multistate in the 20-state kernels (the reference instantiates its Sankoff kernels for 2, 4, 20
      // and 32 states, sprparsimony.cpp:559-639): the caller's sref x sref matrix, renumbered like the symbols; a state no tip
      // can have costs more to enter or leave than any real change, so no minimum is ever taken through it
      auto row_of = [&](int c) { return datatype_ == MPF_GENERIC ? gmap_[c] : (c < S ? c : -1); };
      uint32_t hi = 0;
      for (int i = 0; i < sref_; i++) for (int j = 0; j < sref_; j++) if (row_of(i) >= 0 && row_of(j) >= 0) hi = std::max(hi, cost[i * sref_ + j]);
      cost_.assign((size_t)S * S, hi + 1);
      for (int i = 0; i < S; i++) cost_[(size_t)i * S + i] = 0;
      for (int i = 0; i < sref_; i++) for (int j = 0; j < sref_; j++) if (row_of(i) >= 0 && row_of(j) >= 0) cost_[(size_t)row_of(i) * S + row_of(j)] = cost[i * sref_ + j];
    }
    for (int k = 0; k < S; k++)
      for (int i = 0; i < S; i++)
        for (int j = 0; j < S; j++)
          if (cost_[i * S + j] > cost_[i * S + k] + cost_[k * S + j]) cost_[i * S + j] = cost_[i * S + k] + cost_[k * S + j];
    uint32_t hi = 0;
    for (int i = 0; i < S; i++)
      for (int j = 0; j < S; j++) {
        if (cost_[i * S + j] != cost_[j * S + i]) asym_ = true;
        hi = std::max(hi, cost_[i * S + j]);
      }
    if (hi >= 65535u) { set_error("Sankoff costs too large"); return MPF_E_UNSUPPORTED; }
    sankoff_ = true;
    g_.sankoff = 1;
    g_.highest_cost = hi + 1;                   // highest_cost, reference sprparsimony.cpp:160
  }
  g_.vw = 1;
  g_.reduce = 0;
  g_.map = 1;            // XCD-aware work mapping (speed only)
  wgt_.assign(weights, weights + P_);
  inf_.assign(P_, 0);
  first_site_.assign(P_, -1);
  HIPCHK(hipStreamCreateWithFlags(&st_, hipStreamNonBlocking));
  HIPCHK(hipEventCreate(&ev0_));
  HIPCHK(hipEventCreate(&ev1_));
  HIPCHK(hipEventCreate(&ev2_));
  HIPCHK(hipEventCreate(&ev3_));
  HIPCHK(hipEventCreate(&ev4_));
  HIPCHK(hipMalloc((void **)&d_codes_, (size_t)n_ * P_));
  HIPCHK(hipMemcpy(d_codes_, codes_.data(), (size_t)n_ * P_, hipMemcpyHostToDevice));
  nslots_ = (size_t)n_ + 3 * (size_t)(n_ - 1);
  HIPCHK(reserve_results(4096));
  HIPCHK(d_done_.reserve(64));
  HIPCHK(hipMemset(d_done_.p, 0, 64 * sizeof(uint32_t)));
  HIPCHK(hipMalloc((void **)&d_tipslots_, (size_t)n_ * sizeof(uint32_t)));
  {
    std::vector<uint32_t> ts(n_);
    for (int i = 0; i < n_; i++) ts[i] = (uint32_t)i;
    HIPCHK(hipMemcpy(d_tipslots_, ts.data(), n_ * sizeof(uint32_t), hipMemcpyHostToDevice));
  }
  back_.assign(3 * (size_t)(2 * n_ - 1) + 3, -1);
  nodep_.assign(2 * (size_t)n_, 0);
  reset_node_order();                              // (nodep[i] = node i's first record, as the reference's tree set-up leaves it: a caller need not ask for it)
  sc_.assign(back_.size(), 0);
  valid_.assign(back_.size(), 0);
  lev_.assign(back_.size(), 0);
  lev_epoch_.assign(back_.size(), 0);
  nvis_val_.assign(back_.size() * 16, 0);
  nvis_epoch_.assign(back_.size() * 16, 0);
  reset_node_order();
  rng_.seed(1);
  return pack();
}

// ---- determineUninformativeSites + compressDNA (reference sprparsimony.cpp:2460-2499, :2596-2634, :2828-2973)
// The filter runs on the host (one pass over the codes); the bit transposition runs on the device.
int Engine::pack()
{
  pack_gen_++;
  const int und = und_;
  long entries = 0;
  ninf_ = 0;
  if (sankoff_) {
    // compressSankoffDNA (reference sprparsimony.cpp:2636-2825): informative patterns are kept once each, their
    // weights go to informativePtnWgt; one 32-bit cost per state and pattern
    inf_index_.clear();
    if (!inf_known_) {
      std::vector<uint32_t> seen((size_t)P_, 0u);
      for (int t = 0; t < n_; t++) {
        const uint8_t *row = codes_.data() + (size_t)t * P_;
        for (int s = 0; s < P_; s++) if (row[s] < und) seen[(size_t)s] |= 1u << row[s];
      }
      for (int s = 0; s < P_; s++) inf_[s] = keep_all_ ? 1 : (__builtin_popcount(seen[(size_t)s]) > 1);
      inf_known_ = true;
    }
    for (int s = 0; s < P_; s++) {
      const int keep = inf_[s];
      first_site_[s] = keep ? (int32_t)inf_index_.size() : -1;
      if (wgt_[s] < 0) { set_error("negative pattern weight"); return MPF_E_INVALID; }
      if (keep) inf_index_.push_back(s);
    }
    ninf_ = (int)inf_index_.size();
    nsites_ = ninf_;
    Wref_ = (ninf_ % 16) ? ninf_ + (16 - ninf_ % 16) : ninf_;     // parsimonyLength with VECSIZE = 16 (u16, AVX)
    int wp = ((ninf_ + 31) / 32) * 32;
    if (wp == 0) wp = 32;
    // 16-bit costs, two patterns per lane (the reference's default "short" arithmetic, sprparsimony.cpp:556-641), while
    // no intermediate can reach 2^16: a view entry is at most (tips below) x max cost, a candidate sums three terms
    const bool want16 = snk16_opt_ != 0 && 3ull * (uint64_t)n_ * (uint64_t)g_.highest_cost < 65536ull;
    if (wp != g_.Wp || !d_vec_ || (int)want16 != g_.snk16) {
      g_.Wp = wp;
      g_.snk16 = want16 ? 1 : 0;
      vec_words_ = nslots_ * (size_t)g_.S * (size_t)(want16 ? g_.Wp / 2 : g_.Wp);
      // weighted mode keeps m(v) = min-plus transform of every vector next to v (second half of the allocation): a
      // transform costs 2 S^2 operations per pattern, and each stored one is used by up to three consumers
      g_.moff = vec_words_;
      HIPCHK(vec_store_fit(2 * vec_words_ * sizeof(uint32_t)));
    }
    std::vector<uint32_t> pw((size_t)g_.Wp, 0u);
    for (int j = 0; j < ninf_; j++) pw[(size_t)j] = (uint32_t)wgt_[(size_t)inf_index_[(size_t)j]];
    HIPCHK(d_pwgt_.reserve(pw.size()));
    HIPCHK(d_cost_.reserve(cost_.size()));
    HIPCHK(d_infidx_.reserve(std::max<size_t>(inf_index_.size(), 1)));
    HIPCHK(hipMemcpyAsync(d_pwgt_.p, pw.data(), pw.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st_));
    {
      // the device copy of the cost matrix: in the 16-bit packing every entry twice (c | c << 16)
      // ... and, behind the S x S entries, the pair-packed rows the scan kernel's 16-bit transform reads (c[z][2j] | c[z][2j + 1] << 16:
      // kernels.hip, mplus on a pair-packed matrix)
      const size_t SS = cost_.size(), S_ = (size_t)g_.S;
      auto with_pairs = [&](std::vector<uint32_t> &dev, auto entry) {
        dev.assign(SS + SS / 2, 0u);
        for (size_t z = 0; z < S_; z++)
          for (size_t x = 0; x < S_; x++) {
            const uint32_t c = entry(z, x);
            dev[z * S_ + x] = g_.snk16 ? (c | (c << 16)) : c;
            if (g_.snk16) dev[SS + z * (S_ / 2) + x / 2] |= (c & 0xFFFFu) << (16 * (x & 1));
          }
      };
      with_pairs(cost_dev_, [&](size_t z, size_t x) { return cost_[z * S_ + x]; });
      HIPCHK(d_cost_.reserve(cost_dev_.size()));
      HIPCHK(hipMemcpyAsync(d_cost_.p, cost_dev_.data(), cost_dev_.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st_));
      g_.costT = nullptr;
      if (asym_) {
        const int S = g_.S;
        (void)S;
        with_pairs(costT_dev_, [&](size_t z, size_t x) { return cost_[x * S_ + z]; });
        HIPCHK(d_costT_.reserve(costT_dev_.size()));
        HIPCHK(hipMemcpyAsync(d_costT_.p, costT_dev_.data(), costT_dev_.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st_));
        g_.costT = d_costT_.p;
      }
    }
    if (!inf_index_.empty())
      HIPCHK(hipMemcpyAsync(d_infidx_.p, inf_index_.data(), inf_index_.size() * sizeof(int32_t), hipMemcpyHostToDevice, st_));
    g_.cost = d_cost_.p;
    g_.pwgt = d_pwgt_.p;
    HIPCHK(launch_pack_tips_sankoff(st_, g_, d_vec_, d_codes_, n_, P_, d_infidx_.p, ninf_, datatype_));
    HIPCHK(hipStreamSynchronize(st_));
    invalidate_vectors();
    return MPF_OK;
  }
  if (!inf_known_) {
    // one pass over the codes, taxon-major so that the rows stream
    std::vector<uint32_t> seen((size_t)P_, 0u);
    for (int t = 0; t < n_; t++) {
      const uint8_t *row = codes_.data() + (size_t)t * P_;
      for (int s = 0; s < P_; s++) if (row[s] < und) seen[(size_t)s] |= 1u << row[s];
    }
    for (int s = 0; s < P_; s++) inf_[s] = keep_all_ ? 1 : (__builtin_popcount(seen[(size_t)s]) > 1);
    inf_known_ = true;
  }
  for (int s = 0; s < P_; s++) {
    const int keep = inf_[s];
    if (wgt_[s] < 0) { set_error("negative pattern weight"); return MPF_E_INVALID; }
    if (keep) { first_site_[s] = (int32_t)entries; entries += wgt_[s]; ninf_++; } else first_site_[s] = -1;
  }
  nsites_ = (int)entries;
  const int ce = (int)((entries + 31) / 32);
  Wref_ = (ce % 8) ? ce + (8 - ce % 8) : ce;           // the reference's parsimonyLength (AVX build)
  int wp = ((ce + 31) / 32) * 32;                      // our row pitch: whole 128-byte lines
  if (wp == 0) wp = 32;
  if (wp != g_.Wp || !d_vec_) {
    g_.Wp = wp;
    vec_words_ = nslots_ * (size_t)g_.S * g_.Wp;
    // below 2 GiB the scan kernel addresses the whole store through one raw buffer (32-bit offsets); above, 64-bit bases
    g_.big = (force_big_ || vec_words_ * sizeof(uint32_t) >= ((size_t)1 << 31)) ? 1 : 0;
    // DNA below 2 GiB: room for the word-major copy the planned scan reads (Geometry::shoff)
    g_.shoff = (g_.S == 4 && vec_words_ * sizeof(uint32_t) < ((size_t)1 << 31)) ? vec_words_ : 0;
    HIPCHK(vec_store_fit((vec_words_ + (g_.shoff ? vec_words_ : 0)) * sizeof(uint32_t)));
  }
  shadow_ok_ = false;                          // (until the tips' copies are rewritten below: invalidate_vectors sets it again)
  std::vector<int32_t> s2p((size_t)std::max(nsites_, 1));
  for (int s = 0; s < P_; s++)
    if (inf_[s])
      for (int w = 0; w < wgt_[s]; w++) s2p[(size_t)first_site_[s] + w] = s;
  HIPCHK(d_site2ptn_.reserve(s2p.size()));
  HIPCHK(hipMemcpyAsync(d_site2ptn_.p, s2p.data(), s2p.size() * sizeof(int32_t), hipMemcpyHostToDevice, st_));
  HIPCHK(launch_pack_tips(st_, g_, d_vec_, d_codes_, n_, P_, d_site2ptn_.p, nsites_, datatype_, d_tipslots_));
  HIPCHK(hipStreamSynchronize(st_));
  invalidate_vectors();                      // (re-weighting changes every vector, not the tree)
  return MPF_OK;
}

// The vector store follows the packing: a re-weighting changes the row pitch (a ratchet climb packs half of the informative sites once
// more: +50 %, and the climb behind it the original sites again).  The ALLOCATION only ever grows -- rounds 1-5 freed and allocated
// it on every change of the pitch, four times per ratchet iteration, and hipFree waits for the whole device: every other engine's
// stream stood still each time (eight chains of an iteration-parallel -bb run spent more time there than in their kernels).
hipError_t Engine::vec_store_fit(size_t bytes)
{
  if (d_vec_ && bytes <= vec_cap_bytes_) return hipSuccess;
  if (d_vec_) { (void)hipFree(d_vec_); d_vec_ = nullptr; vec_cap_bytes_ = 0; }
  const hipError_t e = hipMalloc((void **)&d_vec_, bytes);
  if (e == hipSuccess) vec_cap_bytes_ = bytes;
  return e;
}

int Engine::set_weights(const int32_t *weights)
{
  wgt_.assign(weights, weights + P_);
  // An attached tracker follows the re-weighting: its sample weights are laid out again by the sites of the new packing and
  // climbs under other weights than the attach-time ones are booked as the reference books ratchet climbs
  // (iqtree.cpp:3283-3295; host/ufboot.cpp).  Only weights that leave an attach-time pattern WITHOUT a site (weight 0 --
  // mpboot's ratchet only adds copies, alignment.cpp:1915-1969) suspend the bookkeeping until other weights arrive.
  if (ufb_) {
    const bool other = wgt_ != ufb_->attach_wgt;
    bool lost = false;
    if (other)
      for (int k = 0; k < P_ && !lost; k++) lost = ufb_->attach_wgt[(size_t)k] > 0 && wgt_[(size_t)k] <= 0;
    ufb_->suspended = (other && lost) || (other && !ufb_->ratchet_booking);
    ufb_->ratchet = other && !ufb_->suspended;
    ufb_->rt_valid = false;
  }
  int rc = pack();
  if (rc) return rc;
  if (ufb_ && !ufb_->suspended) return ufb_layout_weights();
  return MPF_OK;
}

int Engine::tip_vector(int tipno, uint32_t *out)
{
  if (tipno < 1 || tipno > n_) { set_error("tip out of range"); return MPF_E_INVALID; }
  if (sankoff_) { set_error("mpf_get_tip_vector: bit-packed tips exist in Fitch mode only"); return MPF_E_UNSUPPORTED; }
  std::vector<uint32_t> tmp((size_t)g_.S * g_.Wp);
  HIPCHK(hipMemcpy(tmp.data(), d_vec_ + (size_t)(tipno - 1) * g_.S * g_.Wp, tmp.size() * sizeof(uint32_t),
                   hipMemcpyDeviceToHost));
  // hand back the reference's rows: its state count (rows the kernels do not carry -- multistate symbols beyond the 20th, never
  // present -- are empty) and its length W; sites behind the alignment are all ones in every row (sprparsimony.cpp:2947-2960)
  for (int k = 0; k < sref_; k++)
    for (int w = 0; w < Wref_; w++) {
      uint32_t v;
      const int er = datatype_ == MPF_GENERIC ? gmap_[k] : (k < g_.S ? k : -1);      // the engine's row of the reference's state k
      if (er >= 0) v = w < g_.Wp ? tmp[(size_t)er * g_.Wp + w] : 0xFFFFFFFFu;
      else if (w >= g_.Wp) v = 0xFFFFFFFFu;
      else {
        // a symbol no taxon has: its row is set exactly where this tip is undetermined (bitVector32[32] = every state) and
        // behind the alignment -- which is what an engine row without a symbol holds (or, with all 20 rows taken, what two
        // rows have in common: a tip with a symbol sets one row only)
        int used_rows = 0;
        for (int c = 0; c < 32; c++) used_rows += gmap_[c] >= 0 ? 1 : 0;
        v = used_rows < g_.S ? tmp[(size_t)used_rows * g_.Wp + w] : (tmp[w] & tmp[(size_t)g_.Wp + w]);
      }
      out[(size_t)k * Wref_ + w] = v;
    }
  return MPF_OK;
}

// ---- tree plumbing
void Engine::reset_node_order()
{
  for (int i = 1; i <= 2 * n_ - 1; i++) nodep_[i] = 3 * i;
}

int Engine::set_tree(const int32_t *back)
{
  const size_t len = 3 * (size_t)(2 * n_ - 1);
  for (int v = 1; v <= 2 * n_ - 2; v++)
    for (int s = 0; s < (v <= n_ ? 1 : 3); s++) {
      const int r = 3 * v + s, b = back[r];
      if (b < 3 || b >= (int)len || back[b] != r) { set_error("mpf_set_tree: inconsistent back links"); return MPF_E_INVALID; }
    }
  const bool same = have_tree_ && ntips_ == n_ && std::equal(back + 3, back + len, back_.begin() + 3);
  std::copy(back, back + len, back_.begin());
  start_ = nodep_[1];
  ntips_ = n_;
  have_tree_ = true;
  if (same) invalidate_vectors(); else invalidate_all();
  return MPF_OK;
}

void Engine::get_tree(int32_t *back) const { std::copy(back_.begin(), back_.begin() + 3 * (size_t)(2 * n_ - 1), back); }

// reorderNodes / nodeRectifierPars (reference sprparsimony.cpp:2046-2101): nodep[n+1+k] = the record by
// which the k-th inner node is entered in a preorder walk from start->back
void Engine::node_rectifier()
{
  start_ = nodep_[1];
  int count = 0;
  std::vector<int> &stack = sv_stack_;             // (a member: this runs once per sweep, an allocation would show)
  stack.clear();
  stack.push_back(back_[start_]);
  while (!stack.empty()) {
    const int p = stack.back();
    stack.pop_back();
    if (tip(p)) continue;
    nodep_[count + n_ + 1] = p;
    count++;
    stack.push_back(back_[nx(nx(p))]);
    stack.push_back(back_[nx(p)]);
  }
}

// ---- directional views ---------------------------------------------------------------------
// vec[r] (r inner) = fitch(vec[back[nx r]], vec[back[nx nx r]]).  Every vector carries a validity flag;
// a topology edit invalidates exactly the vectors whose subtree contains an edited node, and a refresh
// recomputes only the invalid vectors that the requested roots depend on, level by level.

void Engine::invalidate_all()
{
  std::fill(valid_.begin(), valid_.end(), 0);
  n_invalid_ = -1;
  views_valid_ = false;
  kids_dirty_ = true;
  kids_list_.clear();
  all_invalid_ = true;
  topo_epoch_++;
  sched_cache_valid_ = false;
  sweep_cache_valid_ = false;
  shadow_ok_ = g_.shoff != 0;                      // (no valid inner vector left that the word-major copy could disagree with)
}

// every vector stale, topology unchanged (re-weighting, a full re-evaluation, the same tree handed over again): what
// depends on the topology alone stays -- the device copy of kids[], the from-scratch refresh schedule, a sweep's plans
void Engine::invalidate_vectors()
{
  if (!(plan_cache_ & 1) || !kids_list_.empty()) { invalidate_all(); return; }
  std::fill(valid_.begin(), valid_.end(), 0);
  n_invalid_ = -1;
  views_valid_ = false;
  all_invalid_ = true;
  shadow_ok_ = g_.shoff != 0;
}

// Invariant: a valid vector has valid inputs.  The vectors containing `node` are its own three and, walking
// outwards, the two outward-looking vectors of every node reached; the walk stops at vectors that are
// already invalid (everything beyond them is invalid by the invariant).
void Engine::invalidate_node(int node)
{
  if (node <= n_) return;
  std::vector<int> stack;
  for (int s = 0; s < 3; s++) {
    const int r = 3 * node + s;
    if (valid_[r]) { valid_[r] = 0; if (n_invalid_ >= 0) n_invalid_++; }
    const int w = back_[r];
    if (w >= 0 && !tip(w)) stack.push_back(w);
  }
  while (!stack.empty()) {
    const int w = stack.back();           // record on a neighbouring node, looking back at where we came from
    stack.pop_back();
    const int o[2] = {nx(w), nx(nx(w))};
    for (int k = 0; k < 2; k++) {
      const int r = o[k];
      if (!valid_[r]) continue;
      valid_[r] = 0;
      if (n_invalid_ >= 0) n_invalid_++;
      const int u = back_[r];
      if (u >= 0 && !tip(u)) stack.push_back(u);
    }
  }
  views_valid_ = false;
  // the device copy of the topology (kids[]) changes exactly at the records of edited nodes; while it is complete
  // (no wholesale invalidation pending) a refresh sends just these
  if (!kids_dirty_) for (int s = 0; s < 3; s++) kids_list_.push_back(3 * node + s);
  topo_epoch_++;
  sched_cache_valid_ = false;
  sweep_cache_valid_ = false;
}

// N(q, m) = 1 + [q inner and m > 1] * (N(c1, m-1) + N(c2, m-1)): the records addTraverseParsimony visits
// from q with m levels left (reference sprparsimony.cpp:2208-2218); depends on the topology only
// the whole table in one bottom-up sweep (a full-sweep plan asks for nearly every entry anyway)
void Engine::fill_visit_counts(int maxm)
{
  // dense table for ONE m (plan_walk only ever asks for m = maxtrav), built with two rolling arrays that stay in L1.
  // Indexed by vector slot, over the host mirror of the device topology (kids_host_: the slots of the two records behind an
  // inner record, complete after the full schedule_views that precedes a sweep's planning): one sequential pass per level,
  // no record arithmetic in the loop (0.16 -> 0.11 ms of planning per C3 sweep).
  const size_t ns = nslots_;
  if (kids_host_.size() != ns) kids_host_.assign(ns, make_uint2(0u, 0u));
  if (kids_dirty_ || !kids_list_.empty()) {
    // (the mirror is not known to be complete -- no full refresh since the last wholesale invalidation: rebuild it from the
    //  record links; it stays marked as it was, this copy only serves the counts below)
    for (size_t r = 3 * ((size_t)n_ + 1); r < back_.size(); r++)
      if (back_[r] >= 0 && back_[nx((int)r)] >= 0 && back_[nx(nx((int)r))] >= 0)
        kids_host_[slot((int)r)] = make_uint2(slot(back_[nx((int)r)]), slot(back_[nx(nx((int)r))]));
  }
  vis_a_.assign(ns, 1);
  vis_dense_.assign(ns, 1);
  std::vector<int32_t> *prev = &vis_a_, *cur = &vis_dense_;
  const uint2 *kd = kids_host_.data();
  for (int m = 2; m <= maxm; m++) {
    const int32_t *pa = prev->data();
    int32_t *pc = cur->data();
    for (size_t c = (size_t)n_; c < ns; c++) pc[c] = 1 + pa[kd[c].x] + pa[kd[c].y];        // tips keep 1
    std::swap(prev, cur);
  }
  if (prev != &vis_dense_) vis_dense_.swap(vis_a_);               // prev holds the last level written
  vis_dense_m_ = maxm;
}

int Engine::count_visits(int q, int m)
{
  if (m <= 1 || tip(q)) return 1;
  if (m == vis_dense_m_ && visits_filled_epoch_ == topo_epoch_) return vis_dense_[(size_t)slot(q)];
  const size_t key = (size_t)q * 16 + (size_t)m;
  if (nvis_epoch_[key] == topo_epoch_) return nvis_val_[key];
  const int v = 1 + count_visits(back_[nx(q)], m - 1) + count_visits(back_[nx(nx(q))], m - 1);
  nvis_val_[key] = v;
  nvis_epoch_[key] = topo_epoch_;
  return v;
}

// Cut the refresh's dependency graph (ops = `order`, topologically sorted) into chains for k_newview_chain: op j continues
// the chain of op d when d is j's ONLY stale input (its other input is valid already, so the wave can prefetch it); of the
// up to two consumers of d the one with the longer path above it continues, the other starts a chain of the next level.
// A chain's level is one more than the deepest chain it takes an input from; per level the chains are spread over the 16
// waves of a workgroup, longest first.
void Engine::build_chains(const std::vector<int> &order)
{
  const int N = (int)order.size();
  if (sv_idx_.size() != back_.size()) sv_idx_.assign(back_.size(), 0);
  for (int i = 0; i < N; i++) sv_idx_[(size_t)order[(size_t)i]] = i;
  auto dep_of = [&](int x) { return (!tip(x) && lev_epoch_[x] == epoch_) ? sv_idx_[(size_t)x] : -1; };
  ch_d0_.resize((size_t)N); ch_d1_.resize((size_t)N); ch_h_.assign((size_t)N, 0); ch_next_.assign((size_t)N, -1);
  ch_chain_.resize((size_t)N);
  for (int i = 0; i < N; i++) {
    const int r = order[(size_t)i];
    ch_d0_[(size_t)i] = dep_of(back_[nx(r)]);
    ch_d1_[(size_t)i] = dep_of(back_[nx(nx(r))]);
  }
  for (int i = N - 1; i >= 0; i--) {
    const int h = ch_h_[(size_t)i] + 1;
    const int d0 = ch_d0_[(size_t)i], d1 = ch_d1_[(size_t)i];
    if (d0 >= 0 && ch_h_[(size_t)d0] < h) ch_h_[(size_t)d0] = h;
    if (d1 >= 0 && ch_h_[(size_t)d1] < h) ch_h_[(size_t)d1] = h;
  }
  auto single = [&](int j) { const int d0 = ch_d0_[(size_t)j], d1 = ch_d1_[(size_t)j]; return (d0 >= 0) != (d1 >= 0) ? (d0 >= 0 ? d0 : d1) : -1; };
  for (int j = 0; j < N; j++) {
    const int d = single(j);
    if (d >= 0 && (ch_next_[(size_t)d] < 0 || ch_h_[(size_t)j] > ch_h_[(size_t)ch_next_[(size_t)d]])) ch_next_[(size_t)d] = j;
  }
  // chains in order of their heads; level of a chain from its head's inputs
  ch_head_.clear(); ch_len_.clear(); ch_slev_.clear();
  int nlev = 0;
  for (int j = 0; j < N; j++) {
    const int d = single(j);
    if (d >= 0 && ch_next_[(size_t)d] == j) continue;           // a link, reached from its head
    const int c = (int)ch_head_.size();
    int lev = 0;
    if (ch_d0_[(size_t)j] >= 0) lev = std::max(lev, ch_slev_[(size_t)ch_chain_[(size_t)ch_d0_[(size_t)j]]] + 1);
    if (ch_d1_[(size_t)j] >= 0) lev = std::max(lev, ch_slev_[(size_t)ch_chain_[(size_t)ch_d1_[(size_t)j]]] + 1);
    int len = 0;
    for (int k = j; k >= 0; k = ch_next_[(size_t)k]) { ch_chain_[(size_t)k] = c; len++; }
    ch_head_.push_back(j);
    ch_len_.push_back(len);
    ch_slev_.push_back(lev);
    nlev = std::max(nlev, lev + 1);
  }
  const int C = (int)ch_head_.size();
  // per level: chains longest first onto the least loaded wave
  ch_lev_off_.assign((size_t)nlev + 1, 0);
  for (int c = 0; c < C; c++) ch_lev_off_[(size_t)ch_slev_[(size_t)c] + 1]++;
  for (int l = 0; l < nlev; l++) ch_lev_off_[(size_t)l + 1] += ch_lev_off_[(size_t)l];
  ch_sorted_.resize((size_t)C);
  {
    std::vector<int> &fill = sv_fill_;
    fill.assign(ch_lev_off_.begin(), ch_lev_off_.end() - 1);
    for (int c = 0; c < C; c++) ch_sorted_[(size_t)fill[(size_t)ch_slev_[(size_t)c]]++] = c;
  }
  ch_wave_.resize((size_t)C);
  ch_off_.assign((size_t)nlev * 16 + 1, 0);
  for (int l = 0; l < nlev; l++) {
    int *b = ch_sorted_.data() + ch_lev_off_[(size_t)l], *e = ch_sorted_.data() + ch_lev_off_[(size_t)l + 1];
    if (e - b > 16) std::sort(b, e, [&](int x, int y) { return ch_len_[(size_t)x] != ch_len_[(size_t)y] ? ch_len_[(size_t)x] > ch_len_[(size_t)y] : x < y; });
    int load[16] = {0};
    for (int *p = b; p < e; p++) {
      int w = 0;
      for (int k = 1; k < 16; k++) if (load[k] < load[w]) w = k;
      ch_wave_[(size_t)*p] = w;
      load[w] += ch_len_[(size_t)*p];
    }
    for (int w = 0; w < 16; w++) ch_off_[(size_t)l * 16 + (size_t)w + 1] = load[w];
  }
  for (size_t i = 1; i < ch_off_.size(); i++) ch_off_[i] += ch_off_[i - 1];
  // emit: position of every op
  ch_ops_.resize((size_t)N);
  {
    std::vector<int> &fill = sv_fill_;
    fill.assign(ch_off_.begin(), ch_off_.end() - 1);
    for (int l = 0; l < nlev; l++)
      for (int i = ch_lev_off_[(size_t)l]; i < ch_lev_off_[(size_t)l + 1]; i++) {
        const int c = ch_sorted_[(size_t)i];
        int &at = fill[(size_t)l * 16 + (size_t)ch_wave_[(size_t)c]];
        bool first = true;
        for (int k = ch_head_[(size_t)c]; k >= 0; k = ch_next_[(size_t)k]) {
          const int r = order[(size_t)k];
          ChainOp &o = ch_ops_[(size_t)at++];
          o.rec = r;
          // a link's memory operand is the input that is NOT the previous op
          o.other = first ? -1 : (ch_d0_[(size_t)k] >= 0 ? back_[nx(nx(r))] : back_[nx(r)]);
          first = false;
        }
      }
  }
  ch_levels_ = nlev;
}

// A new topology, nothing valid, the tree complete: the schedule of the refresh can be made on the device (k_sched)
bool Engine::dev_sched_usable() const
{
  return have_tree_ && all_invalid_ && dev_sched_ && !sankoff_ && ntips_ == n_ && n_ >= 4 && g_.vw == 1 && g_.nv_pipe && views_mode_ >= 1 &&
         nslots_ <= kSchedMaxSlots;
}

// From-scratch refresh with the schedule made on the device: the host only lays out the topology array (kids[cid], 8 bytes per
// vector) and uploads it; levels, ops in level order and level offsets come from one workgroup (k_sched) in front of the refresh
// kernel, while the host goes on.  sweep_maxtrav > 0: a second workgroup of the same launch lays out the scan descriptors of a
// whole sweep of that radius (nodep_ must be current) -- what plan_walk does on the host -- and tells the host how many.
int Engine::schedule_views_dev(int sweep_maxtrav)
{
  ScopedMs timer(stats.host_views_ms_total);
  const size_t nops = 3 * (size_t)(n_ - 2);
  const size_t n_prune = sweep_maxtrav > 0 ? 2 * (size_t)n_ - 2 : 0;
  if (kids_host_.size() != nslots_) kids_host_.assign(nslots_, make_uint2(0u, 0u));
  for (int v = n_ + 1; v <= 2 * n_ - 2; v++) {
    const int r0 = 3 * v;
    const uint32_t s0 = slot(back_[r0]), s1 = slot(back_[r0 + 1]), s2 = slot(back_[r0 + 2]);
    uint2 *k = &kids_host_[slot(r0)];            // the three records of a node sit side by side, in `next` order
    k[0] = make_uint2(s1, s2);
    k[1] = make_uint2(s2, s0);
    k[2] = make_uint2(s0, s1);
  }
  kids_list_.clear();
  // staging layout: [kids][prune records of the sweep][ops][level offsets][level count]; the first two go up in one copy
  const size_t kids_bytes = nslots_ * sizeof(uint2);
  const size_t nodep_off = (kids_bytes + 15) & ~(size_t)15;
  const size_t ops_off = nodep_off + ((n_prune * sizeof(uint32_t) + 15) & ~(size_t)15);
  const size_t lev_off_b = ops_off + ((nops * sizeof(NvOp) + 15) & ~(size_t)15);
  const size_t nlev_off = lev_off_b + ((((size_t)n_ + 2) * sizeof(int32_t) + 15) & ~(size_t)15);
  const size_t total_b = nlev_off + 64;
  // the topology array and the prune records stay in pinned host memory of their own (k_sched reads them over the bus: 40 KB,
  // no copy dispatch in front of the launch; nothing else writes this buffer, and a result of the launch is waited for before
  // the next one is prepared)
  HIPCHK(h_kstage_.reserve(ops_off));
  HIPCHK(d_vstage_.reserve(total_b));
  std::memcpy(h_kstage_.p, kids_host_.data(), kids_bytes);
  uint8_t *src = d_vstage_.p;
  SweepDescArgs sw;
  if (n_prune) {
    uint32_t *np = reinterpret_cast<uint32_t *>(h_kstage_.p + nodep_off);
    for (size_t i = 0; i < n_prune; i++) np[i] = slot(nodep_[i + 1]);
    const size_t cap = 8 * n_prune;              // a prune node has two neighbourhoods of at most four parts ...
    HIPCHK(d_walk_.reserve(cap));
    HIPCHK(d_parts_.reserve(cap));
    HIPCHK(h_dsw_.reserve(4 + cap));
    HIPCHK(d_prog_.reserve(scan_prog_bytes((int)cap)));
    HIPCHK(reserve_results(cap * 63));           // ... of at most 2^6 - 1 insertion tests each (everything the launches below touch is in place now)
    sw.nodep = reinterpret_cast<const uint32_t *>(h_kstage_.p + nodep_off);
    sw.n_prune = (uint32_t)n_prune;
    sw.maxtrav = (uint32_t)sweep_maxtrav;
    sw.split_cands = (uint32_t)std::max(0, split_cands_);
    sw.desc = d_walk_.p;
    sw.parts = d_parts_.p;
    sw.hdr_host = h_dsw_.p;
    sw.part_node = h_dsw_.p + 4;
    sw.hdr_dev = reinterpret_cast<uint32_t *>(src + nlev_off) + 8;
    __atomic_store_n(h_dsw_.p + 3, 0u, __ATOMIC_RELAXED);
    __atomic_store_n(h_dsw_.p + 2, 0u, __ATOMIC_RELAXED);
    walk_gen_++;                                 // whatever descriptors, program and part table were on the device are gone
    n_walk_ = 0;
    walk_out_ = 0;
    sweep_cache_valid_ = false;
  }
  kids_dirty_ = false;
  kids_upload_ = false;
  kids_dev_ready_ = true;
  NvOp *dops = reinterpret_cast<NvOp *>(src + ops_off);
  int32_t *dlo = reinterpret_cast<int32_t *>(src + lev_off_b);
  int32_t *dnl = reinterpret_cast<int32_t *>(src + nlev_off);
  const int tiles = std::max(tiles_for(g_), tiles_for_levels(g_));
  HIPCHK(d_cntp_.reserve((size_t)tiles * nslots_));
  zeroed_ptr_ = nullptr;
  zeroed_words_ = 0;
  for (int i = 0; i < 2; i++) ride_[i].dev = nullptr;
  cnt_on_host_ = false;
  HIPCHK(launch_sched(st_, reinterpret_cast<const uint2 *>(h_kstage_.p), (uint32_t)n_, (uint32_t)nops, dops, dlo, dnl, sw,
                      reinterpret_cast<uint2 *>(src)));
  if (timing_ >= 2) HIPCHK(hipEventRecord(ev2_, st_));   // (view kernel time = the refresh proper; k_sched reports its own: option sched_ticks)
  RefreshExtra x;
  x.n_lev_ptr = dnl;
  if (n_prune && plan_ride_) {
    // the walk plan of the sweep rides on the refresh launch (extra workgroups on the CUs the refresh leaves idle): no dispatch
    // of its own between refresh and scan.  (The fold of the refresh's mutation counts does NOT ride on the scan launch: its
    // 588 000 strided 4-byte reads beside the scan's first waves made the scan 60 us longer -- measured, dropped.)
    x.wp_kids = reinterpret_cast<const uint2 *>(src);
    x.wp_n = (uint32_t)n_;
    x.wp_desc = d_walk_.p;
    x.wp_hdr = sw.hdr_dev;
    x.wp_prog = d_prog_.p;
    x.wp_out = d_out();
    x.wp_max_parts = (uint32_t)(8 * n_prune);
  }
  HIPCHK(launch_newview_levels(st_, g_, d_vec_, dops, dlo, 1, d_cntp_.p, (uint32_t)nslots_, d_cnt(), nullptr, x));
  stats.view_launches++;
  HIPCHK(launch_cntsum(st_, g_, dops, (int)nops, d_cntp_.p, (uint32_t)nslots_, d_cnt(), tiles_for_levels(g_)));
  if (timing_ >= 2) { HIPCHK(hipEventRecord(ev3_, st_)); view_events_pending_ = true; }
  cnt_copy_pending_ = true;
  // what the host still needs is a dependency order for the subtree scores (finish_views): the views looking away from
  // start_, children first, then the ones looking towards it, parents first -- while the device works
  std::vector<int> &all = sv_all_, &stack = sv_stack_;
  all.clear();
  stack.clear();
  stack.push_back(back_[start_]);
  while (!stack.empty()) {
    const int r = stack.back();
    stack.pop_back();
    if (tip(r)) continue;
    all.push_back(r);
    stack.push_back(back_[nx(nx(r))]);
    stack.push_back(back_[nx(r)]);
  }
  upd_order_.resize(nops);
  size_t at = 0;
  if (3 * all.size() != nops) { set_error("refresh schedule: the tree is not complete"); return MPF_E_STATE; }
  for (size_t i = all.size(); i-- > 0;) upd_order_[at++] = all[i];
  for (int u : all) { upd_order_[at++] = nx(u); upd_order_[at++] = nx(nx(u)); }
  for (int r : upd_order_) valid_[r] = 1;
  all_invalid_ = false;
  n_invalid_ = 0;
  views_valid_ = true;
  pending_scores_ = true;
  sched_cache_valid_ = (plan_cache_ & 2) != 0;
  sched_gen_++;
  if (n_prune) dsw_sched_gen_ = sched_gen_;       // the sweep descriptors belong to this schedule's topology
  sc_nlev_off_ = nlev_off;                         // (also where the diagnostics of the last k_sched launch are read from)
  sched_on_dev_ = true;
  if (sched_cache_valid_) {
    sc_nops_ = nops;
    sc_maxlev_ = -1;
    sc_ops_off_ = ops_off;
    sc_lev_off_b_ = lev_off_b;
    sc_order_ = upd_order_;
  }
  stats.newview_ops += nops;
  stats.algorithmic_bytes += (uint64_t)nops * 3u * (uint64_t)g_.S * (uint64_t)Wref_ * 4u;
  return MPF_OK;
}

int Engine::schedule_views(const std::vector<int> *roots)
{
  if (!have_tree_) { set_error("no tree set"); return MPF_E_STATE; }
  ScopedMs timer(stats.host_views_ms_total);
  if (!roots && all_invalid_ && sched_cache_valid_ && !kids_dirty_ && !sankoff_) {
    // the same topology as when the last from-scratch schedule was built (the tree was handed over again, re-weighted or
    // re-evaluated): its ops, level offsets and the topology array are still on the device -- launch, nothing else
    const uint8_t *src = d_vstage_.p;
    const NvOp *dops = reinterpret_cast<const NvOp *>(src + sc_ops_off_);
    const int32_t *dlo = reinterpret_cast<const int32_t *>(src + sc_lev_off_b_);
    const size_t nops = sc_nops_;
    const int tiles = std::max(tiles_for(g_), tiles_for_levels(g_));
    HIPCHK(d_cntp_.reserve((size_t)tiles * nslots_));
    zeroed_ptr_ = nullptr;
    zeroed_words_ = 0;
    for (int i = 0; i < 2; i++) ride_[i].dev = nullptr;
    cnt_on_host_ = false;
    if (timing_ >= 2) HIPCHK(hipEventRecord(ev2_, st_));
    RefreshExtra x;
    if (sc_maxlev_ < 0) x.n_lev_ptr = reinterpret_cast<const int32_t *>(src + sc_nlev_off_);      // (made by k_sched: the count lives there)
    HIPCHK(launch_newview_levels(st_, g_, d_vec_, dops, dlo, sc_maxlev_ < 0 ? 1 : sc_maxlev_, d_cntp_.p, (uint32_t)nslots_, d_cnt(), nullptr, x));
    stats.view_launches++;
    HIPCHK(launch_cntsum(st_, g_, dops, (int)nops, d_cntp_.p, (uint32_t)nslots_, d_cnt(), tiles_for_levels(g_)));
    if (timing_ >= 2) { HIPCHK(hipEventRecord(ev3_, st_)); view_events_pending_ = true; }
    cnt_copy_pending_ = true;
    if (!(g_.vw == 1 && g_.nv_pipe)) shadow_ok_ = false;          // (another refresh kernel than k_newview_wgq: row-major store only)
    upd_order_ = sc_order_;
    for (int r : upd_order_) valid_[r] = 1;
    all_invalid_ = false;
    n_invalid_ = 0;
    views_valid_ = true;
    pending_scores_ = true;
    stats.newview_ops += nops;
    stats.algorithmic_bytes += (uint64_t)nops * 3u * (uint64_t)g_.S * (uint64_t)Wref_ * 4u;
    return MPF_OK;
  }
  if (!roots && dev_sched_usable()) return schedule_views_dev(0);
  // scratch vectors are members: this runs once per scan batch, allocations would show
  std::vector<int> &all = sv_all_;
  all.clear();
  if (!roots) {
    // every record of the component containing start_
    std::vector<int> &stack = sv_stack_;
    std::vector<char> &seen = sv_seen_;
    stack.clear();
    seen.assign(2 * (size_t)n_ + 1, 0);
    stack.push_back(back_[start_]);
    while (!stack.empty()) {
      const int r = stack.back();                  // the record by which the node is entered: it faces the root (start_)
      stack.pop_back();
      if (r < 0 || tip(r) || seen[num(r)]) continue;
      seen[num(r)] = 1;
      all.push_back(r);
      all.push_back(nx(r));
      all.push_back(nx(nx(r)));
      stack.push_back(back_[nx(nx(r))]);
      stack.push_back(back_[nx(r)]);
    }
    roots = &all;
  }
  // closure of invalid inputs, post-order, with dependency levels (epoch-stamped scratch arrays)
  epoch_++;
  std::vector<int> &order = sv_order_;
  std::vector<std::pair<int, int>> &stack = sv_pairs_;
  order.clear();
  stack.clear();
  auto lev_of = [&](int r) { return (tip(r) || lev_epoch_[r] != epoch_) ? 0 : lev_[r]; };
  const bool from_scratch = roots == &all && all_invalid_;
  if (from_scratch) {
    // nothing is valid (new topology): two sweeps over the tree rooted at start_ instead of the generic closure.
    // `all` lists the nodes in preorder, three records each, the first one facing the root.
    // views looking away from the root, children first: level = height
    for (size_t i = all.size(); i >= 3; i -= 3) {
      const int u = all[i - 3];
      const int a = back_[nx(u)], b = back_[nx(nx(u))];
      lev_[u] = 1 + std::max(tip(a) ? 0 : lev_[a], tip(b) ? 0 : lev_[b]);
      lev_epoch_[u] = epoch_;
      order.push_back(u);
    }
    // views looking towards the root, parents first: inputs are the parent's view towards us and the sibling's subtree
    for (size_t i = 0; i < all.size(); i += 3) {
      const int u = all[i], r1 = nx(u), r2 = nx(r1);
      const int p = back_[u], c1 = back_[r1], c2 = back_[r2];
      const int lp = tip(p) ? 0 : lev_[p];
      lev_[r1] = 1 + std::max(lp, tip(c2) ? 0 : lev_[c2]);
      lev_[r2] = 1 + std::max(lp, tip(c1) ? 0 : lev_[c1]);
      lev_epoch_[r1] = lev_epoch_[r2] = epoch_;
      order.push_back(r1);
      order.push_back(r2);
    }
  }
  for (int r0 : *roots) {
    if (from_scratch) break;
    if (r0 < 0 || tip(r0) || valid_[r0] || lev_epoch_[r0] == epoch_) continue;
    stack.emplace_back(r0, 0);
    while (!stack.empty()) {
      auto &top = stack.back();
      const int r = top.first;
      const int a = back_[nx(r)], b = back_[nx(nx(r))];
      if (top.second == 0) {
        top.second = 1;
        if (!tip(a) && !valid_[a] && lev_epoch_[a] != epoch_) { stack.emplace_back(a, 0); continue; }
      }
      if (top.second == 1) {
        top.second = 2;
        if (!tip(b) && !valid_[b] && lev_epoch_[b] != epoch_) { stack.emplace_back(b, 0); continue; }
      }
      if (lev_epoch_[r] != epoch_) {
        lev_[r] = 1 + std::max(lev_of(a), lev_of(b));
        lev_epoch_[r] = epoch_;
        order.push_back(r);
      }
      stack.pop_back();
    }
  }
  const size_t nops = order.size();
  const bool full = roots == &all;
  if (kids_host_.size() != nslots_) kids_host_.assign(nslots_, make_uint2(0u, 0u));
  // chained refresh for the incremental case (few ops, deep and narrow: paths away from an edit); a refresh of most of the
  // tree is wide, the level kernel's two-ops-in-flight loop suits it and cutting it into chains would cost the host more
  // than it saves the device
  const bool chains = views_mode_ == 2 && !sankoff_ && nops > 0 && (long)nops <= chain_max_ops_ && g_.S * g_.vw <= 8;   // (wider tiles would not fit four register sets)
  // ... and then the kernel reads its few KB of input (ops, offsets, topology updates) straight from the pinned staging
  // buffer: no copy dispatch in front of it
  const bool direct = chains && kids_dev_ready_ && (kids_dirty_ ? roots->size() + nops : kids_list_.size()) <= 4096;
  kid_upd_.clear();
  if (!kids_dirty_ && !kids_list_.empty()) {
    for (int r : kids_list_) {
      if (back_[r] < 0) continue;
      const uint2 k = make_uint2(slot(back_[nx(r)]), slot(back_[nx(nx(r))]));
      kids_host_[slot(r)] = k;
      if (direct) { kid_upd_.push_back(slot(r)); kid_upd_.push_back(k.x); kid_upd_.push_back(k.y); }
    }
    kids_list_.clear();
    kids_upload_ = !direct;                        // the mirror changed: a non-direct refresh uploads it whole
  }
  if (kids_dirty_) {
    // topology for the device-walked scans: kids[cid] = the two records behind an inner record
    auto put = [&](int r) {
      const uint2 k = make_uint2(slot(back_[nx(r)]), slot(back_[nx(nx(r))]));
      kids_host_[slot(r)] = k;
      if (direct) { kid_upd_.push_back(slot(r)); kid_upd_.push_back(k.x); kid_upd_.push_back(k.y); }
    };
    for (int r : *roots)
      if (r >= 0 && !tip(r)) put(r);
    if (!full)                                     // (a full refresh's roots contain every op)
      for (int r : order) put(r);
  }
  int maxlev = 0;
  for (int r : order) maxlev = std::max(maxlev, lev_[r]);
  if (chains) build_chains(order);                 // -> ch_ops_ (op order), ch_off_ (per level and wave), ch_levels_
  // staging layout: [kids][ops][level offsets], one upload
  const size_t kids_bytes = nslots_ * sizeof(uint2);
  const size_t ops_off = (kids_bytes + 15) & ~(size_t)15;
  const size_t lev_off_b = ops_off + ((nops * sizeof(NvOp) + 15) & ~(size_t)15);
  const size_t n_off = chains ? ch_off_.size() : (size_t)maxlev + 2;
  const size_t total_b = lev_off_b + ((n_off * sizeof(int32_t) + 15) & ~(size_t)15);
  const size_t upd_b = (kid_upd_.size() * sizeof(uint32_t) + 15) & ~(size_t)15;
  // the following scan's input rides along (Fitch refresh kernels also clear its outputs); not without a launch
  const bool can_ride = !sankoff_ && nops > 0 && views_mode_ >= 1;
  zeroed_ptr_ = nullptr;                           // (a promise from an earlier refresh that nobody collected is void)
  zeroed_words_ = 0;
  size_t ride_off[2] = {0, 0}, tail = total_b + upd_b;
  for (int i = 0; i < 2; i++) {
    ride_[i].dev = nullptr;
    if (can_ride && ride_[i].src && ride_[i].bytes) { ride_off[i] = tail; tail += (ride_[i].bytes + 15) & ~(size_t)15; }
  }
  HIPCHK(h_vstage_.reserve(tail));
  for (int i = 0; i < 2; i++)
    if (ride_off[i]) std::memcpy(h_vstage_.p + ride_off[i], ride_[i].src, ride_[i].bytes);
  if (direct) {
    if (!kid_upd_.empty()) std::memcpy(h_vstage_.p + total_b, kid_upd_.data(), kid_upd_.size() * sizeof(uint32_t));
    HIPCHK(d_cstage_.reserve(tail - ops_off));
  } else {
    HIPCHK(d_vstage_.reserve(tail));               // (may move: the whole topology is uploaded again below)
    std::memcpy(h_vstage_.p, kids_host_.data(), kids_bytes);
  }
  NvOp *hops = reinterpret_cast<NvOp *>(h_vstage_.p + ops_off);
  int32_t *lo = reinterpret_cast<int32_t *>(h_vstage_.p + lev_off_b);   // lo[0..maxlev]: offsets of levels 1..maxlev
  if (chains) {
    upd_order_.resize(nops);
    for (size_t at = 0; at < nops; at++) {
      const ChainOp &c = ch_ops_[at];
      const int r = c.rec;
      NvOp &o = hops[at];
      o.dst = slot(r);
      o.a = c.other < 0 ? slot(back_[nx(r)]) : 0xFFFFFFFFu;        // a link takes the previous result from registers
      o.b = c.other < 0 ? slot(back_[nx(nx(r))]) : slot(c.other);
      o.pad = (uint32_t)r;
      upd_order_[at] = r;
    }
    std::memcpy(lo, ch_off_.data(), ch_off_.size() * sizeof(int32_t));
  } else if (nops) {
    for (int l = 0; l <= maxlev + 1; l++) lo[l] = 0;
    for (int r : order) lo[lev_[r]]++;
    int acc = 0;
    for (int l = 1; l <= maxlev; l++) { const int c = lo[l]; lo[l] = acc; acc += c; }
    lo[0] = 0;
    std::vector<int> &fill = sv_fill_;
    fill.assign(lo, lo + maxlev + 1);
    upd_order_.resize(nops);
    for (int r : order) {
      const int at = fill[lev_[r]]++;
      NvOp &o = hops[at];
      o.dst = slot(r);
      o.a = slot(back_[nx(r)]);
      o.b = slot(back_[nx(nx(r))]);
      o.pad = (uint32_t)r;
      upd_order_[(size_t)at] = r;
    }
    for (int l = 1; l <= maxlev; l++) lo[l - 1] = lo[l];
    lo[maxlev] = (int32_t)nops;
  }
  if (direct) {
    // one small upload (ops, offsets, topology deltas, scan descriptors); the topology array itself stays where it is
    // (letting the kernels read the pinned buffer itself instead costs more than this copy: 25 workgroups fetching their
    //  descriptors over PCIe -- a C3 climb takes 0.35 s instead of 0.29 s)
    HIPCHK(hipMemcpyAsync(d_cstage_.p, h_vstage_.p + ops_off, tail - ops_off, hipMemcpyHostToDevice, st_));
    if (full) kids_dirty_ = false;
  } else if (kids_dirty_ || kids_upload_ || nops) {
    kids_upload_ = false;
    const size_t up = nops ? tail : kids_bytes;
    HIPCHK(hipMemcpyAsync(d_vstage_.p, h_vstage_.p, up, hipMemcpyHostToDevice, st_));
    kids_dev_ready_ = true;
    if (full) kids_dirty_ = false;
  }
  if (nops == 0) {
    if (!direct) sched_cache_valid_ = false;       // (d_vstage_ may have been rewritten above)
    if (full) { n_invalid_ = 0; views_valid_ = true; }
    return MPF_OK;
  }
  const uint8_t *src = direct ? d_cstage_.p - ops_off : d_vstage_.p;      // (same offsets in both layouts)
  const NvOp *dops = reinterpret_cast<const NvOp *>(src + ops_off);
  const int32_t *dlo = reinterpret_cast<const int32_t *>(src + lev_off_b);
  const int tiles = std::max(tiles_for(g_), tiles_for_levels(g_));
  HIPCHK(d_cntp_.reserve((size_t)tiles * nslots_));
  if (timing_ >= 2) HIPCHK(hipEventRecord(ev2_, st_));
  // the per-tile mutation counts are folded by the refresh kernel's last workgroup when there are few ops, by a separate
  // chip-wide launch when there are many (one workgroup would need longer than the launch costs)
  const bool fold_inside = views_mode_ >= 1 && !sankoff_ && nops <= 512;   // (independent of chain_max_ops_)
  RefreshExtra x;
  cnt_on_host_ = false;
  if (fold_inside && want_host_results_) { x.cnt_host = h_cnt(); cnt_on_host_ = true; }   // small batch: counts land in the host mirror
  for (int i = 0; i < 2; i++)
    if (ride_off[i]) ride_[i].dev = src + ride_off[i];
  bool can_ride_used = ride_off[0] || ride_off[1];
  if (can_ride && zero_req_ptr_) {
    x.zero_ptr = zero_req_ptr_;                   // the outputs of the scan that follows, cleared by the refresh launch
    x.zero_words = (uint32_t)zero_req_words_;
    zeroed_ptr_ = zero_req_ptr_;
    zeroed_words_ = zero_req_words_;
    can_ride_used = true;
  }
  if (chains) {
    if (direct) {
      x.kid_upd = reinterpret_cast<const uint32_t *>(src + total_b);
      x.n_kid_upd = (int)(kid_upd_.size() / 3);
      x.kids = reinterpret_cast<uint2 *>(d_vstage_.p);
    }
    HIPCHK(launch_newview_chains(st_, g_, d_vec_, dops, dlo, ch_levels_, (int)nops, d_cntp_.p, (uint32_t)nslots_, d_cnt(), fold_inside ? d_done_.p : nullptr, x));
    stats.view_launches++;
  } else if (views_mode_ >= 1) {
    // narrow levels (the partial trees of the addition phase, small refreshes): fewer waves per workgroup -- a wave of the
    // one-word-per-lane kernel takes 64 / tile ops per round, so 2 x (ops per level) x tile / 64 waves cover a level in two rounds
    if (nv_waves_ > 0) x.waves_hint = nv_waves_;
    else if (nv_waves_ == 0 && maxlev > 0 && g_.vw == 1 && g_.nv_pipe && !sankoff_) {
      const long per_level = ((long)nops + maxlev - 1) / maxlev;
      const long need = (2 * per_level * newview_tile(g_) + 63) / 64;
      x.waves_hint = need <= 2 ? 2 : need <= 4 ? 4 : need <= 8 ? 8 : 16;
    }
    HIPCHK(launch_newview_levels(st_, g_, d_vec_, dops, dlo, maxlev, d_cntp_.p, (uint32_t)nslots_, d_cnt(), fold_inside ? d_done_.p : nullptr, x));
    stats.view_launches++;
  } else {
    for (int l = 0; l < maxlev; l++) {
      HIPCHK(launch_newview(st_, g_, d_vec_, dops + lo[l], lo[l + 1] - lo[l], d_cntp_.p, (uint32_t)nslots_));
      stats.view_launches++;
    }
  }
  if (!fold_inside)
    // (no host mirror here: hundreds of scattered 4-byte writes over PCIe cost more than the one copy-back they would save)
    HIPCHK(launch_cntsum(st_, g_, dops, (int)nops, d_cntp_.p, (uint32_t)nslots_, d_cnt(),
                         (views_mode_ >= 1 && !chains) ? tiles_for_levels(g_) : 0));   // rows of cntp the refresh kernel wrote
  if (timing_ >= 2) { HIPCHK(hipEventRecord(ev3_, st_)); view_events_pending_ = true; }
  cnt_copy_pending_ = true;                 // copied back together with the scan results (or by update_views)
  {
    // shadow_ok_ = "every valid vector has its word-major copy": true whenever nothing is valid (invalidate_all / _vectors),
    // kept by k_newview_wgq and k_newview_chain, which write both layouts, lost when any other kernel writes vectors
    const bool both = views_mode_ >= 1 && !sankoff_ && g_.vw == 1 && g_.shoff != 0 && (chains || g_.nv_pipe);
    if (!both) shadow_ok_ = false;
  }
  for (int r : order) valid_[r] = 1;
  all_invalid_ = false;
  if (full) { n_invalid_ = 0; views_valid_ = true; }
  else if (n_invalid_ > 0) {
    n_invalid_ -= (long)nops;
    if (n_invalid_ <= 0) {
      n_invalid_ = 0;
      views_valid_ = true;
      if (check_counts_)                          // self-check of the bookkeeping: every vector of the tree must be valid now
        for (size_t r = 3 * ((size_t)n_ + 1); r < back_.size(); r++)
          if (back_[r] >= 0 && !valid_[r]) { set_error("view bookkeeping: invalid vector left behind"); return MPF_E_STATE; }
    }
  }
  pending_scores_ = true;
  if (!direct) {
    // d_vstage_ was (re)written: it holds a reusable schedule only if this was the from-scratch refresh of the whole tree
    // on the level kernel with the per-tile counts folded by the separate launch (the shape the fast path above replays)
    sched_cache_valid_ = (plan_cache_ & 2) && from_scratch && full && !chains && views_mode_ >= 1 && !fold_inside && !sankoff_ && !can_ride_used;
    sched_gen_++;
    sched_on_dev_ = false;                         // (d_vstage_ was rewritten by the host's schedule)
    if (sched_cache_valid_) {
      sc_nops_ = nops;
      sc_maxlev_ = maxlev;
      sc_ops_off_ = ops_off;
      sc_lev_off_b_ = lev_off_b;
      sc_order_ = upd_order_;
    }
  }
  stats.newview_ops += nops;
  dbg_levels_ += (uint64_t)(chains ? ch_levels_ : maxlev);
  stats.algorithmic_bytes += (uint64_t)nops * 3u * (uint64_t)g_.S * (uint64_t)Wref_ * 4u;
  return MPF_OK;
}

// subtree scores in dependency order (reference: tr->parsimonyScore[p] = total + score[q] + score[r], :874);
// call only after the stream has been synchronised
void Engine::finish_views()
{
  if (view_events_pending_) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, ev2_, ev3_) == hipSuccess) stats.view_kernel_ms_total += ms;
    view_events_pending_ = false;
  }
  if (!pending_scores_) return;
  for (int r : upd_order_) {
    const int a = back_[nx(r)], b = back_[nx(nx(r))];
    // weighted mode: parsimonyScore[p] is the node's own unweighted minimum sum only (reference :490, :544-548)
    sc_[r] = sankoff_ ? h_cnt()[slot(r)] : h_cnt()[slot(r)] + (tip(a) ? 0u : sc_[a]) + (tip(b) ? 0u : sc_[b]);
  }
  pending_scores_ = false;
}

int Engine::update_views()
{
  int rc = schedule_views(nullptr);
  if (rc) return rc;
  if (cnt_copy_pending_) {
    HIPCHK(hipMemcpyAsync(h_cnt(), d_cnt(), nslots_ * sizeof(uint32_t), hipMemcpyDeviceToHost, st_));
    cnt_copy_pending_ = false;
  }
  HIPCHK(hipStreamSynchronize(st_));
  finish_views();
  return MPF_OK;
}

// the vectors the scans of prune record p will read: pruned-subtree vector, the two gap ends, and every
// record the DFS of addTraverseParsimony visits (both sides)
void Engine::collect_scan_roots(int p, int mintrav, int maxtrav, std::vector<int> &roots) const
{
  (void)mintrav;
  if (maxtrav > ntips_ - 3) maxtrav = ntips_ - 3;
  if (maxtrav < 1) return;
  struct Fr { int q, d; };
  std::vector<Fr> st;
  const int xs[2] = {p, back_[p]};
  roots.push_back(p);            // the scores of both ends of the prune branch give the base length
  roots.push_back(back_[p]);
  for (int k = 0; k < 2; k++) {
    const int x = xs[k];
    if (tip(x)) continue;
    roots.push_back(back_[x]);
    const int x1 = back_[nx(x)], x2 = back_[nx(nx(x))];
    roots.push_back(x1);
    roots.push_back(x2);
    for (int side = 0; side < 2; side++) {
      const int a = side ? x2 : x1;
      if (tip(a)) continue;
      st.push_back(Fr{back_[nx(nx(a))], 1});
      st.push_back(Fr{back_[nx(a)], 1});
      while (!st.empty()) {
        const Fr f = st.back();
        st.pop_back();
        roots.push_back(f.q);
        if (!tip(f.q) && f.d < maxtrav) {
          st.push_back(Fr{back_[nx(nx(f.q))], f.d + 1});
          st.push_back(Fr{back_[nx(f.q)], f.d + 1});
        }
      }
    }
  }
}

// Fitch length of the current tree = score(x) + score(back x) + #empty(vec x, vec back x) on any branch;
// on the branch start--back[start] (start is a tip) that is score(back[start]) + one evaluate.
int Engine::tree_length(uint32_t *len)
{
  if (!views_valid_) { int rc = update_views(); if (rc) return rc; }
  // weighted mode: evaluateParsimony(tr->start) = min_x(left[x] + m(right)[x]) with left = the far end of the edge, right = the
  // record handed over (reference :880-961) -- the order matters when the matrix is not symmetric
  const int a = sankoff_ ? back_[start_] : start_, b = sankoff_ ? start_ : back_[start_];
  EvOp op{slot(a), slot(b), 0, 0};
  HIPCHK(d_evops_.reserve(1));
  HIPCHK(reserve_results(1));
  HIPCHK(hipMemcpyAsync(d_evops_.p, &op, sizeof(op), hipMemcpyHostToDevice, st_));
  HIPCHK(hipMemsetAsync(d_out(), 0, clear_words(1) * sizeof(uint32_t), st_));
  HIPCHK(launch_evaluate(st_, g_, d_vec_, d_evops_.p, 1, d_out()));
  HIPCHK(hipMemcpyAsync(h_out(), d_out(), sizeof(uint32_t), hipMemcpyDeviceToHost, st_));
  HIPCHK(hipStreamSynchronize(st_));
  tree_len_ = sankoff_ ? h_out()[0] : (tip(a) ? 0u : sc_[a]) + (tip(b) ? 0u : sc_[b]) + h_out()[0];
  *len = tree_len_;
  return MPF_OK;
}

int Engine::score_tree(uint32_t *score)
{
  if (!have_tree_) { set_error("no tree set"); return MPF_E_STATE; }
  node_rectifier();
  invalidate_vectors();                      // evaluateParsimony(start, PLL_TRUE): every vector again, same topology
  return tree_length(score);
}

// ---- SPR scan programs ------------------------------------------------------------------------
// addTraverseParsimony (reference sprparsimony.cpp:2208-2218) turned into a list of chain steps
void Engine::add_traverse(int q, int sib, int depth, int mintrav, int maxtrav, ScanPlan &plan)
{
  const bool test = (--mintrav <= 0);
  ScanOp o;
  o.own = slot(q);
  o.sib = slot(sib);
  o.meta = (uint32_t)depth | ((test ? 1u : 0u) << 8) | ((uint32_t)SCAN_CHAIN << 16);
  o.out = test ? prog_out_ : 0u;
  prog_ops_.push_back(o);
  prog_max_depth_ = std::max(prog_max_depth_, depth);
  if (test) plan.cands.push_back(Candidate{q, prog_out_++});
  if (!tip(q) && --maxtrav > 0) {
    const int c1 = back_[nx(q)], c2 = back_[nx(nx(q))];
    add_traverse(c1, c2, depth + 1, mintrav, maxtrav, plan);
    add_traverse(c2, c1, depth + 1, mintrav, maxtrav, plan);
  }
}

// rearrangeParsimony (reference sprparsimony.cpp:2259-2376) as two scan programs (p side, q side)
int Engine::plan_scan(int p, int mintrav, int maxtrav, ScanPlan &plan)
{
  plan.rec = p;
  plan.walked = false;
  plan.cands.clear();
  plan.n_p = plan.n_total = 0;
  if (maxtrav > ntips_ - 3) maxtrav = ntips_ - 3;
  if (mintrav != 1) { set_error("mintrav must be 1 (reference asserts it, sprparsimony.cpp:2280)"); return MPF_E_INVALID; }
  // (any radius: beyond the levels the kernels keep in registers -- Fitch 12, weighted 12 for DNA and 6 otherwise -- the levels'
  //  vectors live in HBM scratch, k_scan_deep / k_snk_scan_deep)
  if (maxtrav > 255) { set_error("maxtrav above the supported chain depth"); return MPF_E_UNSUPPORTED; }
  const int q = back_[p];
  plan.base = sankoff_ ? 0u : (tip(p) ? 0u : sc_[p]) + (tip(q) ? 0u : sc_[q]);   // weighted: the kernel returns full lengths
  if (maxtrav < mintrav) return MPF_OK;
  auto one_side = [&](int x, int mt) {
    const int x1 = back_[nx(x)], x2 = back_[nx(nx(x))];
    ScanHdr h;
    h.op_begin = (uint32_t)prog_ops_.size();
    h.s_slot = slot(back_[x]);
    h.pad = 0;
    auto side = [&](int a, int other) {
      if (tip(a)) return;
      ScanOp r;
      r.own = slot(other);
      r.sib = 0;
      r.meta = (uint32_t)SCAN_ROOT << 16;
      r.out = 0;
      prog_ops_.push_back(r);
      const int c1 = back_[nx(a)], c2 = back_[nx(nx(a))];
      add_traverse(c1, c2, 1, mt, maxtrav, plan);
      add_traverse(c2, c1, 1, mt, maxtrav, plan);
    };
    side(x1, x2);
    side(x2, x1);
    h.op_end = (uint32_t)prog_ops_.size();
    if (h.op_end > h.op_begin) prog_hdr_.push_back(h);
  };
  if (!tip(p)) {
    const int p1 = back_[nx(p)], p2 = back_[nx(nx(p))];
    if (!tip(p1) || !tip(p2)) one_side(p, mintrav);
  }
  plan.n_p = (int)plan.cands.size();
  if (!tip(q) && maxtrav > 0) {
    const int q1 = back_[nx(q)], q2 = back_[nx(nx(q))];
    if ((!tip(q1) && (!tip(back_[nx(q1)]) || !tip(back_[nx(nx(q1))]))) ||
        (!tip(q2) && (!tip(back_[nx(q2)]) || !tip(back_[nx(nx(q2))]))))
      one_side(q, mintrav > 2 ? mintrav : 2);
  }
  plan.n_total = (int)plan.cands.size();
  return MPF_OK;
}

// spin on a flag word a kernel raises in pinned host memory; false after 50 ms of wall clock (the caller then synchronises
// the stream, which also surfaces a failed launch at once instead of after a long spin)
static inline void cpu_relax()
{
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#elif defined(__aarch64__)
  asm volatile("yield");
#endif
}
// How many host threads of this process are waiting for the device right now, against the CPU time the process may use (the
// cgroup's quota: a container that is granted 16 of 256 cores is THROTTLED as a whole once its spinning threads have burnt the
// period's quota -- 32 engines polling at full speed stall each other and the HIP runtime's own threads).  Crowded = more
// waiters than the budget: they sleep between looks instead of spinning.
namespace {
std::atomic<int> g_waiters{0};
int cpu_budget()
{
  static const int budget = [] {
    int b = (int)std::thread::hardware_concurrency();
    if (b <= 0) b = 1;
    if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {                      // cgroup v2: "<quota|max> <period>"
      char q[32] = {0};
      long period = 0;
      if (std::fscanf(f, "%31s %ld", q, &period) == 2 && period > 0 && std::strcmp(q, "max") != 0) b = std::min(b, (int)std::max(1L, std::atol(q) / period));
      std::fclose(f);
    } else {
      long quota = -1, period = 0;                                                  // cgroup v1
      if (FILE *g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (std::fscanf(g, "%ld", &quota) != 1) quota = -1; std::fclose(g); }
      if (FILE *g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (std::fscanf(g, "%ld", &period) != 1) period = 0; std::fclose(g); }
      if (quota > 0 && period > 0) b = std::min(b, (int)std::max(1L, quota / period));
    }
    return b;
  }();
  return budget;
}
}  // namespace
Engine::WaitScope::WaitScope() { g_waiters.fetch_add(1, std::memory_order_relaxed); }
Engine::WaitScope::~WaitScope() { g_waiters.fetch_sub(1, std::memory_order_relaxed); }
void Engine::wait_pause()
{
  if (g_waiters.load(std::memory_order_relaxed) > cpu_budget()) {
    struct timespec ts = {0, 20000};               // (20 us + the timer slack: the core goes to somebody who has work)
    nanosleep(&ts, nullptr);
  } else {
    sched_yield();                                 // several engines per GPU poll from threads that may share cores
  }
}
bool Engine::wait_host_flag(const uint32_t *flag)
{
  std::chrono::steady_clock::time_point t0;
  WaitScope waiting;
  for (long spin = 0;; spin++) {
    if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == 1u) return true;
    cpu_relax();
    if ((spin & 255) == 255) {
      wait_pause();
      const auto now = std::chrono::steady_clock::now();
      if (spin == 255) t0 = now;
      else if (now - t0 > std::chrono::milliseconds(50)) return false;
    }
  }
}

int Engine::run_scans(std::vector<ScanPlan> &plans, std::vector<uint32_t> &out_host)
{
  (void)plans;
  ScopedMs timer(stats.host_scan_ms_total);
  const size_t nops = prog_ops_.size(), nh = prog_hdr_.size(), nout = prog_out_;
  out_host.assign(nout, 0);
  if (scan_vals_) vals_rows_ = (uint32_t)nout;
  if (nh == 0 || nout == 0) { prog_ops_.clear(); prog_hdr_.clear(); prog_out_ = 0; prog_max_depth_ = 0; return MPF_OK; }
  HIPCHK(d_scanops_.reserve(nops));
  HIPCHK(d_scanhdr_.reserve(nh));
  HIPCHK(reserve_results(nout));
  // the program may have gone up with the refresh in front (addition_phase), which then cleared the outputs as well
  const ScanOp *dprog = static_cast<const ScanOp *>(ride_[0].dev);
  const ScanHdr *dhdr = static_cast<const ScanHdr *>(ride_[1].dev);
  ride_[0].dev = ride_[1].dev = nullptr;
  if (!dprog || !dhdr) {
    HIPCHK(hipMemcpyAsync(d_scanops_.p, prog_ops_.data(), nops * sizeof(ScanOp), hipMemcpyHostToDevice, st_));
    HIPCHK(hipMemcpyAsync(d_scanhdr_.p, prog_hdr_.data(), nh * sizeof(ScanHdr), hipMemcpyHostToDevice, st_));
    dprog = d_scanops_.p;
    dhdr = d_scanhdr_.p;
  }
  if (!(zeroed_ptr_ == d_out() && zeroed_words_ >= clear_words(nout)))
    HIPCHK(hipMemsetAsync(d_out(), 0, clear_words(nout) * sizeof(uint32_t), st_));
  zeroed_ptr_ = nullptr;
  zeroed_words_ = 0;
  if (timing_) HIPCHK(hipEventRecord(ev0_, st_));
  const bool deep = prog_max_depth_ > scan_reg_depth(g_.S, sankoff_);
  if (deep && !g_.deep_scratch) {
    // per-level up-vectors of the scans' waves: 256 MB, the launches are cut to fit (launch_scan)
    HIPCHK(d_deep_.reserve(deep_scratch_words_));
    g_.deep_scratch = d_deep_.p;
    g_.deep_scratch_words = deep_scratch_words_;
  }
  const bool host_direct = want_host_results_ && !sankoff_ && !deep && nout <= 16384 && (cnt_on_host_ || !cnt_copy_pending_);
  if (host_direct) __atomic_store_n(h_out() + nout, 0u, __ATOMIC_RELAXED);       // the flag word behind the results
  uint16_t *vals = nullptr;
  uint32_t *vmax = nullptr;
  if (scan_vals_ && ufb_) {
    // weighted tracker: per-pattern lengths of every tentative tree, one more row for the current tree behind them
    HIPCHK(ufb_->vals.reserve((nout + 1) * (size_t)g_.Wp));
    HIPCHK(ufb_->vmax.reserve(4));                // (zeroed by the caller)
    vals = ufb_->vals.p;
    vmax = ufb_->vmax.p;
    vals_rows_ = (uint32_t)nout;
  }
  HIPCHK(launch_scan(st_, g_, d_vec_, dhdr, (int)nh, dprog, d_out(), prog_max_depth_, host_direct ? h_out() : nullptr, (uint32_t)nout,
                     d_done_.p + 16, vals, (uint32_t)g_.Wp, vmax));
  if (timing_) HIPCHK(hipEventRecord(ev1_, st_));
  if (host_direct) {           // the kernels wrote the host's result buffers themselves
    cnt_copy_pending_ = false;
  } else if (cnt_copy_pending_) {     // a refresh was enqueued just before: bring its mutation counts back in the same copy
    HIPCHK(hipMemcpyAsync(h_cnt(), d_cnt(), (out_off() + nout) * sizeof(uint32_t), hipMemcpyDeviceToHost, st_));
    cnt_copy_pending_ = false;
  } else {
    HIPCHK(hipMemcpyAsync(h_out(), d_out(), nout * sizeof(uint32_t), hipMemcpyDeviceToHost, st_));
  }
  if (!(host_direct && host_poll_ && !timing_ && wait_host_flag(h_out() + nout))) HIPCHK(hipStreamSynchronize(st_));
  finish_views();
  float ms = 0;
  if (timing_ && hipEventElapsedTime(&ms, ev0_, ev1_) == hipSuccess) { stats.last_scan_kernel_ms = ms; stats.scan_kernel_ms_total += ms; }
  std::copy(h_out(), h_out() + nout, out_host.begin());
  stats.scan_launches++;
  size_t n_tests = nout;
  if (scan_vals_)                                // (weighted tracker: the slots reserved for the current tree are not insertion tests)
    for (const ScanPlan &pl : plans) n_tests -= pl.self_idx >= 0 ? 1u : 0u;
  stats.insertion_tests += n_tests;
  stats.algorithmic_bytes += (uint64_t)n_tests * 6u * (uint64_t)g_.S * (uint64_t)Wref_ * 4u;
  prog_ops_.clear();
  prog_hdr_.clear();
  prog_out_ = 0;
  prog_max_depth_ = 0;
  return MPF_OK;
}


// ---- device-walked scans -------------------------------------------------------------------
// the same enumeration as add_traverse, host side, only to NAME a candidate (needed when a move is accepted)
int Engine::candidate_record(const ScanPlan &plan, size_t c)
{
  if (!plan.walked) return plan.cands[c].q;
  std::vector<int32_t> q;
  if ((int)c < plan.n_p) {
    enumerate_side(plan.rec, 1, plan.maxtrav, q);
    return q[c];
  }
  enumerate_side(back_[plan.rec], plan.mintrav_q, plan.maxtrav, q);
  return q[c - (size_t)plan.n_p];
}

// rearrangeParsimony's applicability tests (reference sprparsimony.cpp:2304-2310, :2330-2347) decide which
// of the two scans of a prune record exist; the kernel does the rest
int Engine::plan_walk(int p, int mintrav, int maxtrav, ScanPlan &plan, bool split)
{
  plan.rec = p;
  plan.walked = true;
  plan.cands.clear();
  plan.n_p = plan.n_total = 0;
  plan.n_parts = plan.n_parts_p = 0;
  if (maxtrav > ntips_ - 3) maxtrav = ntips_ - 3;
  if (mintrav != 1) { set_error("mintrav must be 1 (reference asserts it, sprparsimony.cpp:2280)"); return MPF_E_INVALID; }
  if (maxtrav > 255) { set_error("maxtrav above the supported chain depth"); return MPF_E_UNSUPPORTED; }
  plan.maxtrav = maxtrav;
  const int q = back_[p];
  plan.base = 0;                                   // filled in after the refresh has been synchronised
  plan.self_idx = -1;
  if (n_walk_ == 0) walk_gen_++;                   // a new set of descriptors: whatever a cached sweep left on the device goes
  if (maxtrav < mintrav) return MPF_OK;
  if (scan_masks_) plan.self_idx = (int64_t)walk_out_++;      // the current tree's own saveCurrentTree call, :2285-2289
  if (maxtrav > 6) split = false;
  // candidate counts are known from the topology (count_visits), so the outputs of all parts are laid
  // out back to back.  whole scan = one part; split scan (latency, small batches) = four parts
  // (gap end x first-level child)
  bool oom = false;
  auto add = [&](int x, int mt, uint32_t side_mask, uint32_t child_mask, int count, int count_first_end) {
    if (h_walk_.cap < n_walk_ + 1) {
      PinBuf<WalkDesc> bigger;
      if (bigger.reserve(2 * (n_walk_ + 1) + 4096) != hipSuccess) { oom = true; return; }
      if (n_walk_) std::memcpy(bigger.p, h_walk_.p, n_walk_ * sizeof(WalkDesc));
      std::swap(bigger.p, h_walk_.p);
      std::swap(bigger.cap, h_walk_.cap);
    }
    WalkDesc &d = h_walk_.p[n_walk_];
    d.s_cid = slot(back_[x]);
    d.xa_cid = slot(back_[nx(x)]);
    d.xb_cid = slot(back_[nx(nx(x))]);
    d.trav = (uint32_t)mt | ((uint32_t)maxtrav << 8) | (side_mask << 16) | (child_mask << 18);
    d.out_base = walk_out_;
    d.pad0 = (uint32_t)count;
    d.pad1 = (uint32_t)count_first_end;            // candidates behind the first gap end: where the second end's indices start (k_walk_plan)
    d.pad2 = 0;
    plan.part_desc[plan.n_parts] = (int)n_walk_;
    plan.part_off[plan.n_parts] = walk_out_;
    plan.part_cnt[plan.n_parts] = count;
    plan.n_parts++;
    plan.n_total += count;
    walk_out_ += (uint32_t)count + (scan_masks_ ? 1u : 0u);     // masked scans: one extra slot per part (home edge)
    n_walk_++;
  };
  // N(q, maxtrav): straight from the sweep's dense table when it is current
  const bool dense = maxtrav >= 2 && maxtrav == vis_dense_m_ && visits_filled_epoch_ == topo_epoch_;
  auto cv = [&](int q) { return tip(q) ? 1 : dense ? vis_dense_[(size_t)slot(q)] : count_visits(q, maxtrav); };
  auto phase = [&](int x, int mt) {
    const int xs[2] = {back_[nx(x)], back_[nx(nx(x))]};
    const int skip = mt > 1 ? 1 : 0;                  // the q side does not test the first level (mintrav2 = 2)
    int cnt[2][2] = {{0, 0}, {0, 0}};
    for (int side = 0; side < 2; side++) {
      if (tip(xs[side])) continue;
      cnt[side][0] = cv(back_[nx(xs[side])]) - skip;
      cnt[side][1] = cv(back_[nx(nx(xs[side]))]) - skip;
    }
    const int total = cnt[0][0] + cnt[0][1] + cnt[1][0] + cnt[1][1];
    // throughput batches: a neighbourhood is one part unless it is long -- the waves of one launch should not differ
    // in length by two orders of magnitude (the longest ones would run on alone at the end)
    // (masked scans stay whole: every part costs a home row of the REPS product)
    if (!split && (total <= split_cands_ || maxtrav > 6 || scan_masks_)) { add(x, mt, 3u, 3u, total, cnt[0][0] + cnt[0][1]); return; }
    for (uint32_t side = 0; side < 2; side++) {
      if (tip(xs[side])) continue;
      add(x, mt, 1u << side, 1u, cnt[side][0], 0);
      add(x, mt, 1u << side, 2u, cnt[side][1], 0);
    }
  };
  if (!tip(p)) {
    const int p1 = back_[nx(p)], p2 = back_[nx(nx(p))];
    if (!tip(p1) || !tip(p2)) phase(p, mintrav);
  }
  plan.n_parts_p = plan.n_parts;
  plan.n_p = plan.n_total;
  if (!tip(q) && maxtrav > 0) {
    const int q1 = back_[nx(q)], q2 = back_[nx(nx(q))];
    if ((!tip(q1) && (!tip(back_[nx(q1)]) || !tip(back_[nx(nx(q1))]))) ||
        (!tip(q2) && (!tip(back_[nx(q2)]) || !tip(back_[nx(nx(q2))])))) {
      plan.mintrav_q = mintrav > 2 ? mintrav : 2;
      phase(q, plan.mintrav_q);
    }
  }
  if (oom) { set_error("pinned alloc failed"); return MPF_E_NOMEM; }
  return MPF_OK;
}

int Engine::run_walks(std::vector<ScanPlan> &plans, const uint32_t **out_host)
{
  ScopedMs timer(stats.host_scan_ms_total);
  const size_t nd = n_walk_, nout = walk_out_;
  *out_host = nullptr;
  int maxd = 0;
  for (size_t i = 0; i < nd; i++) maxd = std::max(maxd, (int)((h_walk_.p[i].trav >> 8) & 0xFFu));
  if (nd > 0) {
    HIPCHK(d_walk_.reserve(nd));
    HIPCHK(d_ncand_.reserve(nd));
    HIPCHK(h_ncand_.reserve(nd));
    HIPCHK(reserve_results(nout));
    // a small batch's descriptors went up with the refresh's own upload and its outputs were cleared by the refresh
    // kernel: no second copy and no memset dispatch on the critical path of a climb's batches
    const WalkDesc *descs = static_cast<const WalkDesc *>(ride_[0].dev);
    ride_[0].dev = nullptr;
    if (walk_dev_reuse_) descs = d_walk_.p;        // a cached sweep: descriptors (and the planned program) are still there
    // small batch: the kernels write the host's copies themselves (mutation counts: the refresh's fold; candidate costs:
    // the scan's last workgroup) -- no copy-back dispatch
    const bool host_direct = want_host_results_ && nout <= 16384 && !scan_masks_ && !check_counts_ && maxd <= kWalkMaxDepth && (cnt_on_host_ || !cnt_copy_pending_);
    if (maxd > kWalkMaxDepth && !g_.deep_scratch) {
      // parked up-vectors of the deep walks' waves: the launches are cut to fit (launch_scan_walk)
      HIPCHK(d_deep_.reserve(deep_scratch_words_));
      g_.deep_scratch = d_deep_.p;
      g_.deep_scratch_words = deep_scratch_words_;
    }
    // planned program (plan kernel + pipelined scan) for throughput batches; the device-walked kernel for the small,
    // latency-bound batches inside a climb (scan_prog 2: always planned), for masks, protein and radii above 6
    const bool prog = scan_prog_ > 0 && !scan_masks_ && scan_prog_supported(g_, maxd) && (scan_prog_ >= 2 || nd > (size_t)prog_min_descs_);
    const bool plan_now = prog && (!walk_dev_reuse_ || prog_gen_ != walk_gen_);
    const bool use_pmin = want_part_min_ && !host_direct && !scan_masks_ && !check_counts_ && nout > 16384;
    const bool parts_now = use_pmin && !(walk_dev_reuse_ && parts_gen_ == walk_gen_ && n_parts_dev_ > 0);
    if (parts_now) {
      // the (offset, count) list of the scan parts for k_part_min goes up in front of the scan, not between it and the reduction
      size_t np = 0;
      for (const ScanPlan &pl : plans) np += (size_t)pl.n_parts;
      HIPCHK(h_parts_.reserve(std::max<size_t>(np, 1)));
      n_parts_dev_ = 0;
      for (const ScanPlan &pl : plans)
        for (int pi = 0; pi < pl.n_parts; pi++) h_parts_.p[n_parts_dev_++] = make_uint2(pl.part_off[pi], (uint32_t)pl.part_cnt[pi]);
      HIPCHK(d_parts_.reserve(std::max<size_t>(n_parts_dev_, 1)));
      // (the pinned list is rewritten only by the next newly planned sweep, i.e. after this sweep's results have come back)
      HIPCHK(hipMemcpyAsync(d_parts_.p, h_parts_.p, n_parts_dev_ * sizeof(uint2), hipMemcpyHostToDevice, st_));
      parts_gen_ = walk_gen_;
    }
    if (!descs) { HIPCHK(hipMemcpyAsync(d_walk_.p, h_walk_.p, nd * sizeof(WalkDesc), hipMemcpyHostToDevice, st_)); descs = d_walk_.p; }
    const bool zeroed = zeroed_ptr_ == d_out() && zeroed_words_ >= clear_words(nout);
    const bool zero_in_plan = !zeroed && plan_now;           // the plan kernel clears the outputs: no memset dispatch
    if (!zeroed && !zero_in_plan) HIPCHK(hipMemsetAsync(d_out(), 0, clear_words(nout) * sizeof(uint32_t), st_));
    zeroed_ptr_ = nullptr;
    zeroed_words_ = 0;
    if (timing_) HIPCHK(hipEventRecord(ev0_, st_));
    uint32_t *mask_ptr = nullptr;
    uint2 *info_ptr = nullptr;
    if (scan_masks_) {
      int rc = ufb_reserve_scan(nout);
      if (rc) return rc;
      mask_ptr = ufb_->masks.p;
      info_ptr = ufb_->info.p;
      ufb_rows_ = (uint32_t)nout;
    }
    if (host_direct) __atomic_store_n(h_out() + nout, 0u, __ATOMIC_RELAXED);     // the flag word behind the results
    if (prog) {
      HIPCHK(d_prog_.reserve(scan_prog_bytes((int)nd)));
      // (a cached sweep re-uses the program only if one was planned for exactly these descriptors: the sweep may have run on
      //  the device-walked kernel when it was cached -- option prog_min_descs -- and other batches plan into the same buffer)
      if (plan_now) {
        prog_gen_ = walk_gen_;
        if (timing_) HIPCHK(hipEventRecord(ev4_, st_));
        HIPCHK(launch_walk_plan(st_, d_kids(), n_, descs, (int)nd, d_prog_.p, zero_in_plan ? d_out() : nullptr,
                                zero_in_plan ? (uint32_t)clear_words(nout) : 0u));
        if (timing_) { HIPCHK(hipEventRecord(ev0_, st_)); plan_event_pending_ = true; }   // the scan kernel's own time starts here
      }
      unsigned long long *trace = nullptr;
      if (scan_trace_) {
        trace_words_ = scan_prog_blocks(g_, (int)nd) * 4;
        HIPCHK(d_trace_.reserve(trace_words_));
        HIPCHK(hipMemsetAsync(d_trace_.p, 0, trace_words_ * sizeof(unsigned long long), st_));
        trace = d_trace_.p;
      }
      HIPCHK(launch_scan_prog(st_, g_, d_vec_, descs, (int)nd, d_prog_.p, d_out(), d_ncand_.p,
                              host_direct ? h_out() : nullptr, (uint32_t)nout, d_done_.p + 8, trace, shadow_ok_ && scan_shadow_));
      stats.plan_launches++;
    } else {
      HIPCHK(launch_scan_walk(st_, g_, d_vec_, d_kids(), n_, descs, (int)nd, d_out(), d_ncand_.p, maxd, mask_ptr, info_ptr,
                              host_direct ? h_out() : nullptr, (uint32_t)nout, d_done_.p + 8, shadow_ok_ && scan_shadow_));
    }
    if (timing_) HIPCHK(hipEventRecord(ev1_, st_));
    part_min_used_ = false;
    if (ufb_async_ && scan_masks_ && !check_counts_) {
      walk_async_ = true;
      walk_async_nd_ = nd;
      walk_async_nout_ = nout;
      return MPF_OK;
    }
    bool part_min_polled = false;
    if (use_pmin) {
      // only the cheapest candidate of every scan part is wanted: reduce on the device, minima straight to the host
      HIPCHK(h_pmin_.reserve(n_parts_dev_ + 1));
      // the mutation counts of the refresh in front ride along, and a polling host is told through a flag word behind the minima
      // (the scan's timing events lie in front of this launch: they are complete when the flag is up)
      part_min_polled = host_poll_ && n_parts_dev_ > 0;
      if (part_min_polled) __atomic_store_n(h_pmin_.p + n_parts_dev_, 0u, __ATOMIC_RELAXED);
      HIPCHK(launch_part_min(st_, d_out(), d_parts_.p, (int)n_parts_dev_, h_pmin_.p, d_cnt(), cnt_copy_pending_ ? h_cnt() : nullptr,
                             (uint32_t)nslots_, part_min_polled ? d_done_.p + 24 : nullptr));
      cnt_copy_pending_ = false;
      part_min_used_ = true;
    } else if (host_direct) {
      cnt_copy_pending_ = false;
    } else if (cnt_copy_pending_) {
      HIPCHK(hipMemcpyAsync(h_cnt(), d_cnt(), (out_off() + nout) * sizeof(uint32_t), hipMemcpyDeviceToHost, st_));
      cnt_copy_pending_ = false;
    } else if (nout) {
      HIPCHK(hipMemcpyAsync(h_out(), d_out(), nout * sizeof(uint32_t), hipMemcpyDeviceToHost, st_));
    }
    if (check_counts_) HIPCHK(hipMemcpyAsync(h_ncand_.p, d_ncand_.p, nd * sizeof(uint32_t), hipMemcpyDeviceToHost, st_));
    if (scan_masks_ && nout) {
      HIPCHK(ufb_->h_info.reserve(nout));
      HIPCHK(hipMemcpyAsync(ufb_->h_info.p, ufb_->info.p, nout * sizeof(uint2), hipMemcpyDeviceToHost, st_));
    }
    if (host_direct && host_poll_ && !timing_ && wait_host_flag(h_out() + nout)) {
      // the scan's last workgroup has written the costs and raised the flag behind them: no need to wait for the stream
    } else if (part_min_polled && wait_host_flag(h_pmin_.p + n_parts_dev_)) {
      // (the same behind the per-part minima)
    } else {
      HIPCHK(hipStreamSynchronize(st_));
    }
    if (check_counts_)
      for (size_t i = 0; i < nd; i++)
        if (h_ncand_.p[i] != h_walk_.p[i].pad0) { set_error("device/host candidate count mismatch"); return MPF_E_STATE; }
  }
  else if (pending_scores_) {
    if (cnt_copy_pending_) {
      HIPCHK(hipMemcpyAsync(h_cnt(), d_cnt(), nslots_ * sizeof(uint32_t), hipMemcpyDeviceToHost, st_));
      cnt_copy_pending_ = false;
    }
    HIPCHK(hipStreamSynchronize(st_));
  }
  return run_walks_finish(plans, out_host);
}

int Engine::run_walks_finish(std::vector<ScanPlan> &plans, const uint32_t **out_host)
{
  const size_t nd = n_walk_;
  walk_async_ = false;
  if (nd > 0) {
    float ms = 0;
    if (timing_ && hipEventElapsedTime(&ms, ev0_, ev1_) == hipSuccess) { stats.last_scan_kernel_ms = ms; stats.scan_kernel_ms_total += ms; }
    if (plan_event_pending_) {
      if (hipEventElapsedTime(&ms, ev4_, ev0_) == hipSuccess) stats.plan_kernel_ms_total += ms;
      plan_event_pending_ = false;
    }
    stats.scan_launches++;
  }
  finish_views();
  uint64_t tests = 0;
  for (ScanPlan &pl : plans) {
    const int q = back_[pl.rec];
    pl.base = (tip(pl.rec) ? 0u : sc_[pl.rec]) + (tip(q) ? 0u : sc_[q]);
    tests += (uint64_t)pl.n_total;
  }
  stats.insertion_tests += tests;
  stats.algorithmic_bytes += tests * 6u * (uint64_t)g_.S * (uint64_t)Wref_ * 4u;
  *out_host = h_out();
  n_walk_ = 0;
  walk_out_ = 0;
  return MPF_OK;
}

// plan + run the scans of `count` prune records in the configured mode
int Engine::scan_batch(std::vector<ScanPlan> &plans, const int *recs, int count, int mintrav, int maxtrav, const uint32_t **out)
{
  if (ufb_) { ufb_->st_valid = false; ufb_->st_dev = nullptr; }
  int mt = std::min(maxtrav, ntips_ - 3);
  // (above 8 levels the plain scans are host-planned programs; the tracker's masks exist in the device-walked kernel only, which
  //  parks its up-vectors in HBM scratch there: k_scan_walk_deep)
  const bool walk = scan_mode_ == 1 && (mt <= kWalkMaxDepth || scan_masks_) && !sankoff_;
  plans.resize((size_t)count);
  if (walk && !views_valid_ && count < n_ / 2 && count <= small_batch_max_) {
    // small batch (the inside of a climb): plan first -- planning needs the topology only -- so that the refresh launch
    // can clear the scan's outputs, then refresh just the vectors these scans read
    {
      ScopedMs timer(stats.host_plan_ms_total);
      for (int i = 0; i < count; i++) {
        int rc = plan_walk(recs[i], mintrav, maxtrav, plans[(size_t)i], count <= split_below_);
        if (rc) return rc;
      }
    }
    HIPCHK(reserve_results(walk_out_));
    zero_req_ptr_ = d_out();
    zero_req_words_ = clear_words(walk_out_);
    ride_[0].src = h_walk_.p;
    ride_[0].bytes = n_walk_ * sizeof(WalkDesc);
    const bool stage = ufb_async_ && scan_masks_ && ufb_;
    if (stage) {                                   // the tracker's staging block of this batch goes up with the same copy
      int rc = ufb_stage_small(plans, count);
      if (rc) return rc;
      ride_[1].src = ufb_->h_small.p;
      ride_[1].bytes = (size_t)ufb_->st_words * sizeof(uint32_t);
    }
    want_host_results_ = true;
    sb_roots_.clear();
    for (int i = 0; i < count; i++) collect_scan_roots(recs[i], mintrav, maxtrav, sb_roots_);
    int rc = schedule_views(&sb_roots_);
    zero_req_ptr_ = nullptr;
    zero_req_words_ = 0;
    ride_[0].src = nullptr;
    if (stage) {
      ufb_->st_dev = static_cast<const uint32_t *>(ride_[1].dev);
      ride_[1].src = nullptr;
      ride_[1].dev = nullptr;
    }
    if (!rc) rc = run_walks(plans, out);
    want_host_results_ = false;
    cnt_on_host_ = false;
    return rc;
  }
  if (!views_valid_) {
    if (!walk) { int rc = update_views(); if (rc) return rc; }      // host-planned programs need the scores first
    else { int rc = schedule_views(nullptr); if (rc) return rc; }
  }
  // a whole sweep of the same topology with the same options as the last one planned into these very plans: descriptors,
  // output layout and the device program are still in place (the topology alone determines them)
  const int key[6] = {mintrav, maxtrav, count, split_below_, split_cands_, scan_prog_ * 16 + g_.vw * 4 + g_.map * 2 + g_.big};
  const bool hit = walk && (plan_cache_ & 4) && sweep_cache_valid_ && !scan_masks_ && &plans == &sweep_plans_ && count >= n_ / 2 &&
                   sweep_cache_gen_ == walk_gen_ && std::equal(key, key + 6, sweep_cache_key_) && n_walk_ == 0 && !check_counts_ &&
                   sweep_cache_recs_.size() == (size_t)count && std::equal(recs, recs + count, sweep_cache_recs_.begin());
  if (hit) {
    n_walk_ = sweep_cache_nwalk_;
    walk_out_ = sweep_cache_out_;
    walk_dev_reuse_ = true;
    int rc = run_walks(plans, out);
    walk_dev_reuse_ = false;
    return rc;
  }
  {
    ScopedMs timer(stats.host_plan_ms_total);
    if (walk && count >= n_ / 2 && (visits_filled_epoch_ != topo_epoch_ || vis_dense_m_ != std::min(mt, 15))) {
      fill_visit_counts(std::min(mt, 15));
      visits_filled_epoch_ = topo_epoch_;
    }
    for (int i = 0; i < count; i++) {
      int64_t self = -1;
      if (!walk && scan_vals_) self = (int64_t)prog_out_++;      // the current tree's slot in front of this prune node's candidates
      int rc = walk ? plan_walk(recs[i], mintrav, maxtrav, plans[(size_t)i], count <= split_below_)
                    : plan_scan(recs[i], mintrav, maxtrav, plans[(size_t)i]);
      if (rc) return rc;
      if (!walk) plans[(size_t)i].self_idx = self;
      if (self >= 0 && sankoff_ && asym_) {
        // an asymmetric matrix: the current tree has another length, and other per-pattern lengths, at every prune node's visit
        // -- the reference evaluates it at that node's edge (evaluateParsimony(p), :2285).  One more program of one op writes the
        // row and the length to the visit's slot: min_x(vec[q][x] + m(vec[p])[x]), vec[q] = m(x1) + m(x2) of q's children.
        const int pr = recs[i], q = back_[pr];
        ScanHdr h;
        h.op_begin = (uint32_t)prog_ops_.size();
        h.s_slot = slot(pr);
        h.pad = 0;
        ScanOp o;
        if (!tip(q)) { o.own = slot(back_[nx(q)]); o.sib = slot(back_[nx(nx(q))]); o.meta = ((uint32_t)SCAN_JOIN << 16) | (1u << 8); }
        else { o.own = 0; o.sib = slot(q); o.meta = ((uint32_t)SCAN_EVAL << 16) | (1u << 8); }
        o.out = (uint32_t)self;
        prog_ops_.push_back(o);
        h.op_end = (uint32_t)prog_ops_.size();
        prog_hdr_.push_back(h);
      }
    }
  }
  if (walk && (plan_cache_ & 4) && !scan_masks_ && &plans == &sweep_plans_ && count >= n_ / 2) {
    sweep_cache_valid_ = true;
    sweep_cache_gen_ = walk_gen_;
    sweep_cache_nwalk_ = n_walk_;
    sweep_cache_out_ = walk_out_;
    std::copy(key, key + 6, sweep_cache_key_);
    sweep_cache_recs_.assign(recs, recs + count);
  }
  if (walk) return run_walks(plans, out);
  int rc = run_scans(plans, out_scratch_);
  *out = out_scratch_.data();
  return rc;
}

int Engine::spr_scan(int rec, int mintrav, int maxtrav, std::vector<int32_t> &q, std::vector<uint32_t> &mp, int &n_p)
{
  if (!have_tree_) { set_error("no tree set"); return MPF_E_STATE; }
  if (rec < 3 || rec >= 3 * (2 * n_ - 1) || back_[rec] < 0) { set_error("bad prune record"); return MPF_E_INVALID; }
  std::vector<ScanPlan> plans;
  const uint32_t *out = nullptr;
  int rc = scan_batch(plans, &rec, 1, mintrav, maxtrav, &out);
  if (rc) return rc;
  const ScanPlan &pl = plans[0];
  q.clear();
  mp.clear();
  if (pl.walked) {
    if (pl.n_parts_p > 0) enumerate_side(pl.rec, 1, pl.maxtrav, q);
    if ((int)q.size() != pl.n_p) { set_error("device/host enumeration mismatch"); return MPF_E_STATE; }
    if (pl.n_parts > pl.n_parts_p) enumerate_side(back_[pl.rec], pl.mintrav_q, pl.maxtrav, q);
    if ((int)q.size() != pl.n_total) { set_error("device/host enumeration mismatch"); return MPF_E_STATE; }
    for (size_t c = 0; c < q.size(); c++) mp.push_back(pl.base + pl.cost(c, out));
  } else {
    for (size_t c = 0; c < pl.cands.size(); c++) { q.push_back(pl.cands[c].q); mp.push_back(pl.base + pl.cost(c, out)); }
  }
  n_p = pl.n_p;
  return MPF_OK;
}

int Engine::sweep_scan(int mintrav, int maxtrav, uint64_t *n_tests, uint32_t *min_mp)
{
  if (!have_tree_) { set_error("no tree set"); return MPF_E_STATE; }
  ScopedMs sweep_timer(stats.host_sweep_ms_total);
  node_rectifier();
  {
    // nothing valid (a tree just handed over or re-weighted) and the planned scan applies: schedule, descriptors and program
    // are all made on the device
    const int mt = std::min(maxtrav, ntips_ - 3);
    if (dev_plan_ && mintrav == 1 && mt >= 1 && mt <= 6 && all_invalid_ && scan_mode_ == 1 && scan_prog_ > 0 && scan_prog_supported(g_, mt) &&
        !scan_masks_ && !check_counts_ && !scan_trace_ && host_poll_ && n_ >= 8 &&
        (dev_sched_usable() || (dsw_valid_ && sched_cache_valid_ && !kids_dirty_ && !sankoff_)))
      return sweep_scan_dev(mt, n_tests, min_mp);
  }
  std::vector<ScanPlan> &plans = sweep_plans_;
  const uint32_t *out = nullptr;
  want_part_min_ = true;
  part_min_used_ = false;
  int rc = scan_batch(plans, nodep_.data() + 1, 2 * n_ - 2, mintrav, maxtrav, &out);
  want_part_min_ = false;
  if (rc) return rc;
  uint32_t best = UINT_MAX;
  uint64_t tests = 0;
  size_t part_at = 0;                              // part_min_used_: the parts' minima, in plan order
  for (const ScanPlan &pl : plans) {
    const size_t nc = pl.walked ? (size_t)pl.n_total : pl.cands.size();
    if (pl.walked && part_min_used_) {
      uint32_t m = UINT_MAX;
      for (int i = 0; i < pl.n_parts; i++) m = std::min(m, h_pmin_.p[part_at++]);
      if (nc) best = std::min(best, pl.base + m);
    } else if (pl.walked) {
      uint32_t m = UINT_MAX;
      for (int i = 0; i < pl.n_parts; i++) {
        const uint32_t *o = out + pl.part_off[i];
        const int cnt = pl.part_cnt[i];
        if (cnt > 0) m = std::min(m, min_u32(o, cnt));
      }
      if (nc) best = std::min(best, pl.base + m);
    } else {
      for (size_t c = 0; c < nc; c++) best = std::min(best, pl.base + pl.cost(c, out));
    }
    tests += nc;
  }
  if (n_tests) *n_tests = tests;
  if (min_mp) *min_mp = best;
  return MPF_OK;
}

// mpf_sweep_scan on a tree that has no valid vector, with nothing of it planned on the host: topology array up, then k_sched
// (refresh schedule + scan descriptors), refresh, k_walk_plan, k_scan_prog, k_part_min; the host learns the number of scan parts
// from a flag word (it needs it for the grids), then waits for the per-part minima and the refresh's mutation counts.
int Engine::sweep_scan_dev(int mt, uint64_t *n_tests, uint32_t *min_mp)
{
  const int key[4] = {mt, split_cands_, g_.vw * 4 + g_.map * 2 + g_.big, 0};
  // the same topology as the last device-planned sweep, nothing else planned since: descriptors, program and part table are
  // still in place (mpf_set_tree found the links unchanged: only the vectors were invalidated)
  const bool warm = (plan_cache_ & 4) && dsw_valid_ && dsw_walk_gen_ == walk_gen_ && prog_gen_ == walk_gen_ && parts_gen_ == walk_gen_ &&
                    sched_cache_valid_ && dsw_sched_gen_ == sched_gen_ && !kids_dirty_ && kids_list_.empty() && std::equal(key, key + 4, dsw_key_);
  int rc;
  if (warm) rc = schedule_views(nullptr);          // (the cached schedule: one launch, no host work)
  else {
    if (!dev_sched_usable()) { dsw_valid_ = false; return sweep_scan(1, mt, n_tests, min_mp); }   // (cannot happen: the caller checked)
    dsw_valid_ = false;
    rc = schedule_views_dev(mt);
  }
  if (rc) return rc;
  ScopedMs timer(stats.host_scan_ms_total);
  if (!warm) {
    if (!wait_host_flag(h_dsw_.p + 3)) HIPCHK(hipStreamSynchronize(st_));
    dsw_parts_ = h_dsw_.p[0];
    dsw_out_ = h_dsw_.p[1];
    dsw_walk_gen_ = walk_gen_;
    std::copy(key, key + 4, dsw_key_);
    dsw_valid_ = true;
  }
  const size_t nd = dsw_parts_, nout = dsw_out_;
  uint32_t best = UINT_MAX;
  if (nd > 0) {
    HIPCHK(d_ncand_.reserve(nd));
    HIPCHK(reserve_results(nout));
    HIPCHK(d_prog_.reserve(scan_prog_bytes((int)nd)));      // (cold: all of these are in place already, schedule_views_dev saw to it)
    HIPCHK(h_pmin_.reserve(nd + 1));
    if (warm) HIPCHK(hipMemsetAsync(d_out(), 0, clear_words(nout) * sizeof(uint32_t), st_));
    if (timing_) HIPCHK(hipEventRecord(ev0_, st_));
    if (!warm) {
      // (the program was planned and the outputs were cleared by the refresh launch's extra workgroups)
      if (!plan_ride_) HIPCHK(launch_walk_plan(st_, d_kids(), n_, d_walk_.p, (int)nd, d_prog_.p, d_out(), (uint32_t)clear_words(nout)));
      prog_gen_ = walk_gen_;
      parts_gen_ = walk_gen_;
      n_parts_dev_ = nd;
    }
    HIPCHK(launch_scan_prog(st_, g_, d_vec_, d_walk_.p, (int)nd, d_prog_.p, d_out(), d_ncand_.p, nullptr, (uint32_t)nout, d_done_.p + 8, nullptr,
                            shadow_ok_ && scan_shadow_));
    stats.plan_launches++;
    if (timing_) HIPCHK(hipEventRecord(ev1_, st_));
    __atomic_store_n(h_pmin_.p + nd, 0u, __ATOMIC_RELAXED);
    HIPCHK(launch_part_min(st_, d_out(), d_parts_.p, (int)nd, h_pmin_.p, d_cnt(), cnt_copy_pending_ ? h_cnt() : nullptr, (uint32_t)nslots_,
                           d_done_.p + 24));
    cnt_copy_pending_ = false;
    if (!wait_host_flag(h_pmin_.p + nd)) HIPCHK(hipStreamSynchronize(st_));
    float ms = 0;
    if (timing_ && hipEventElapsedTime(&ms, ev0_, ev1_) == hipSuccess) { stats.last_scan_kernel_ms = ms; stats.scan_kernel_ms_total += ms; }
    if (plan_event_pending_) {
      if (hipEventElapsedTime(&ms, ev4_, ev0_) == hipSuccess) stats.plan_kernel_ms_total += ms;
      plan_event_pending_ = false;
    }
    stats.scan_launches++;
  } else {
    if (cnt_copy_pending_) {
      HIPCHK(hipMemcpyAsync(h_cnt(), d_cnt(), nslots_ * sizeof(uint32_t), hipMemcpyDeviceToHost, st_));
      cnt_copy_pending_ = false;
    }
    HIPCHK(hipStreamSynchronize(st_));
  }
  finish_views();
  // base length of a prune node's candidates = the two subtrees either side of the cut (reference :2158-2160)
  const uint32_t *node = h_dsw_.p + 4;
  for (size_t i = 0; i < nd; i++) {
    const uint32_t m = h_pmin_.p[i];
    if (m == UINT_MAX) continue;                   // (a part without candidates)
    const int p = nodep_[node[i] + 1], q = back_[p];
    best = std::min(best, (tip(p) ? 0u : sc_[p]) + (tip(q) ? 0u : sc_[q]) + m);
  }
  stats.insertion_tests += nout;
  stats.algorithmic_bytes += (uint64_t)nout * 6u * (uint64_t)g_.S * (uint64_t)Wref_ * 4u;
  if (n_tests) *n_tests = nout;
  if (min_mp) *min_mp = best;
  return MPF_OK;
}

int Engine::sweep_costs(int mintrav, int maxtrav, uint64_t cap, uint32_t *mp, uint64_t *offsets, uint64_t *n_tests)
{
  if (!have_tree_) { set_error("no tree set"); return MPF_E_STATE; }
  // a caller sizes its buffer with a first call (cap = 0) and fetches with a second one: the first call's sweep is kept, the
  // second one copies (same tree, same packing, same radius) instead of scanning the whole tree again
  const uint64_t key[4] = {(uint64_t)topo_epoch_, (uint64_t)mintrav, (uint64_t)maxtrav, pack_gen_};
  if (sc_keep_valid_ && std::equal(key, key + 4, sc_keep_key_) && cap >= sc_keep_mp_.size()) {
    *n_tests = sc_keep_mp_.size();
    std::copy(sc_keep_mp_.begin(), sc_keep_mp_.end(), mp);
    if (offsets) std::copy(sc_keep_off_.begin(), sc_keep_off_.end(), offsets);
    sc_keep_valid_ = false;
    sc_keep_mp_.clear();
    sc_keep_mp_.shrink_to_fit();
    return MPF_OK;
  }
  sc_keep_valid_ = false;
  node_rectifier();
  std::vector<ScanPlan> &plans = sweep_plans_;
  const uint32_t *out = nullptr;
  int rc = scan_batch(plans, nodep_.data() + 1, 2 * n_ - 2, mintrav, maxtrav, &out);
  if (rc) return rc;
  uint64_t tests = 0;
  for (const ScanPlan &pl : plans) tests += pl.walked ? (uint64_t)pl.n_total : (uint64_t)pl.cands.size();
  *n_tests = tests;
  const bool keep = cap < tests;                   // a sizing call: fill our own copy
  if (keep) { sc_keep_mp_.resize(tests); sc_keep_off_.assign(plans.size() + 1, 0); }
  uint32_t *dst = keep ? sc_keep_mp_.data() : mp;
  uint64_t *doff = keep ? sc_keep_off_.data() : offsets;
  uint64_t at = 0;
  size_t i = 0;
  for (const ScanPlan &pl : plans) {
    const size_t nc = pl.walked ? (size_t)pl.n_total : pl.cands.size();
    if (doff) doff[i] = at;
    for (size_t c = 0; c < nc; c++) dst[at + c] = pl.base + pl.cost(c, out);
    at += nc;
    i++;
  }
  if (doff) doff[i] = at;
  if (keep) {
    if (offsets) std::copy(sc_keep_off_.begin(), sc_keep_off_.end(), offsets);
    std::copy(key, key + 4, sc_keep_key_);
    sc_keep_valid_ = true;
  }
  return MPF_OK;
}

int Engine::node_order(int32_t *recs)
{
  if (!have_tree_) { set_error("no tree set"); return MPF_E_STATE; }
  node_rectifier();
  for (int i = 1; i <= 2 * n_ - 2; i++) recs[i - 1] = nodep_[(size_t)i];
  return MPF_OK;
}

// pllComputePatternParsimony (reference sprparsimony.cpp:3363-3392) for the current tree: the joins of the
// traversal rooted on the branch start--back[start] are counted per site on the device
int Engine::pattern_scores(uint16_t *ptn, int32_t *total)
{
  if (!have_tree_) { set_error("no tree set"); return MPF_E_STATE; }
  if (!views_valid_) { int rc = update_views(); if (rc) return rc; }
  if (sankoff_) {
    // pllComputeSankoffPatternParsimony (reference :3341-3355): the per-pattern cost of the root branch
    DevBuf<uint16_t> d_p;
    std::vector<uint16_t> hp((size_t)g_.Wp);
    HIPCHK(d_p.reserve((size_t)g_.Wp));
    HIPCHK(launch_sankoff_pattern(st_, g_, d_vec_, slot(back_[start_]), slot(start_), d_p.p));      // (left = far end, right = start: as tree_length)
    HIPCHK(hipMemcpyAsync(hp.data(), d_p.p, hp.size() * sizeof(uint16_t), hipMemcpyDeviceToHost, st_));
    HIPCHK(hipStreamSynchronize(st_));
    long sum = 0;
    for (int k = 0; k < P_; k++) ptn[k] = 0;
    for (int j = 0; j < ninf_; j++) {
      const int k = inf_index_[(size_t)j];
      ptn[k] = hp[(size_t)j];
      sum += (long)ptn[k] * wgt_[k];
    }
    if (total) *total = (int32_t)sum;
    return MPF_OK;
  }
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
  DevBuf<uint32_t> planes;
  DevBuf<int32_t> d_first;
  DevBuf<uint16_t> d_ptn;
  HIPCHK(d_evops_.reserve(ops.size()));
  HIPCHK(planes.reserve(site_planes_words(g_, (int)ops.size())));
  HIPCHK(d_first.reserve((size_t)P_));
  HIPCHK(d_ptn.reserve((size_t)P_));
  HIPCHK(hipMemcpyAsync(d_evops_.p, ops.data(), ops.size() * sizeof(EvOp), hipMemcpyHostToDevice, st_));
  HIPCHK(hipMemcpyAsync(d_first.p, first_site_.data(), (size_t)P_ * sizeof(int32_t), hipMemcpyHostToDevice, st_));
  HIPCHK(launch_site_counts(st_, g_, d_vec_, d_evops_.p, (int)ops.size(), planes.p, d_first.p, P_, d_ptn.p));
  HIPCHK(hipMemcpyAsync(ptn, d_ptn.p, (size_t)P_ * sizeof(uint16_t), hipMemcpyDeviceToHost, st_));
  HIPCHK(hipStreamSynchronize(st_));
  if (total) {
    long sum = 0;
    for (int k = 0; k < P_; k++) sum += (long)ptn[k] * wgt_[k];
    *total = (int32_t)sum;
  }
  return MPF_OK;
}

// pllComputeSiteParsimony (reference sprparsimony.cpp:3403-3450): per expanded site, from the per-pattern lengths
int Engine::site_scores(int32_t *site_pars, int n_sites, int32_t *total)
{
  if (sankoff_) { set_error("mpf_site_scores: Fitch mode only (the weighted engine keeps patterns, not expanded sites)"); return MPF_E_UNSUPPORTED; }
  std::vector<uint16_t> ptn((size_t)P_);
  int rc = pattern_scores(ptn.data(), nullptr);
  if (rc) return rc;
  long sum = 0;
  int site = 0;
  for (int k = 0; k < P_ && site < n_sites; k++) {
    if (first_site_[(size_t)k] < 0) continue;
    for (int w = 0; w < wgt_[(size_t)k] && site < n_sites; w++) { site_pars[site++] = ptn[(size_t)k]; sum += ptn[(size_t)k]; }
  }
  for (; site < n_sites; site++) site_pars[site] = 0;
  if (total) *total = (int32_t)sum;
  return MPF_OK;
}

int Engine::set_option(const std::string &key, int64_t v)
{
  if (key == "scan_batch") { if (v < 1) return MPF_E_INVALID; scan_batch_ = (int)v; return MPF_OK; }
  if (key == "words_per_lane") {
    if (sankoff_ && v != 1) { set_error("words_per_lane: weighted mode uses one pattern per lane"); return MPF_E_INVALID; }
    if (!(v == 1 || v == 2 || v == 4) || (g_.S == 20 && v == 4)) { set_error("words_per_lane: 1|2|4 (protein: 1|2)"); return MPF_E_INVALID; }
    // (20 or 32 state rows fill a lane's registers with ONE word: two words per lane put 40-register tiles in every kernel and all of
    //  them spilled to scratch -- k_newview_wg<20, 2> 352-408 bytes, k_scan<20, 2, 12> 500, k_newview_chain<20, 2> 1 KiB per lane,
    //  round 5's disassembly --; the knob rests for these alphabets since round 6 and the instantiations are gone)
    if (g_.S >= 20) return MPF_OK;
    g_.vw = (int)v;
    return MPF_OK;
  }
  if (key == "dev_sched") { dev_sched_ = v != 0; sched_cache_valid_ = false; dsw_valid_ = false; return MPF_OK; }
  if (key == "dev_plan") { dev_plan_ = v != 0; plan_ride_ = !(v & 2); dsw_valid_ = false; return MPF_OK; }   // (bit 1: the walk plan as a launch of its own)               // scan descriptors of mpf_sweep_scan laid out on the device   // refresh schedule of a new topology made on the device (k_sched)
  if (key == "scan_shadow") { scan_shadow_ = v != 0; return MPF_OK; }        // planned scan reads the word-major copy when it is current (A/B switch)
  if (key == "reduce") { g_.reduce = v ? 1 : 0; return MPF_OK; }
  if (key == "xcd_map") { g_.map = v ? 1 : 0; return MPF_OK; }
  if (key == "scan_mode") { scan_mode_ = v ? 1 : 0; return MPF_OK; }
  if (key == "views_pipe") { g_.nv_pipe = v != 0; sched_cache_valid_ = false; return MPF_OK; }
  if (key == "views_tile") {
    if (v != 32 && v != 16 && v != 8 && v != 4 && v != 0) { set_error("views_tile: 32, 16, 8 or 4 words (0: chosen from the row length)"); return MPF_E_INVALID; }
    g_.nv_tile = (int)v;
    sched_cache_valid_ = false;
    return MPF_OK;
  }
  if (key == "views_mode") { views_mode_ = v < 0 ? 0 : v > 2 ? 2 : (int)v; sched_cache_valid_ = false; return MPF_OK; }
  if (key == "host_poll") { host_poll_ = v ? 1 : 0; return MPF_OK; }
  if (key == "ufb_fast") { ufb_fast_ = v ? 1 : 0; return MPF_OK; }
  if (key == "ufb_quiet") { ufb_quiet_ = v ? 1 : 0; return MPF_OK; }
  if (key == "deep_scratch_kwords") {             // (tests: a small scratch cuts a deep scan into many launches)
    if (v < 1 || v > (1 << 22)) { set_error("deep_scratch_kwords: 1 .. 4194304"); return MPF_E_INVALID; }
    deep_scratch_words_ = (size_t)v << 10; d_deep_.release(); g_.deep_scratch = nullptr; g_.deep_scratch_words = 0;
    return MPF_OK;
  }
  if (key == "grow_device") { grow_device_ = v ? 1 : 0; return MPF_OK; }
  if (key == "grow_tile") { if (v != 0 && v != 1 && v != 2 && v != 4 && v != 8 && v != -1) { set_error("grow_tile: 0 (word-major copy where there is one, else fitted), -1 (fitted quad tiles), 1, 2, 4 or 8"); return MPF_E_INVALID; } grow_vw_ = (int)v; return MPF_OK; }
  if (key == "grow_fault") { grow_fault_ = v; return MPF_OK; }
  if (key == "max_visits") { max_visits_ = std::max<int64_t>(0, v); return MPF_OK; }
  if (key == "small_batch_max") { small_batch_max_ = (int)std::max<int64_t>(1, std::min<int64_t>(v, 1 << 30)); return MPF_OK; }
  if (key == "ufb_moot") { ufb_moot_ = v ? 1 : 0; return MPF_OK; }
  if (key == "ufb_memo") { ufb_memo_ = v ? 1 : 0; return MPF_OK; }
  if (key == "ufb_cut_batch") { ufb_cut_batch_ = (int)std::max<int64_t>(1, std::min<int64_t>(v, 1 << 20)); return MPF_OK; }
  if (key == "ufb_pipe") { ufb_pipe_ = v ? 1 : 0; return MPF_OK; }
  if (key == "ufb_thread") { ufb_thread_ = v ? 1 : 0; return MPF_OK; }
  if (key == "ufb_event_cap") { if (v < 16 || v > (1ll << 28)) { set_error("ufb_event_cap: 16 .. 2^28"); return MPF_E_INVALID; } ufb_event_cap_ = v; return MPF_OK; }
  if (key == "plan_cache") { plan_cache_ = v ? 7 : 0; sched_cache_valid_ = false; sweep_cache_valid_ = false; return MPF_OK; }
  if (key == "split_below") { split_below_ = (int)v; return MPF_OK; }
  if (key == "split_cands") { split_cands_ = v < 0 ? 0 : (int)v; return MPF_OK; }
  if (key == "scan_trace") { scan_trace_ = v ? 1 : 0; return MPF_OK; }
  if (key == "scan_prog") { scan_prog_ = v < 0 ? 0 : v > 2 ? 2 : (int)v; return MPF_OK; }
  if (key == "prog_min_descs") { prog_min_descs_ = v < 0 ? 0 : (int)v; return MPF_OK; }
  if (key == "check_counts") { check_counts_ = v ? 1 : 0; return MPF_OK; }
  if (key == "chain_max_ops") { chain_max_ops_ = v < 0 ? 0 : (long)v; return MPF_OK; }
  if (key == "timing") { timing_ = v < 0 ? 0 : v > 2 ? 2 : (int)v; return MPF_OK; }
  if (key == "force_big") {                     // test hook: use the >= 2 GiB addressing path on any size
    force_big_ = v ? 1 : 0;
    if (!sankoff_) g_.big = (force_big_ || vec_words_ * sizeof(uint32_t) >= ((size_t)1 << 31)) ? 1 : 0;
    return MPF_OK;
  }
  if (key == "climb_device") { climb_device_ = v < 0 ? 0 : v > 2 ? 2 : (int)v; return MPF_OK; }
  if (key == "climb_tile") {
    if (v != 1 && v != 2 && v != 4 && v != 8) { set_error("climb_tile: 1|2|4|8 words per lane group (tiles of 16, 32, 64, 128 words)"); return MPF_E_INVALID; }
    climb_vw_ = (int)v;
    climb_vw_set_ = true;
    return MPF_OK;
  }
  if (key == "climb_batch_min") { climb_batch_min_ = v < 1 ? 1 : v > 16 ? 16 : (int)v; climb_batch_min_set_ = true; return MPF_OK; }
  if (key == "climb_batch_max") { climb_batch_max_ = v < 1 ? 1 : v > 16 ? 16 : (int)v; return MPF_OK; }
  if (key == "climb_near_q") { climb_near_q_ = v < 1 ? 1 : v > 32 ? 32 : (int)v; climb_near_q_set_ = true; return MPF_OK; }
  if (key == "climb_batch_max_sparse") { climb_batch_max_sparse_ = v < 1 ? 1 : v > 16 ? 16 : (int)v; return MPF_OK; }
  if (key == "climb_idle") { climb_idle_ = v < 1 ? 1 : (int)v; return MPF_OK; }
  if (key == "climb_word_major") { climb_word_major_ = v != 0; return MPF_OK; }
  if (key == "many_word_major") { many_word_major_ = v != 0; return MPF_OK; }
  if (key == "many_moves_cap") { many_moves_cap_ = v < 0 ? 0 : (int)std::min<long long>(v, 1 << 20); return MPF_OK; }
  if (key == "many_sweeps_inside") { many_sweeps_inside_ = v != 0; return MPF_OK; }
  if (key == "climb_groups") { climb_groups_ = v < 0 ? 0 : v > 4096 ? 4096 : (int)v; return MPF_OK; }
  if (key == "climb_fault") { climb_fault_ = v; return MPF_OK; }          // (tests of the recovery paths: climb.hpp)
  if (key == "views_waves") { nv_waves_ = (int)v; return MPF_OK; }         // waves per refresh workgroup: 0 = by level width, -1 = always sixteen, 2 .. 16
  if (key == "refine_chunk") { refine_chunk_ = v < 1 ? 1 : (int)std::min<int64_t>(v, 1 << 30); return MPF_OK; }
  if (key == "climb_trace") { climb_trace_ = v ? 1 : 0; cd_.h_trace.clear(); cd_.trace_records = 0; return MPF_OK; }
  if (key == "sankoff_short") {                 // 0 = always 32-bit costs (the reference's -short_off); takes effect at the next re-pack
    snk16_opt_ = v ? 1 : 0;
    if (sankoff_) return pack();
    return MPF_OK;
  }
  set_error("unknown option " + key);
  return MPF_E_INVALID;
}

// diagnostic: the per-workgroup timeline of the last planned-program scan launch (option "scan_trace")
int Engine::scan_trace(uint64_t *out, uint64_t cap, uint64_t *n)
{
  if (climb_trace_) {
    // option "climb_trace": eight 32-bit words per prune node the device-resident climb visited (prune index, prune vector id,
    // insertion tests, of them on the p side, cheapest, bestParsimony afterwards, chosen candidate, accepted), two per 64-bit word
    *n = cd_.h_trace.size() / 2;
    if (cap >= *n && *n) std::memcpy(out, cd_.h_trace.data(), cd_.h_trace.size() * sizeof(uint32_t));
    return MPF_OK;
  }
  *n = trace_words_;
  if (!trace_words_ || cap < trace_words_) return MPF_OK;
  activate();
  HIPCHK(hipMemcpy(out, d_trace_.p, trace_words_ * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return MPF_OK;
}

int Engine::get_option(const std::string &key, int64_t *v) const
{
  if (key == "scan_prog") *v = scan_prog_;
  else if (key == "prog_min_descs") *v = prog_min_descs_;
  else if (key == "scan_batch") *v = scan_batch_;
  else if (key == "words_per_lane") *v = g_.vw;
  else if (key == "kernel_states") *v = g_.S;            // state rows the kernels carry: 4 (DNA, binary), 20 (protein, multistate <= 20 symbols), 32
  else if (key == "reduce") *v = g_.reduce;
  else if (key == "xcd_map") *v = g_.map;
  else if (key == "scan_mode") *v = scan_mode_;
  else if (key == "views_mode") *v = views_mode_;
  else if (key == "views_pipe") *v = g_.nv_pipe;
  else if (key == "views_tile") *v = g_.nv_tile;
  else if (key == "plan_cache") *v = plan_cache_;
  else if (key == "sched_levels" || key == "sched_ticks" || key == "sched_desc_ticks") {
    // diagnostics of the last device-made schedule: dependency levels, duration of k_sched's two workgroups (10 ns ticks)
    int32_t w[16] = {0};
    if (sched_on_dev_ && d_vstage_.p) { (void)hipStreamSynchronize(st_); (void)hipMemcpy(w, d_vstage_.p + sc_nlev_off_, 64, hipMemcpyDeviceToHost); }
    *v = w[key == "sched_levels" ? 0 : key == "sched_ticks" ? 1 : 2];
  }
  else if (key == "split_below") *v = split_below_;
  else if (key == "split_cands") *v = split_cands_;
  else if (key == "chain_max_ops") *v = chain_max_ops_;
  else if (key == "timing") *v = timing_;
  else if (key == "ufb_fast") *v = ufb_fast_;
  else if (key == "ufb_quiet") *v = ufb_quiet_;
  else if (key == "grow_device") *v = grow_device_;
  else if (key == "grow_tile") *v = grow_vw_;
  else if (key == "grow_launches") *v = (int64_t)grow_launches_;
  else if (key == "grow_steps") *v = (int64_t)grow_steps_;
  else if (key == "grow_us") *v = (int64_t)(grow_ms_total_ * 1000.0);
  else if (key == "grow_last_err") *v = grow_last_err_;
  else if (key.rfind("grow_ticks_", 0) == 0 && key.size() == 12 && key[11] >= '0' && key[11] <= '6') *v = (int64_t)grow_phase_ticks_[key[11] - '0'];
  else if (key == "max_visits") *v = max_visits_;
  else if (key == "ufb_moot") *v = ufb_moot_;
  else if (key == "ufb_memo") *v = ufb_memo_;
  else if (key == "ufb_memo_batches") *v = ufb_ ? (int64_t)ufb_->memo_batches : 0;
  else if (key == "ufb_cut_batch") *v = ufb_cut_batch_;
  else if (key == "ufb_quiet_climbs") *v = ufb_stat_quiet_;
  else if (key == "ufb_pipe") *v = ufb_pipe_;
  else if (key == "ufb_thread") *v = ufb_thread_;
  else if (key == "ufb_event_cap") *v = ufb_event_cap_;
  else if (key == "ufb_batches") *v = ufb_stat_batches_;
  else if (key == "ufb_early_batches") *v = ufb_stat_early_;
  else if (key == "force_big") *v = force_big_;
  else if (key == "sankoff_short") *v = snk16_opt_;
  else if (key == "check_counts") *v = check_counts_;
  else if (key == "climb_device") *v = climb_device_;
  else if (key == "climb_tile") *v = climb_vw_;
  else if (key == "climb_batch_min") *v = climb_batch_min_;
  else if (key == "climb_batch_max") *v = climb_batch_max_;
  else if (key == "climb_batch_max_sparse") *v = climb_batch_max_sparse_;
  else if (key == "climb_near_q") *v = climb_near_q_;
  else if (key == "climb_idle") *v = climb_idle_;
  else if (key == "refine_chunk") *v = refine_chunk_;
  else if (key == "views_waves") *v = nv_waves_;
  else if (key == "climb_trace") *v = climb_trace_;
  else if (key == "climb_groups") *v = climb_groups_;
  else if (key == "many_moves_cap") *v = many_moves_cap_;
  else if (key == "climb_word_major") *v = climb_word_major_ ? 1 : 0;
  else if (key == "many_word_major") *v = many_word_major_ ? 1 : 0;
  else if (key == "climb_tile_many") *v = const_cast<Engine *>(this)->climb_fit_vw(true);      // the width mpf_optimize_spr_many runs this engine's climbs on
  else if (key == "many_sweeps_inside") *v = many_sweeps_inside_ ? 1 : 0;
  else if (key.rfind("climb_ctr", 0) == 0 && key.size() == 10 && key[9] >= '0' && key[9] <= '3') *v = (int64_t)climb_ctr_[key[9] - '0'];   // refresh ops, closure rounds, invalidation rounds, chains
  else if (key.rfind("climb_phase_us", 0) == 0 && key.size() == 15 && ((key[14] >= '0' && key[14] <= '9') || (key[14] >= 'a' && key[14] <= 'f')))
    *v = (int64_t)(climb_phase_ticks_[key[14] <= '9' ? key[14] - '0' : key[14] - 'a' + 10] / 100ull);
  else { set_error("unknown option " + key); return MPF_E_INVALID; }
  return MPF_OK;
}

}  // namespace mpf
