// engine.hpp -- host side of the MI355X Fitch engine (one instance = one alignment on one GPU).
//
// The engine keeps, for the CURRENT tree, every directional vector in HBM: for each node
// record r the Fitch state sets of the subtree seen when looking from back[r] towards r
// (tips: the packed tip itself).  With those resident, every quantity the reference obtains
// by lazy re-orientation (xPars flags, sprparsimony.cpp:420-467) is a pure function of
// already-computed vectors, so whole neighbourhoods of SPR candidates are scored by one
// launch (kernels.hip, k_scan) and the host only replays the reference's accept/tie logic.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <chrono>
#include <cstdint>
#include <map>
#include <unordered_map>
#include <memory>
#include <climits>
#include <set>
#include <string>
#include <utility>
#include <vector>

#include "../../include/mpfitch.h"
#include "../host/rng.hpp"
#include "../host/ufb_books.hpp"
#include "climb.hpp"
#include "kernels.hpp"
#include "ufboot.hpp"

namespace mpf {

void set_error(const std::string &msg);
const std::string &last_error();

template <typename T>
struct DevBuf {
  T *p = nullptr;
  size_t cap = 0;
  ~DevBuf() { release(); }
  void release() { if (p) { (void)hipFree(p); p = nullptr; cap = 0; } }
  void swap(DevBuf &o) { std::swap(p, o.p); std::swap(cap, o.cap); }
  hipError_t reserve(size_t n)
  {
    if (n <= cap) return hipSuccess;
    release();
    size_t want = n + n / 2 + 64;
    hipError_t e = hipMalloc((void **)&p, want * sizeof(T));
    if (e == hipSuccess) cap = want;
    return e;
  }
};

template <typename T>
struct PinBuf {
  T *p = nullptr;
  size_t cap = 0;
  ~PinBuf() { release(); }
  void release() { if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; } }
  void swap(PinBuf &o) { std::swap(p, o.p); std::swap(cap, o.cap); }
  hipError_t reserve(size_t n)
  {
    if (n <= cap) return hipSuccess;
    release();
    size_t want = n + n / 2 + 64;
    hipError_t e = hipHostMalloc((void **)&p, want * sizeof(T), hipHostMallocDefault);
    if (e == hipSuccess) cap = want;
    return e;
  }
};

struct Candidate {
  int32_t q;       // record q of testInsertParsimony(p, q)
  uint32_t out;    // index into the scan output buffer
};

// the insertion tests of one prune record, in the reference's order
struct ScanPlan {
  int32_t rec = -1;
  uint32_t base = 0;        // length(rest) + length(pruned subtree)
  int n_p = 0;              // candidates of the p side come first
  int n_total = 0;
  std::vector<Candidate> cands;     // host-planned mode only
  // device-walked mode: up to 8 parts (P phase first), each a descriptor with its own output block
  bool walked = false;
  int n_parts = 0, n_parts_p = 0;
  int part_desc[8];
  uint32_t part_off[8];
  int part_cnt[8];
  int mintrav_q = 2, maxtrav = 0;
  // masked scans (online UFBoot): the output index reserved in front of this prune node's candidates for the CURRENT tree
  // -- rearrangeParsimony books it before it starts inserting (sprparsimony.cpp:2285-2289); -1 = none
  int64_t self_idx = -1;
  inline uint32_t cost(size_t c, const uint32_t *out) const
  {
    if (!walked) return out[cands[c].out];
    for (int i = 0; i < n_parts; i++) {
      if (c < (size_t)part_cnt[i]) return out[part_off[i] + c];
      c -= (size_t)part_cnt[i];
    }
    return 0;
  }
};

struct Move { int32_t remove_rec, insert_rec; uint32_t score; };

uint64_t lcg64_skip(uint64_t state, uint64_t k);      // the tie stream k draws on (host/ufboot.cpp)

// online UFBoot-MP: the arrays IQTree keeps per bootstrap sample (iqtree.cpp:213-262) plus the device buffers of
// the masked scan, the REPS product and the event extraction (ufboot.hip)
struct UfbState : books::Deferred {      // (boot_trees, store, refs, topo_index: the deferred state, host/ufb_books.hpp)
  int B = 0, Bp = 0, planes = 1;                 // B = all samples of the run (host arrays); Bp = padded LOCAL columns
  // sample sharding (multi-GPU online phase): this engine multiplies only its own samples, ids[c] = global sample of
  // local column c; every rank replays the merged events of all ranks, so the search chain stays identical everywhere
  int Bl = 0;
  std::vector<int32_t> ids;
  mpf_ufb_exchange_fn exchange = nullptr;
  void *exchange_arg = nullptr;
  double eps = 0.5;                              // params->ufboot_epsilon (tools.cpp:725)
  double logl_cutoff = 0.0;                      // IQTree::logl_cutoff, 0 = none (iqtree.cpp:68, :3343)
  // host bookkeeping (scores are parsimony lengths, i.e. -boot_logl; UINT32_MAX = "-LONG_MAX")
  std::vector<uint32_t> boot_score;
  std::vector<int32_t> boot_counts;
  // topologies whose complete move-less sweep produced no candidate event, with the widest cut-off (largest admissible length)
  // that held under: nothing to multiply when the search comes back to one (host/ufboot.cpp, `memo`)
  // The entry speaks of the insertion tests of ONE neighbourhood shape: the key carries the radii the event-free sweep ran under
  // (a later climb at a larger radius on the same topology has tests the memo never saw -- ADVICE r5).
  std::unordered_map<std::string, uint32_t> quiet_topo;
  std::string quiet_key(int mintrav, int maxtrav, int n_taxa) const
  {
    std::string k = self_key;
    const int hi = maxtrav < n_taxa - 3 ? maxtrav : n_taxa - 3;        // (no neighbourhood reaches further than the tree does)
    k.push_back((char)(mintrav & 0xFF)); k.push_back((char)(hi & 0xFF)); k.push_back((char)((hi >> 8) & 0xFF));
    return k;
  }
  uint64_t memo_batches = 0;
  // boot_tree_orig_logl (iqtree.h:766, -cutoff_from_btrees): the logl under which each sample's tree was booked; cur_logl_now = that
  // of the tree the replay has in hand (on ratchet climbs the value saveCurrentTree replaced it by)
  std::vector<int32_t> boot_orig;
  int32_t cur_logl_now = 0;
  bool cut_btrees = false;                       // params->cutoff_from_btrees: mpf_ufboot_next_cutoff = min(boot_orig)
  std::vector<uint32_t> treels;                  // treels_logl as lengths
  using Pending = books::Pending;                // accepted during the current prune node, not yet materialised
  std::vector<Pending> pending;
  // deferred part of the default update rule (host/ufboot.cpp, ufb_drain_log): which tree a sample points to, the topology map
  // and the stored topologies never feed back into a draw or into the search -- the replay only logs the acceptances (and the end
  // of every prune node's scan) and the log is worked off while the device runs the next batch
  using LogEntry = books::LogEntry;              // b = 0xFFFFFFFF: end of the scan of prune node `plan`
  std::vector<LogEntry> log;
  std::vector<int32_t> log_back;                 // the topology the log's candidates refer to
  int32_t log_epoch = 0;
  std::vector<int32_t> log_bk;
  std::vector<double> inv;                       // inv[k] = 1.0 / (double)k, grown on demand (the very quotient the rule compares a draw with)
  bool ids_identity = true;                      // local column c is sample c (an unsharded tracker): the bookings of eight samples at a time
  std::string log_key;
  double t_defer = 0;
  const std::vector<ScanPlan> *log_plans = nullptr;
  uint64_t draws = 0, events = 0, gemm_rows = 0, batches = 0, stored = 0;
  double gemm_ms = 0.0;
  double t_lookup = 0;                           // ... of t_replay: canonical forms for the topology map
  uint64_t lookups = 0;
  double t_scan = 0, t_prep = 0, t_dev = 0, t_sort = 0, t_replay = 0, t_rt = 0;   // host wall-clock split (ms), MPF_UFB_PROFILE=1 prints it
  // device
  DevBuf<uint8_t> wt;                            // [planes][Wp/2][Bp/16][4][16][16] signed bytes
  size_t plane_bytes = 0;
  DevBuf<uint32_t> masks;                        // [rows padded to kUfbRowTile][Wp]
  DevBuf<uint2> info;                            // per scan output index: (mask row, part) | (.., ~0) for a home slot
  DevBuf<int32_t> C;                             // [rows padded][Bp]
  DevBuf<uint32_t> jmasks;                       // join masks of the current tree (R_T from scratch)
  DevBuf<int32_t> C2;                            // [kUfbRowTile][Bp]: product of single rows (move outside the saved set)
  DevBuf<uint32_t> sel2;
  PinBuf<uint2> h_info;                          // host copy of info (same synchronisation as the scan results)
  DevBuf<int32_t> rt;                            // [Bp] REPS of the current tree (lengths)
  DevBuf<uint32_t> best;                         // [Bp] boot_score on the device (padding columns: 0 -> never an event)
  DevBuf<uint32_t> thr, home, cmin, pre, evcount;
  DevBuf<UfbEvent> ev;
  PinBuf<UfbEvent> h_ev;
  PinBuf<uint32_t> h_small;                      // staging: thr | home | best | event count
  // the staged block of a chained batch: words, layout and where it lies on the device (nullptr: not uploaded yet)
  bool st_valid = false;
  uint32_t st_n_idx = 0, st_n_parts = 0, st_n_self = 0, st_o_self = 0, st_words = 0;
  const uint32_t *st_dev = nullptr;
  PinBuf<uint32_t> p_flag_s[2], p_flag_e[2];     // pipelined climb: flags of the scan results / of the events, by batch parity
  PinBuf<int32_t> p_rt[2];
  PinBuf<UfbEvent> p_ev[2];
  PinBuf<uint32_t> h_flag;                       // [0] event count, [1] flag: written by the extraction kernel's last workgroup
  bool rt_valid = false;
  std::vector<int32_t> attach_wgt;               // pattern weights in force at attach time = IQTree's original_sample
  bool suspended = false;                        // weights in force that leave an attach-time pattern without a site: no bookkeeping
  // re-weighted (ratchet) climbs, reference iqtree.cpp:3283-3295: saveCurrentTree replaces cur_logl by the REPS of
  // _pattern_pars against original_sample BEFORE that array is refreshed for the candidate -- i.e. by the original-
  // alignment length of the tree booked last (or of the climb's start tree, which the IQ-TREE kernel left there)
  bool ratchet = false;                          // other weights than attach_wgt in force, every attach-time pattern still packed
  uint32_t stale_len = 0;                        // that length: what the cut-off filter sees and treels_logl records
  uint32_t rt_orig = 0;                          // original-alignment length of the current tree (host copy of rt[orig column])
  bool gate_closed = false;                      // a booked tree failed the cut-off: nothing else is booked in this climb
  bool ratchet_booking = true;                   // false = params->no_hclimb1_bb (iqtree.cpp:3280): re-weighted climbs are not booked
  // weighted (Sankoff) engine: a tentative tree's per-pattern lengths are not 0/1 increments of a mask -- the scan writes them
  // as 16-bit values (vals), k_vals_planes slices them into bit planes (bitp) and REPS = sum_k 2^k (plane k x weights); the
  // current tree's own row rides along in every product (it is the "home" row of the event formula, so s = C[candidate])
  bool snk = false;
  int Wp_s = 0;                                  // words per plane row (a multiple of 8), patterns in engine order
  DevBuf<uint16_t> vals;                         // [rows][npat]
  DevBuf<uint32_t> bitp;                         // [K][rows padded][Wp_s]
  DevBuf<uint32_t> vmax;                         // largest per-pattern length of the batch (atomic max)
  PinBuf<uint32_t> h_vmax;
  // -mulhits (params->multiple_hits, iqtree.cpp:3498-3540): per sample the SET of trees that reach its best REPS; trees of
  // one topology share the index of the first of them that hit (the reference's treels string map); no draws
  bool mulhits = false;
  // -storetrees (params->store_candidate_trees, iqtree.cpp:3302-3346): every tree that reaches saveCurrentTree is looked up in
  // topo_index first; one met before is not booked again unless its length improved on treels[index]
  bool store_trees = false;
  uint64_t duplicates = 0;                       // duplication_counter
  std::vector<std::set<int64_t>> hit_sets;                   // boot_trees_parsimony
  // -mulhits -topboot N (params->store_top_boot_trees, iqtree.cpp:3542-3585): per sample the N best NEW trees, best first, and
  // boot_threshold (INT_MIN + 1 until the first replacement in a full list, as in the reference)
  int topboot = 0;
  std::vector<std::vector<std::pair<int64_t, int32_t>>> top;
  std::vector<int32_t> top_thr;
  // -distinct_iter_top_boot k (params->distinct_iter_top_boot, iqtree.cpp:3587-3680; without -mulhits): the same list, at most k
  // entries, one representative per search iteration (cur_it = IQTree::curIt), accepted against boot_threshold with a
  // k / boot_counts tie draw
  int distinct = 0, cur_it = 0;
  std::vector<std::vector<int32_t>> top_iter;
  std::string self_key;                                      // canonical form of the current tree ...
  uint64_t self_key_epoch = ~0ull;                           // ... as of this topology epoch
  DevBuf<uint16_t> d_samples;                    // [Bl + 1][P]: the local samples as given + the row of original frequencies
  DevBuf<int32_t> d_first, d_cur;                // per pattern: first expanded site / weight of the packing in force
  DevBuf<int32_t> d_col;                         // one column of C, contiguous
  PinBuf<int32_t> h_col;
  PinBuf<int32_t> h_rt;                          // host copy of R_T (the current tree's own bookings are walked on the host)
};

class Engine {
 public:
  Engine() = default;
  ~Engine();
  int init(const mpf_config &cfg, const uint8_t *codes, const int32_t *weights, const uint32_t *cost = nullptr);

  // ---- alignment
  int set_weights(const int32_t *weights);
  int tip_vector(int tip, uint32_t *out);
  // HIP's current device is per host thread: every entry point re-selects the engine's device first
  void activate() const { (void)hipSetDevice(dev_); }
  bool broken() const { return broken_; }
  int tree_length_at_start(uint32_t *len) { return tree_length(len); }      // evaluateParsimony at start_'s edge, views as they are
  // evaluate (score_tree / pattern_scores) at another leaf's edge for the life of the guard: with an asymmetric cost matrix the
  // length of a tree depends on the edge it is rooted at (ParsTree::computeParsimony roots at IQ-TREE's `root` leaf)
  struct StartGuard {
    Engine &e;
    int saved;
    StartGuard(Engine &eng, int taxon) : e(eng), saved(eng.start_) { if (taxon >= 1) e.start_ = 3 * taxon; }
    ~StartGuard() { e.start_ = saved; }
  };
  int n() const { return n_; }
  int P() const { return P_; }
  int S() const { return sref_; }               // the reference's state count (the kernels' geometry may be wider: BIN in 4, GENERIC in 20)
  int Wref() const { return Wref_; }
  int Wp() const { return g_.Wp; }
  int n_informative() const { return ninf_; }
  const std::vector<int32_t> &informative() const { return inf_; }

  // ---- tree state (mirrors the PLL instance: back links, nodep order, start)
  int set_tree(const int32_t *back);
  void get_tree(int32_t *back) const;
  void reset_node_order();
  void node_rectifier();                       // sprparsimony.cpp:2046-2101

  // ---- scoring
  int score_tree(uint32_t *score);             // evaluateParsimony(start, full)
  int pattern_scores(uint16_t *ptn, int32_t *total);
  int site_scores(int32_t *site_pars, int n_sites, int32_t *total);   // pllComputeSiteParsimony
  int update_views();                          // make every directional vector of the current tree valid (syncs)
  int tree_length(uint32_t *len);              // from valid views
  // validity-tracked refresh: only invalid vectors that the given roots depend on are recomputed
  void invalidate_all();
  void invalidate_vectors();                   // vectors stale, topology (and what was planned from it) kept
  void invalidate_node(int node);              // every vector whose subtree contains `node`
  int schedule_views(const std::vector<int> *roots);   // enqueue (no sync); nullptr = every record of the tree
  bool dev_sched_usable() const;
  int schedule_views_dev(int sweep_maxtrav);
  int sweep_scan_dev(int maxtrav, uint64_t *n_tests, uint32_t *min_mp);
  void finish_views();                         // after a stream sync: subtree scores of the refreshed vectors
  void collect_scan_roots(int p, int mintrav, int maxtrav, std::vector<int> &roots) const;

  // ---- SPR neighbourhoods
  int plan_scan(int rec, int mintrav, int maxtrav, ScanPlan &plan);   // appends ops to the staging program
  int run_scans(std::vector<ScanPlan> &plans, std::vector<uint32_t> &out_host);
  // device-walked variant: the kernel enumerates the neighbourhood itself (k_scan_walk)
  int plan_walk(int rec, int mintrav, int maxtrav, ScanPlan &plan, bool split);
  int run_walks(std::vector<ScanPlan> &plans, const uint32_t **out_host);
  // ufb_async_ (the online tracker's climb batches): run_walks returns right behind the scan launch -- no copy-back, no
  // synchronisation -- with walk_async_ set; the caller enqueues the bookkeeping behind it, waits once for everything and then
  // calls run_walks_finish (scores of the refresh, base lengths, statistics)
  int run_walks_finish(std::vector<ScanPlan> &plans, const uint32_t **out_host);
  static bool wait_host_flag(const uint32_t *flag);
  struct WaitScope { WaitScope(); ~WaitScope(); };   // a host thread waiting for the device (counted: see wait_pause)
  static void wait_pause();                          // between two looks of a polling loop: yield, or sleep when the process is crowded
  int scan_batch(std::vector<ScanPlan> &plans, const int *recs, int count, int mintrav, int maxtrav, const uint32_t **out);
  int candidate_record(const ScanPlan &plan, size_t c);               // q of the c-th insertion test
  void enumerate_side(int x, int mintrav, int maxtrav, std::vector<int32_t> &q) const { enumerate_side(back_, x, mintrav, maxtrav, q); }
  void enumerate_side(const std::vector<int32_t> &bk, int x, int mintrav, int maxtrav, std::vector<int32_t> &q) const { books::enumerate_side(n_, bk, x, mintrav, maxtrav, q); }   // (on any topology)
  int spr_scan(int rec, int mintrav, int maxtrav, std::vector<int32_t> &q, std::vector<uint32_t> &mp, int &n_p);
  int sweep_scan(int mintrav, int maxtrav, uint64_t *n_tests, uint32_t *min_mp);
  int sweep_costs(int mintrav, int maxtrav, uint64_t cap, uint32_t *mp, uint64_t *offsets, uint64_t *n_tests);
  int node_order(int32_t *recs);

  // ---- search (host/search.cpp)
  void seed_ties(int mode, int seed) { tie_mode_ = mode; rng_.seed(seed); }
  void set_rand(double (*fn)(void *), void *arg) { rand_fn_ = fn; rand_arg_ = arg; }
  void set_tie_state(uint64_t s) { rng_.state = s; }
  uint64_t tie_state() const { return rng_.state; }
  int optimize_spr(int mintrav, int maxtrav, uint32_t *score);
  int make_parsimony_tree(int64_t seed, int spr_dist, uint32_t *score);
  int stepwise_addition(int64_t seed, uint32_t *best_per_step, int32_t *insert_per_step, uint32_t *score);
  const std::vector<Move> &moves() const { return moves_; }

  // ---- online UFBoot-MP bookkeeping (host/ufboot.cpp; reference IQTree::saveCurrentTree, iqtree.cpp:3271-3785)
  int ufboot_attach(int n_samples, const uint16_t *samples, double epsilon, int n_local = -1, const int32_t *sample_ids = nullptr,
                    mpf_ufb_exchange_fn exchange = nullptr, void *exchange_arg = nullptr);
  void ufboot_detach();
  // the first sweep of pllOptimizeSprParsimony from the current tree under every attached sample's weights at once
  int climb_fit_vw(bool one_workgroup = false);  // tile width k_climb will run with on this device (0: does not fit)
  // ClimbParams::batch_max of a k_climb launch: the plain climb's bound, or the quiet stretch's (a tracked climb under a cut-off)
  int climb_batch_bound(bool quiet) const { return std::max(std::max(1, std::min(climb_batch_min_, 16)), std::min(quiet ? climb_batch_max_sparse_ : climb_batch_max_, 16)); }
  int many_batch_max() const { return climb_batch_bound(false); }      // ClimbParams::batch_max of a climb of mpf_optimize_spr_many
  int ufboot_refine_sweep(int maxtrav, const int32_t *tie_seeds, uint32_t *scores, uint8_t *stable, int32_t *first_move_visit);
  int ufboot_set_mulhits(int on);
  int ufboot_set_cutoff_from_btrees(int on);
  int ufboot_orig_logl(int32_t *out) const;
  int ufboot_set_store_trees(int on);
  int ufboot_duplicates(uint64_t *n) const;
  int ufboot_set_topboot(int n_top);
  int ufboot_set_distinct_iter(int k);
  int ufboot_set_iteration(int cur_it);
  int ufboot_sample_iters(int sample, int32_t *iters, int cap, int *n) const;
  int ufboot_sample_top(int sample, int64_t *trees, int32_t *rell, int cap, int *n, int32_t *threshold) const;
  // the -topboot block of saveCurrentTree for one (tree, sample): true if the tree went onto the sample's list
  bool ufb_topboot_offer(uint32_t b, int32_t rell, int64_t tree_index, bool newly_added);
  template <class Lookup> bool ufb_distinct_offer(uint32_t b, int32_t rell, int64_t &tree_index, bool &looked_up, Lookup lookup);
  // what the device compares a (tree, sample) score with to decide whether the host must see it: the sample's best score
  // (default rule, -mulhits: start of a running minimum), or its fixed top-list bound (-topboot)
  uint32_t ufb_event_bound(uint32_t b) const
  {
    const UfbState &u = *ufb_;
    if (u.distinct) return u.top_thr[b] == -INT_MAX ? UINT32_MAX : (uint32_t)(-(int64_t)u.top_thr[b]);      // rell >= threshold
    if (!u.topboot) return u.boot_score[b];
    if ((int)u.top[b].size() < u.topboot || u.top_thr[b] == -INT_MAX) return UINT32_MAX;
    return (uint32_t)(-(int64_t)u.top_thr[b]) - 1u;              // rell > threshold  <=>  length <= -threshold - 1
  }
  int ufboot_sample_trees(int sample, int64_t *out, int cap, int *n) const;
  using CanonScratch = books::CanonScratch;
  void canonical_topology(const std::vector<int32_t> &bk, std::string &key, CanonScratch &sc) const { books::canonical_topology(n_, bk, key, sc); }
  void canonical_topology(const std::vector<int32_t> &bk, std::string &key) const { canonical_topology(bk, key, ct_); }
  mutable CanonScratch ct_;                      // the calling thread's scratch (the tracker's log worker brings its own)
  bool ufboot_attached() const { return (bool)ufb_; }
  int ufboot_set_cutoff(double logl_cutoff);
  int ufboot_set_ratchet_booking(int on);
  double ufboot_next_cutoff(int percent) const;
  int ufboot_n_samples() const;
  int64_t ufboot_n_trees() const;
  int ufboot_tree_logl(double *out) const;
  int ufboot_state(double *boot_logl, int32_t *boot_counts, int32_t *boot_trees) const;
  int ufboot_tree(int64_t tree_index, int32_t *back) const;
  hipError_t vec_store_fit(size_t bytes);
  size_t vec_cap_bytes_ = 0;
  int ufboot_adopt(int n_upd, const int32_t *sample, const uint32_t *score, const int32_t *tree_of, int n_trees, const int32_t *backs,
                   const uint32_t *lengths, int32_t *n_taken);
  int ufboot_counters(uint64_t *draws, uint64_t *events, uint64_t *gemm_rows, double *gemm_ms) const;

  mpf_stats stats{};
  int set_option(const std::string &key, int64_t v);
  int get_option(const std::string &key, int64_t *v) const;
  int scan_trace(uint64_t *out, uint64_t cap, uint64_t *n);

 private:
  // helpers
  inline int num(int r) const { return r / 3; }
  inline bool tip(int r) const { return r / 3 <= n_; }
  static inline int nx(int r) { int v = r / 3, s = r % 3; return 3 * v + (s + 1) % 3; }
  inline uint32_t slot(int r) const { int v = r / 3; return v <= n_ ? (uint32_t)(v - 1) : (uint32_t)(n_ + 3 * (v - n_ - 1) + r % 3); }
  inline void hookup(int a, int b) { back_[a] = b; back_[b] = a; }
  double tie_draw() { return rand_fn_ ? rand_fn_(rand_arg_) : rng_.next(); }

  int pack();                                   // compressDNA on the device
  void add_traverse(int q, int sib, int depth, int mintrav, int maxtrav, ScanPlan &plan);
  int spr_sweeps(int mintrav, int maxtrav, uint32_t randomMP, uint32_t *final_score);
  // where a sweep loop stands: handed from the plain loop to the tracked one when a climb under a logl_cutoff reaches the trees the
  // tracker books (host/search.cpp: spr_sweeps_run)
  struct SweepCursor { uint32_t startMP = 0, randomMP = 0; unsigned iter_hits = 1; int i = 1; bool stopped = false; };
  int spr_sweeps_run(int mintrav, int maxtrav, SweepCursor &cur, uint32_t stop_len);
  uint32_t climb_stop_len_ = 0;                  // ClimbParams::stop_len of the next k_climb launch
  int spr_sweeps_ufboot(int mintrav, int maxtrav, uint32_t randomMP, uint32_t *final_score);
  int spr_sweeps_ufboot_pipe(int mintrav, int maxtrav, uint32_t randomMP, uint32_t *final_score);
  int nv_waves_ = -1;                            // option "views_waves" (see Engine::set_option)
  int64_t climb_fault_ = 0;                      // tests: fault injected into the next k_climb launch (ClimbParams::fault)
  int climb_cus_ = 0;                            // CUs of the device (admission of persistent launches)
  bool broken_ = false;                          // a device launch did not come back: every further call fails
  int refine_chunk_ = 1 << 30;                   // prune nodes per masked scan + product of the refine sweep (option "refine_chunk")
  int spr_sweeps_ufboot_snk(int mintrav, int maxtrav, uint32_t randomMP, uint32_t *final_score);
  // online UFBoot-MP
  std::unique_ptr<UfbState> ufb_;
  // the tracker's large scratch buffers (candidate masks, product, events: ~1 GB at C3 x 1000 samples) outlive a detach, so
  // that re-attaching (every search iteration re-uses the engine) does not allocate inside the next climb
  struct UfbPool {
    DevBuf<uint32_t> masks, jmasks, sel2, thr, home, cmin, pre;
    PinBuf<uint32_t> h_small;
    DevBuf<uint2> info;
    DevBuf<int32_t> C, C2;
    DevBuf<UfbEvent> ev;
    PinBuf<UfbEvent> h_ev;
    PinBuf<uint2> h_info;
  } ufb_pool_;
  void ufb_pool_swap(UfbState &u);
  bool scan_masks_ = false;                      // the next scan_batch also writes candidate masks (k_scan_walk<MASKS>)
  bool ufb_async_ = false, walk_async_ = false;  // see run_walks_finish
  size_t walk_async_nd_ = 0, walk_async_nout_ = 0;
  int64_t ufb_stat_batches_ = 0, ufb_stat_early_ = 0;   // read-only options ufb_batches / ufb_early_batches: batches of the tracker's climbs since the engine was made, and how many of them were decided from the costs
  int ufb_pipe_ = 1;                             // option "ufb_pipe": the search's decision from the costs alone where they settle it, the next batch launched beside the bookkeeping of this one
  int ufb_cut_batch_ = 128;                      // option "ufb_cut_batch": smallest batch (prune nodes) of a tracked climb under a logl_cutoff
  int small_batch_max_ = 1 << 30;                // option "small_batch_max": batches above this many prune nodes take the whole-sweep path of scan_batch (all stale vectors refreshed, no closure on the host)
  // test aid (option "max_visits"): a climb returns behind this many prune-node visits, its state as that visit left it (0 = no
  // limit).  The persistent kernel is not used then; the pipelined tracked climb stops with its look-ahead batch discarded.
  int64_t max_visits_ = 0, visits_done_ = 0;
  bool visits_out() const { return max_visits_ > 0 && visits_done_ >= max_visits_; }
  int visits_cap(int i, int hi) const { return max_visits_ > 0 ? (int)std::min<int64_t>(hi, (int64_t)i + (max_visits_ - visits_done_) - 1) : hi; }
  int ufb_quiet_ = 1;                            // option "ufb_quiet": a climb under a logl_cutoff runs as the plain one until it reaches the trees the tracker books
  int64_t ufb_stat_quiet_ = 0;                   // read-only option ufb_quiet_climbs: tracked climbs that began with a quiet stretch
  int ufb_fast_ = 1;                             // option "ufb_fast": one dispatch chain and one wait per batch of the tracker's climbs
  uint32_t ufb_rows_ = 0;                        // rows (scan output indices) of the last masked scan
  int ufb_reserve_scan(size_t n_idx);
  void ufb_drain_log();
  // the current tree offered to every local sample under the default update rule with the deferred log (iqtree.cpp:3684-3731 for
  // 1000 samples x every prune-node visit: 2e6 bookings per move-less C3 sweep) -- one tight loop, the reciprocals from a table
  // samples for which an acceptance of the current tree would change nothing (see ufb_self_default)
  struct SelfMoot { std::vector<uint8_t> flag; int n_set = 0, n_le = -1, jump_n = -1; uint64_t jump_a = 1, jump_c = 0; };
  void ufb_self_default(const int32_t *rt, int64_t tree_index, int32_t cur_plan, bool &log_open, uint64_t &n_draws, SelfMoot *moot = nullptr);
  // the update rule of ONE booked tree for ONE sample and the booking of a tree itself, shared by the two-wait loop and the
  // weighted loop (host/ufboot_common.hpp); dc.on = the deferred mode of the default rule (acceptances go to the log)
  struct UfbDeferCtx { bool on = false; int32_t cur_plan = 0; bool *log_open = nullptr; SelfMoot *moot = nullptr; };
  template <class Lookup> void ufb_one_event(uint32_t b, uint32_t s, int64_t &tree_index, bool &looked_up, uint32_t cand_code, Lookup lookup, const UfbDeferCtx &dc);
  template <class KeyFn> int64_t ufb_book_tree(uint32_t cur_len, bool passes_cut, uint32_t cand_code, bool store_trees, KeyFn key);
  int ufb_moot_ = 1;                             // option "ufb_moot"
  int ufb_memo_ = 1;                             // option "ufb_memo": no product for the batches of a topology known to be event-free (UfbState::quiet_topo)
  // the log of one batch against an explicit topology: touches nothing of the engine but n_ and the tracker's deferred state
  // (topology map, boot_trees, reference counts, stored topologies), so that it can run on the worker thread of a pipelined climb
  using DrainScratch = books::DrainScratch;
  void ufb_drain(const std::vector<UfbState::LogEntry> &log, const std::vector<int32_t> &bk, int32_t epoch, const std::vector<ScanPlan> &plans,
                 DrainScratch &sc) { books::drain<ScanPlan>(n_, *ufb_, log, bk, epoch, plans, sc); }
  DrainScratch drain_scratch_;
  int64_t ufb_event_cap_ = 1 << 20;              // option "ufb_event_cap" (tests): first size of the event buffers -- small values exercise the overflow paths
  int ufb_thread_ = 1;                           // option "ufb_thread": the pipelined climb works its log off on a second host thread
  // chained batches: thr | home | best | self list of the batch's plans into UfbState::h_small (the staging block the extraction
  // kernel reads), so that it can go up with the refresh's own upload
  int ufb_stage_small(const std::vector<ScanPlan> &plans, int count);
  int ufb_current_tree_reps();                   // R_T of the current tree (join masks x weights, column sums)
  int ufb_layout_weights();                      // the product's right-hand side for the packing in force
  void ufb_store_tree(int64_t tree_index, int remove_rec, int insert_rec);
  void ufb_candidate_topology(int remove_rec, int insert_rec, std::vector<int32_t> &bk) const;
  void ufb_flush_pending(const ScanPlan &pl);

  int addition_phase(int64_t seed, uint32_t *best_per_step, int32_t *insert_per_step);
  void apply_move(int remove_rec, int insert_rec);

  // ---- device-resident climb (climb.hip; host/search.cpp): one k_climb launch runs the sweep loop from prune index *i on,
  // the moves it reports are replayed onto the host's topology mirror.  The host's view bookkeeping is reset afterwards
  // (the kernel keeps its own); whatever runs next on the host path starts with a full refresh (0.08 ms at C3).
  struct ClimbDev {
    DevBuf<uint16_t> bk, order;
    DevBuf<uint32_t> sct, trace, snap;           // snap / snap_r: ClimbParams::snap (launches of fewer workgroups than tiles)
    DevBuf<uint16_t> snap_r;
    DevBuf<unsigned long long> gsum;
    DevBuf<uint32_t> out;                        // [header | moves]
    uint32_t max_moves = 0;                      // capacity of the move list of the launch prepared last
    PinBuf<uint16_t> h_bk, h_order;
    PinBuf<uint32_t> h_out, h_beat;
    std::vector<uint32_t> h_trace;
    size_t trace_records = 0;
  } cd_;
  unsigned long long climb_phase_ticks_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, climb_ctr_[4] = {0, 0, 0, 0};
  int climb_device_ = 1;                         // 0 = host-driven batches only, 1 = device climb while moves are dense, 2 = always
  int climb_near_q_ = 3;                         // option climb_near_q: ClimbParams::near_q (mpf_optimize_spr_many: 1 unless the option was given)
  bool climb_near_q_set_ = false;
  bool climb_batch_min_set_ = false;             // option "climb_batch_min" was given: mpf_optimize_spr_many does not use its own (1)
  bool climb_vw_set_ = false;                    // option "climb_tile" was given: mpf_optimize_spr_many does not pick its own width
  int climb_vw_ = 1;                             // words per lane group: a tile is 16 x this many words (more tiles = shorter dependent chains per CU)
  int climb_batch_max_sparse_ = 16;              // option climb_batch_max_sparse: prune nodes per step of the quiet stretch of a tracked climb
  int climb_batch_min_ = 2, climb_batch_max_ = 8, climb_idle_ = 96, climb_trace_ = 0;
  bool climb_word_major_ = true;                 // option "climb_word_major": k_climb on 64-word tiles (climb_tile 4) runs four-state data in the word-major shape too (5 % faster; 0 = quads)
  bool many_word_major_ = true;                  // option "many_word_major": k_climb_many on 64-word tiles runs four-state data a word per lane (quadtile.hpp)
  int many_moves_cap_ = 0;                       // option "many_moves_cap" (tests): moves per launch of such a climb (0 = four sweeps' worth)
  bool many_sweeps_inside_ = true;               // option "many_sweeps_inside": a climb of mpf_optimize_spr_many runs all its sweeps in one launch (ClimbParams::sweeps_inside)
  int climb_groups_ = 0;                         // option "climb_groups": workgroups per k_climb launch (0 = one per tile; ClimbParams::groups)
  inline int rec_of(uint32_t cid) const { return cid < (uint32_t)n_ ? 3 * ((int)cid + 1) : 3 * (n_ + 1 + (int)(cid - (uint32_t)n_) / 3) + (int)((cid - (uint32_t)n_) % 3u); }
  int climb_segment(int maxtrav_eff, int total, int *i, uint32_t *randomMP, unsigned *iter_hits, bool may_idle, uint32_t *reason, uint32_t *n_moves);
  int climb_prepare(int maxtrav_eff, int total, int i, uint32_t randomMP, unsigned iter_hits, bool may_idle, int force_groups, hipStream_t st,
                    ClimbParams &p, int *vw_out, int *tiles_out, bool sweeps_inside, uint32_t start_mp);
  int climb_harvest(int total, int tiles, std::chrono::steady_clock::time_point t0, int *i, uint32_t *randomMP, unsigned *iter_hits, uint32_t *reason,
                    uint32_t *n_moves);
public:
  // pllOptimizeSprParsimony on MANY engines at once (independent climbs: start trees, bootstrap refinements): every sweep of every
  // climb is one workgroup of ONE launch (k_climb_many), a host thread feeds the lot (host/climb_host.cpp)
  static int climb_many(Engine **engs, int n, int mintrav, int maxtrav, uint32_t *scores);
  static int climb_many_round(Engine **engs, int n, int mintrav, int maxtrav, uint8_t *state, uint32_t *scores);
private:
  struct ManyState { uint32_t startMP = 0, randomMP = 0; unsigned iter_hits = 1; int i = 1; bool in_sweep = false; int tiles = 0; } many_;   // this engine's climb inside a batch
  struct ManyBufs { DevBuf<ClimbParams> d_params; PinBuf<ClimbParams> h_params; } many_bufs_;   // a batch's parameter blocks (held by the batch's first engine)

  // ---- device-resident stepwise addition (grow.hip; host/climb_host.cpp): one k_grow launch adds every taxon behind the start
  // tree, the insertions it reports are replayed onto the host's topology mirror
  struct GrowDev {
    DevBuf<uint16_t> init;
    DevBuf<uint32_t> xrow, park, ucp, out;
    PinBuf<uint16_t> h_init;
    PinBuf<uint32_t> h_out;
  } gd_;
  int grow_device_ = 1;                          // option "grow_device": 0 = the host's loop (one refresh + one scan + one round trip per taxon)
  int grow_vw_ = -1;                             // option "grow_tile": words per lane group of k_grow's quad tiles (-1 = fitted: at most 32 workgroups; 0 = the word-major DNA layout)
  int64_t grow_fault_ = 0;                       // tests: fault injected into the next k_grow launch
  uint64_t grow_launches_ = 0, grow_steps_ = 0;
  double grow_ms_total_ = 0;
  int grow_last_err_ = 0;                        // reason * 100 + err of the last launch that did not come back clean
  unsigned long long grow_phase_ticks_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int grow_fit_vw() const;
  int grow_segment(const std::vector<int> &perm, uint32_t len0, uint32_t *best_per_step, int32_t *insert_per_step, bool *done);

  // ---- configuration / alignment
  int n_ = 0, P_ = 0, datatype_ = 0, keep_all_ = 0, dev_ = 0, sref_ = 4, und_ = 15;
  int gmap_[32] = {0};                          // MPF_GENERIC: symbol code -> state row of the engine (-1: symbol not in use)
  Geometry g_{};
  int Wref_ = 0, nsites_ = 0, ninf_ = 0;
  std::vector<uint8_t> codes_;
  std::vector<int32_t> wgt_, inf_, first_site_;
  bool inf_known_ = false;                       // informative flags depend on the codes only: computed once
  // Sankoff mode
  bool sankoff_ = false;
  std::vector<uint32_t> cost_, cost_dev_, costT_dev_;
  bool asym_ = false;                            // the (repaired) cost matrix is not symmetric: evaluations are rooted as the reference roots them
  int force_big_ = 0;
  int snk16_opt_ = 1;                            // allow the packed 16-bit cost arithmetic when the values fit
  std::vector<int32_t> inf_index_;               // informative pattern j -> original pattern index
  DevBuf<uint32_t> d_cost_, d_costT_, d_pwgt_;
  DevBuf<int32_t> d_infidx_;
  size_t nslots_ = 0, vec_words_ = 0;

  hipStream_t st_ = nullptr;
  hipEvent_t ev0_ = nullptr, ev1_ = nullptr, ev2_ = nullptr, ev3_ = nullptr, ev4_ = nullptr;
  bool plan_event_pending_ = false;
  uint8_t *d_codes_ = nullptr;
  uint32_t *d_vec_ = nullptr, *d_tipslots_ = nullptr;
  // results buffer: [cnt: nslots][out: ...] so that one copy brings back both
  DevBuf<uint32_t> d_res_;
  PinBuf<uint32_t> h_res_;
  // refresh staging: [kids: nslots x 8 B][ops][level offsets] uploaded with ONE copy
  DevBuf<uint8_t> d_vstage_;
  PinBuf<uint8_t> h_vstage_;
  bool cnt_copy_pending_ = false;
  int host_poll_ = 1;                            // wait for host-written results by polling their flag word instead of a stream synchronisation
  int timing_ = 0;                               // HIP events around kernels: 1 = scan kernels, 2 = refresh kernels too (≈5 µs per event)
  // the candidate costs start on a 256-byte boundary and are cleared in whole 256-byte units: a memset of an
  // unaligned range is split by the runtime into up to three fill kernels (5 us each, per accepted move)
  size_t out_off() const { return (nslots_ + 63) & ~(size_t)63; }
  // (+1: word nout of the host's copy is the "results are there" flag of a launch that writes them itself)
  static size_t clear_words(size_t nout) { return ((nout ? nout : 1) + 1 + 63) & ~(size_t)63; }
  uint32_t *d_cnt() { return d_res_.p; }
  uint32_t *d_out() { return d_res_.p + out_off(); }
  uint32_t *h_cnt() { return h_res_.p; }
  uint32_t *h_out() { return h_res_.p + out_off(); }
  hipError_t reserve_results(size_t nout)
  {
    const size_t need = out_off() + clear_words(nout);
    if (need > d_res_.cap) {
      // grow without losing the mutation counts a pending refresh has already written to the head of the buffer
      DevBuf<uint32_t> bigger;
      hipError_t e = bigger.reserve(need + need / 2);
      if (e != hipSuccess) return e;
      if (d_res_.p && cnt_copy_pending_) {
        e = hipMemcpyAsync(bigger.p, d_res_.p, nslots_ * sizeof(uint32_t), hipMemcpyDeviceToDevice, st_);
        if (e != hipSuccess) return e;
        e = hipStreamSynchronize(st_);
        if (e != hipSuccess) return e;
      }
      std::swap(bigger.p, d_res_.p);
      std::swap(bigger.cap, d_res_.cap);
    }
    if (need > h_res_.cap) {
      PinBuf<uint32_t> bigger;
      hipError_t e = bigger.reserve(need + need / 2);
      if (e != hipSuccess) return e;
      std::swap(bigger.p, h_res_.p);
      std::swap(bigger.cap, h_res_.cap);
    }
    return hipSuccess;
  }
  DevBuf<int32_t> d_site2ptn_;
  size_t deep_scratch_words_ = (size_t)1 << 26;   // option "deep_scratch_kwords" (256 MB)
  DevBuf<uint32_t> d_deep_;                      // k_scan_deep's scratch (Geometry::deep_scratch), allocated by the first scan above kMaxDepth levels
  DevBuf<EvOp> d_evops_;
  DevBuf<ScanOp> d_scanops_;
  DevBuf<ScanHdr> d_scanhdr_;
  DevBuf<uint32_t> d_ncand_;
  DevBuf<uint32_t> d_cntp_;
  DevBuf<uint32_t> d_done_;                      // finished-workgroup counter of k_newview_wg (zero between launches)
  const uint2 *d_kids() const { return reinterpret_cast<const uint2 *>(d_vstage_.p); }
  std::vector<uint8_t> valid_;
  std::vector<int32_t> lev_, lev_epoch_;
  int32_t epoch_ = 0;
  std::vector<int> upd_order_;
  std::vector<int> sv_all_, sv_stack_, sv_order_, sv_fill_;      // schedule_views scratch
  std::vector<char> sv_seen_;
  std::vector<std::pair<int, int>> sv_pairs_;
  bool pending_scores_ = false, kids_dirty_ = true, view_events_pending_ = false;
  long n_invalid_ = -1;                         // -1 = unknown/many, 0 = every vector valid
  int split_below_ = 64;                        // batches of at most this many prune nodes are cut into 4 parts per scan
  int split_cands_ = 64;                        // larger batches: only neighbourhoods with more insertion tests than this are cut
  int views_mode_ = 2;                          // 2 = chained refresh, 1 = all levels in one launch, 0 = one launch per level
  struct ChainOp { int rec, other; };           // other < 0: chain head (both operands from memory)
  void build_chains(const std::vector<int> &order);
  std::vector<int> sv_idx_, ch_d0_, ch_d1_, ch_h_, ch_next_, ch_chain_, ch_head_, ch_len_, ch_slev_, ch_lev_off_, ch_sorted_, ch_wave_;
  std::vector<int32_t> ch_off_;
  std::vector<ChainOp> ch_ops_;
  int ch_levels_ = 0;
  long chain_max_ops_ = 512;                    // refreshes with more ops use the level kernel
  // small refreshes upload only ops, offsets, topology DELTAS and the following scan's descriptors (d_cstage_); the device
  // copy of the topology (d_vstage_) must have been uploaded whole once before
  bool kids_dev_ready_ = false;
  DevBuf<uint8_t> d_cstage_;
  // input of the scan launch that follows a refresh (walk descriptors, or a scan program and its headers): appended to the
  // refresh's own upload when one happens; dev = where it landed, nullptr if the scan has to upload it itself
  struct Ride { const void *src = nullptr; size_t bytes = 0; const void *dev = nullptr; };
  Ride ride_[2];
  std::vector<uint32_t> kid_upd_;
  std::vector<int> kids_list_;                  // records whose kids[] entry changed since the device copy was last complete
  bool kids_upload_ = false;
  // sweep_scan wants the best candidate per prune node only: the device reduces every scan part and writes the minima to the
  // host (k_part_min) instead of every candidate's cost coming back
  bool want_part_min_ = false, part_min_used_ = false;
  DevBuf<uint2> d_parts_;
  PinBuf<uint32_t> h_pmin_;
  PinBuf<uint2> h_parts_;                         // (pinned: uploaded without a stream synchronisation in the middle of a sweep)
  uint64_t parts_gen_ = ~0ull;                   // walk_gen_ the device copy of the part table belongs to
  size_t n_parts_dev_ = 0;
  bool scan_vals_ = false;                       // weighted tracker: host-planned scans also write per-pattern lengths, a slot per prune node is reserved for the current tree
  uint32_t vals_rows_ = 0;                       // output indices of the last such batch
  bool want_host_results_ = false, cnt_on_host_ = false;   // small batches: kernels write the host's result buffers themselves
  bool all_invalid_ = true;                     // no vector has been valid since the last wholesale invalidation
  std::vector<int> sb_roots_;
  uint32_t *zero_req_ptr_ = nullptr, *zeroed_ptr_ = nullptr;   // scan outputs the refresh launch is asked to clear / has cleared
  size_t zero_req_words_ = 0, zeroed_words_ = 0;
  uint64_t dbg_levels_ = 0;                     // MPF_VIEWS_PROFILE=1: levels summed over refreshes
  std::vector<uint2> kids_host_;
  DevBuf<WalkDesc> d_walk_;
  PinBuf<uint32_t> h_ncand_;
  PinBuf<WalkDesc> h_walk_;
  size_t n_walk_ = 0;
  uint32_t walk_out_ = 0;
  int scan_mode_ = 1;                           // 1 = device-walked scans, 0 = host-planned programs
  // 1 = batches of more than prog_min_descs_ descriptors run as planned programs (k_walk_plan + k_scan_prog), 2 = every batch, 0 = never
  int scan_prog_ = 1, prog_min_descs_ = 256;
  DevBuf<uint8_t> d_prog_;
  // what depends on the topology alone is kept while the topology stays ("plan_cache"): the schedule of a from-scratch
  // refresh (ops by level, in d_vstage_) and the plans / descriptors / device program of a whole sweep
  int plan_cache_ = 7;
  bool sched_cache_valid_ = false;
  size_t sc_nops_ = 0, sc_ops_off_ = 0, sc_lev_off_b_ = 0, sc_nlev_off_ = 0;
  int sc_maxlev_ = 0;                           // < 0: the schedule was made on the device, its level count lives at sc_nlev_off_
  bool dev_sched_ = true;                       // option dev_sched
  bool dev_plan_ = true;                        // option dev_plan: a whole sweep's scan descriptors laid out on the device too
  PinBuf<uint8_t> h_kstage_;                    // topology array + prune records of a device-scheduled refresh (read by k_sched from host memory)
  PinBuf<uint32_t> h_dsw_;                      // {parts, candidates, -, flag}, then the prune-node index of every part (written by k_sched)
  bool dsw_valid_ = false;                      // the descriptors on the device are k_sched's, for ...
  uint64_t dsw_walk_gen_ = 0;                   // ... this walk_gen_, ...
  uint64_t sched_gen_ = 0, dsw_sched_gen_ = ~0ull; // ... and the topology of this refresh schedule (counted per schedule made)
  int dsw_key_[4] = {0, 0, 0, 0};               // ... this radius / these options
  uint32_t dsw_parts_ = 0, dsw_out_ = 0;
  bool sched_on_dev_ = false;                   // d_vstage_ holds a k_sched-made schedule (diagnostics: options sched_levels / sched_ticks / sched_desc_ticks)
  bool shadow_ok_ = false;                      // every VALID vector has its word-major copy (Geometry::shoff) in place
  bool scan_shadow_ = true;                     // option scan_shadow
  bool plan_ride_ = true;                       // the walk plan of a device-planned sweep rides on the refresh launch
  std::vector<int> sc_order_;
  bool sweep_cache_valid_ = false, walk_dev_reuse_ = false;
  uint64_t walk_gen_ = 0, sweep_cache_gen_ = 0;
  uint64_t prog_gen_ = ~0ull;                    // walk_gen_ of the descriptors the program in d_prog_ was planned from
  size_t sweep_cache_nwalk_ = 0;
  uint32_t sweep_cache_out_ = 0;
  int sweep_cache_key_[6] = {0, 0, 0, 0, 0, 0};
  std::vector<int> sweep_cache_recs_;               // the prune records the cached sweep was planned for
  int scan_trace_ = 0;
  DevBuf<unsigned long long> d_trace_;
  size_t trace_words_ = 0;
  std::vector<uint32_t> out_scratch_;
  std::vector<ScanPlan> sweep_plans_;
  // sweep_costs: the result of a sizing call, handed out by the fetch call behind it
  std::vector<uint32_t> sc_keep_mp_;
  std::vector<uint64_t> sc_keep_off_;
  uint64_t sc_keep_key_[4] = {0, 0, 0, 0}, pack_gen_ = 0;
  bool sc_keep_valid_ = false;
  // number of records addTraverseParsimony visits below record q with m levels left (memo per topology epoch)
  std::vector<int32_t> nvis_val_, nvis_epoch_;
  int32_t topo_epoch_ = 1;
  int count_visits(int q, int m);
  void fill_visit_counts(int maxm);
  int32_t visits_filled_epoch_ = 0;
  std::vector<int32_t> vis_dense_, vis_a_;      // N(q, vis_dense_m_) for every record, valid while visits_filled_epoch_ == topo_epoch_
  int vis_dense_m_ = -1;
  int check_counts_ = 0;                         // 1 = copy the kernel's own candidate counts back and compare

  // staging program being built
  std::vector<ScanOp> prog_ops_;
  std::vector<ScanHdr> prog_hdr_;
  uint32_t prog_out_ = 0;
  int prog_max_depth_ = 0;

  // ---- tree
  std::vector<int32_t> back_, nodep_;
  int start_ = 3, ntips_ = 0, nextnode_ = 0;
  bool have_tree_ = false, views_valid_ = false;
  std::vector<uint32_t> sc_;                    // directional subtree score per record
  uint32_t tree_len_ = 0;

  // ---- search state (reference globals: bestParsimony, insertNode, removeNode, bestTreeScoreHits)
  uint32_t best_ = 0;
  int insert_rec_ = -1, remove_rec_ = -1;
  unsigned long hits_ = 1;
  int tie_mode_ = MPF_TIE_RANDOM;
  TieRng rng_;
  double (*rand_fn_)(void *) = nullptr;
  void *rand_arg_ = nullptr;
  int64_t randum_seed_ = 12345;
  std::vector<Move> moves_;
  int scan_batch_ = 16;                         // (C3 / C2 / C5 climbs from random trees: 16 is 4-8 % faster than 32, 8 slower again)
  // speculative batch size: prune nodes scanned per launch.  A batch is wasted behind the first accepted move, a small
  // batch costs a launch + synchronisation, so the size follows the observed distance between accepted moves (an
  // engine that has just refined a nearly optimal tree starts the next climb with whole sweeps).  Never changes a result.
  double gap_est_ = -1.0;                        // running estimate of prune nodes between accepted moves (-1: none yet)
  long since_move_ = 0;
  // a launch + synchronisation costs about as much as scanning ~60 prune nodes of a small batch; with moves g prune
  // nodes apart the cost per move, (g / b) launches + b / 2 wasted scans, is least near b = sqrt(2 * 60 * g); once
  // moves are more than a sweep apart (refinement of nearly optimal trees) whole sweeps are scanned
  int batch_for_gap(int total) const
  {
    const int lo = std::max(1, scan_batch_ / 4);
    if (gap_est_ >= (double)total) return total;
    if (gap_est_ < 64.0) return std::min(total, lo);          // busy climb: moves cluster, restart small and double
    const double b = std::sqrt(120.0 * std::max(gap_est_, 0.0));
    return std::min(total, std::max(lo, (int)std::min(b, 1e9)));
  }
  // a climb always ends with a sweep without moves: remember that long gaps are normal for this engine's trees
  void climb_finished(int total) { if (since_move_ >= total) gap_est_ = std::max(gap_est_, (double)since_move_); }
  int first_batch() const { return gap_est_ < 0 ? std::max(1, scan_batch_) : batch_for_gap(1 << 30); }
  int next_batch(int batch, bool moved, int consumed, int total)
  {
    since_move_ += consumed;
    if (moved) {
      // geometric running mean: one long gap (the end of a sweep) must not inflate the batches of a busy climb
      gap_est_ = gap_est_ < 0 ? (double)since_move_ : std::exp(0.7 * std::log(gap_est_ + 1.0) + 0.3 * std::log((double)since_move_ + 1.0)) - 1.0;
      since_move_ = 0;
      return batch_for_gap(total);
    }
    return std::min(total, batch * 2);
  }
};

}  // namespace mpf
