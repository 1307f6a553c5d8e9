// ufb_books.hpp -- the DEFERRED half of the online UFBoot bookkeeping, free of any device code.
//
// A tracked climb decides from the scans' costs and the samples' scores; what only has to be right at the END of a climb --
// which tree index a sample's boot_trees entry names, the reference counts of the saved trees, the stored topologies and the
// map from canonical topology to tree index (iqtree.cpp:3689-3707, :3720) -- is written to a log per batch and worked off
// later: by the climbing thread itself, or by a second host thread that owns this state for the length of a pipelined climb
// (LogWorker).  Everything here is plain C++ over explicit inputs (a topology as a `back` array, the batch's plans, the log):
// libmpfitch.so uses it through Engine::ufb_drain / Engine::canonical_topology / Engine::enumerate_side, and
// tests/cpu/ufb_books_test.cpp compiles the same header with g++ -fsanitize=thread and replays recorded and random logs inline
// and through the worker.
#pragma once
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include <pthread.h>
#include <sched.h>

namespace mpf {
namespace books {

// records: node v owns records 3v, 3v + 1, 3v + 2 (tips use the first); nx = pllNode::next
inline int nx(int r) { const int v = r / 3, s = r % 3; return 3 * v + (s + 1) % 3; }
inline bool is_tip(int r, int n_taxa) { return r / 3 <= n_taxa; }

struct CanonScratch { std::vector<int32_t> q, cp, mn, sz, off; };

// The canonical form of an unrooted topology: pre-order from the neighbour of tip 1, the subtree with the smaller tip first;
// tips by number, an inner node as 0 (16-bit entries while tip numbers fit, else 32-bit with -1).
// One call per booked tree that some sample accepts (3e4 in a C3 climb from a random tree; every insertion test with
// -storetrees), so: no allocation, no stack, selects instead of branches.  Three sequential sweeps over the nodes in
// breadth-first order from the neighbour of tip 1 (a node's children sit side by side at cp[i], cp[i] + 1): the order itself,
// then min tip and sequence length of every subtree backwards, then every node's position in the sequence forwards.
inline void canonical_topology(int n_taxa, const std::vector<int32_t> &bk, std::string &key, CanonScratch &sc)
{
  const int tipmax = 3 * n_taxa + 2;               // records of tips: r <= tipmax
  const size_t cap = 2 * (size_t)n_taxa + 8;
  if (sc.q.size() < cap) { sc.q.resize(cap); sc.cp.resize(cap); sc.mn.resize(cap); sc.sz.resize(cap); sc.off.resize(cap); }
  int32_t *q = sc.q.data(), *cp = sc.cp.data(), *mn = sc.mn.data(), *sz = sc.sz.data(), *off = sc.off.data();
  int len = 1;
  q[0] = bk[3];
  for (int i = 0; i < len; i++) {
    const int r = q[i];
    const bool inner = r > tipmax;
    const int base = (r / 3) * 3, sl = r - base;
    const int r1 = base + (sl == 2 ? 0 : sl + 1), r2 = base + (sl == 0 ? 2 : sl - 1);      // nx(r), nx(nx(r))
    q[len] = bk[(size_t)r1];                       // (a tip's other records are in bounds; what is read there is overwritten)
    q[len + 1] = bk[(size_t)r2];
    cp[i] = inner ? len : i;                       // (a tip points at itself: the selects below stay in bounds)
    len += inner ? 2 : 0;
  }
  for (int i = len - 1; i >= 0; i--) {
    const int r = q[i], c = cp[i];
    const bool inner = r > tipmax;
    const int c2 = inner ? c + 1 : c;
    mn[i] = inner ? std::min(mn[c], mn[c2]) : r / 3;
    sz[i] = inner ? 1 + sz[c] + sz[c2] : 1;
  }
  const bool narrow = n_taxa < 65535;
  key.resize((size_t)len * (narrow ? sizeof(uint16_t) : sizeof(int32_t)));
  uint16_t *k16 = reinterpret_cast<uint16_t *>(&key[0]);
  int32_t *k32 = reinterpret_cast<int32_t *>(&key[0]);
  off[0] = 0;
  for (int i = 0; i < len; i++) {
    const int r = q[i], c = cp[i], o = off[i];
    const bool inner = r > tipmax;
    if (narrow) k16[o] = inner ? (uint16_t)0 : (uint16_t)(r / 3);
    else k32[o] = inner ? -1 : r / 3;
    if (inner) {                                   // the subtree with the smaller tip first
      const bool swap = mn[c] > mn[c + 1];
      const int first = swap ? c + 1 : c, second = swap ? c : c + 1;
      off[first] = o + 1;
      off[second] = o + 1 + sz[first];
    }
  }
}

// the insertion branches of one side of a prune record in the reference's order (addTraverseParsimony, sprparsimony.cpp:
// 2208-2218, called as rearrangeParsimony calls it): q of candidate c = out[c]
inline void enumerate_side(int n_taxa, const std::vector<int32_t> &bk, int x, int mintrav, int maxtrav, std::vector<int32_t> &q)
{
  struct Fr { int q, d; };
  Fr st[520];                                      // (one entry more per level of the walk; a radius is at most 255)
  int sp = 0;
  if (maxtrav > 255) maxtrav = 255;
  const int x1 = bk[(size_t)nx(x)], x2 = bk[(size_t)nx(nx(x))];
  for (int side = 0; side < 2; side++) {
    const int a = side ? x2 : x1;
    if (is_tip(a, n_taxa)) continue;
    st[sp++] = Fr{bk[(size_t)nx(nx(a))], 1};
    st[sp++] = Fr{bk[(size_t)nx(a)], 1};
    while (sp) {
      const Fr f = st[--sp];
      if (f.d >= mintrav) q.push_back(f.q);
      if (!is_tip(f.q, n_taxa) && f.d < maxtrav) {
        st[sp++] = Fr{bk[(size_t)nx(nx(f.q))], f.d + 1};
        st[sp++] = Fr{bk[(size_t)nx(f.q)], f.d + 1};
      }
    }
  }
}

struct Pending { int64_t tree_index; uint32_t cand; };      // accepted during the current prune node, not yet materialised
struct LogEntry { uint32_t b, cand; int64_t tree; int32_t plan; };      // b = 0xFFFFFFFF: end of the scan of prune node `plan`

// what the log is worked off into
struct Deferred {
  std::vector<int64_t> boot_trees;                            // IQTree::boot_trees (iqtree.h:752): per sample, index of its best tree
  std::unordered_map<int64_t, std::vector<int32_t>> store;   // topologies of the trees some sample currently points to
  std::vector<int32_t> refs;                                  // per saved tree: number of samples whose boot_trees entry names it
  std::unordered_map<std::string, int64_t> topo_index;       // canonical topology -> tree index
};

struct DrainScratch {
  CanonScratch canon;
  std::vector<int32_t> bk, q_p, q_q;
  std::string key, self_key;                       // (the current tree's canonical form, valid for topology epoch self_epoch)
  int64_t self_epoch = -1;
  int q_plan = -1;
  std::vector<Pending> pending;
  // counters of the calling thread, added to the tracker's by whoever owns the scratch (the tracker's own words share cache
  // lines with what the replay counts on the other thread)
  uint64_t lookups = 0, stored = 0;
  double t_lookup = 0;
};

inline double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// The log of one batch against an explicit topology `bk` (the tree the batch was scanned on).  Plan: rec, n_p, n_total,
// maxtrav, mintrav_q, walked, cands[c].q -- what names the c-th insertion test of a prune node.
template <class Plan>
void drain(int n_taxa, Deferred &u, const std::vector<LogEntry> &log, const std::vector<int32_t> &bk, int32_t epoch, const std::vector<Plan> &plans,
           DrainScratch &sc)
{
  sc.q_plan = -1;
  // q of the c-th insertion test of plan j on THIS topology: both sides enumerated once per plan
  auto record_of = [&](int j, const Plan &pl, size_t c) -> int {
    if (!pl.walked) return pl.cands[c].q;
    if (sc.q_plan != j) {
      sc.q_p.clear();
      sc.q_q.clear();
      if (pl.n_p > 0) enumerate_side(n_taxa, bk, pl.rec, 1, pl.maxtrav, sc.q_p);
      if (pl.n_total > pl.n_p) enumerate_side(n_taxa, bk, bk[(size_t)pl.rec], pl.mintrav_q, pl.maxtrav, sc.q_q);
      sc.q_plan = j;
    }
    return (int)c < pl.n_p ? sc.q_p[c] : sc.q_q[c - (size_t)pl.n_p];
  };
  auto topology_of = [&](const Plan &pl, int ins, uint32_t cand, std::vector<int32_t> &out) {       // the tree with candidate `cand` applied
    const int p = cand < (uint32_t)pl.n_p ? pl.rec : bk[(size_t)pl.rec];
    out = bk;
    auto hk = [&](int a, int b2) { out[(size_t)a] = b2; out[(size_t)b2] = a; };
    const int a = out[(size_t)nx(p)], b2 = out[(size_t)nx(nx(p))];
    hk(a, b2);
    const int r = out[(size_t)ins];
    hk(nx(p), ins);
    hk(nx(nx(p)), r);
  };
  auto need_ref = [&](int64_t t) { if (u.refs.size() <= (size_t)t) u.refs.resize((size_t)t + 1 + u.refs.size() / 2, 0); };
  int64_t raw = -1, resolved = -1;
  for (const LogEntry &le : log) {
    const Plan &pl = plans[(size_t)le.plan];
    if (le.b == 0xFFFFFFFFu) {
      // end of this prune node's scan: the topologies accepted during it that some sample still points to
      for (const Pending &pe : sc.pending) {
        if (u.refs[(size_t)pe.tree_index] <= 0 || u.store.count(pe.tree_index)) continue;
        if (pe.cand == 0xFFFFFFFFu) u.store.emplace(pe.tree_index, bk);
        else {
          topology_of(pl, record_of(le.plan, pl, (size_t)pe.cand), pe.cand, sc.bk);
          u.store.emplace(pe.tree_index, sc.bk);
        }
        sc.stored++;
      }
      sc.pending.clear();
      continue;
    }
    if (le.tree != raw) {
      raw = le.tree;
      const double tl = now_ms();
      const std::string *key = &sc.self_key;
      if (le.cand == 0xFFFFFFFFu) {
        if (sc.self_epoch != (int64_t)epoch) { canonical_topology(n_taxa, bk, sc.self_key, sc.canon); sc.self_epoch = (int64_t)epoch; }
      } else {
        topology_of(pl, record_of(le.plan, pl, (size_t)le.cand), le.cand, sc.bk);
        canonical_topology(n_taxa, sc.bk, sc.key, sc.canon);
        key = &sc.key;
      }
      resolved = u.topo_index.emplace(*key, raw).first->second;
      sc.t_lookup += now_ms() - tl;
      sc.lookups++;
    }
    need_ref(resolved);
    if (sc.pending.empty() || sc.pending.back().tree_index != resolved) sc.pending.push_back(Pending{resolved, le.cand});
    int64_t &bt = u.boot_trees[le.b];
    if (bt != resolved) {
      if (bt >= 0 && --u.refs[(size_t)bt] == 0) u.store.erase(bt);
      u.refs[(size_t)resolved]++;
      bt = resolved;
    }
  }
}

// The second host thread of a pipelined climb: it owns `Deferred` from the first submit() to finish() and works on copies of
// the topology and the plans; the climbing thread never looks at that state in between.
template <class Plan>
struct LogWorker {
  struct Job { std::vector<LogEntry> log; std::vector<int32_t> back; int32_t epoch = 0; std::vector<Plan> plans; };
  int n_taxa = 0;
  Deferred *d = nullptr;
  std::thread th;
  std::mutex m;
  std::condition_variable cv;
  std::deque<Job *> q;
  std::vector<Job *> spare;
  std::vector<std::unique_ptr<Job>> all;
  size_t inflight = 0;
  bool stop = false, started = false;
  bool pin = true;                                 // keep the worker on the cores that share the submitting thread's last-level cache
  DrainScratch sc;
  void run()
  {
    for (;;) {
      Job *j = nullptr;
      {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return stop || !q.empty(); });
        if (q.empty()) return;
        j = q.front();
        q.pop_front();
      }
      drain<Plan>(n_taxa, *d, j->log, j->back, j->epoch, j->plans, sc);
      j->log.clear();
      {
        std::lock_guard<std::mutex> lk(m);
        spare.push_back(j);
        inflight--;
      }
      cv.notify_all();
    }
  }
  Job *get()
  {
    std::unique_lock<std::mutex> lk(m);
    cv.wait(lk, [&] { return inflight < 256; });
    if (spare.empty()) { all.emplace_back(new Job()); return all.back().get(); }
    Job *j = spare.back();
    spare.pop_back();
    return j;
  }
  bool submit(Job *j)                              // false: no second thread to be had -- the caller works the job off itself
  {
    if (!started) {
      { std::lock_guard<std::mutex> lk(m); stop = false; }
      try { th = std::thread([this] { run(); }); } catch (...) { return false; }
      started = true;
      // (best effort) the log, the plans and the topology copies change hands every batch, and on a two-socket host a worker on
      // the other socket made the replay -- which then writes into lines the worker owns -- 2.4 times slower than it is alone
      const int cpu = pin ? sched_getcpu() : -1;
      if (cpu >= 0) {
        char path[128];
        std::snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%d/cache/index3/shared_cpu_list", cpu);
        if (FILE *fp = std::fopen(path, "r")) {
          char buf[256] = {0};
          if (std::fgets(buf, sizeof buf, fp)) {
            cpu_set_t set, allowed;
            CPU_ZERO(&set);
            CPU_ZERO(&allowed);
            // only CPUs the caller itself may run on (taskset / numactl / a launcher's per-rank binding): the worker never leaves the
            // mask its process was given; no such sibling -> no pinning
            const bool have_mask = sched_getaffinity(0, sizeof allowed, &allowed) == 0;
            int n_set = 0;
            for (char *p = buf; *p && *p != '\n';) {
              char *end = nullptr;
              const long a = std::strtol(p, &end, 10);
              if (end == p) break;
              long b = a;
              p = end;
              if (*p == '-') { b = std::strtol(p + 1, &end, 10); p = end; }
              for (long c = a; c <= b && c < CPU_SETSIZE; c++)
                if (c != cpu && (!have_mask || CPU_ISSET((int)c, &allowed))) { CPU_SET((int)c, &set); n_set++; }
              if (*p == ',') p++;
            }
            if (n_set > 0) (void)pthread_setaffinity_np(th.native_handle(), sizeof set, &set);
          }
          std::fclose(fp);
        }
      }
    }
    { std::lock_guard<std::mutex> lk(m); q.push_back(j); inflight++; }
    cv.notify_all();
    return true;
  }
  void finish()
  {
    if (!started) return;
    { std::unique_lock<std::mutex> lk(m); cv.wait(lk, [&] { return inflight == 0; }); stop = true; }
    cv.notify_all();
    if (th.joinable()) th.join();
    started = false;
  }
  ~LogWorker() { finish(); }
};

// ---- recorded streams: what a pipelined climb handed its worker, as a file (MPF_UFB_RECORD=<path>: host/ufboot.cpp writes it on
// the GPU box; tests/cpu/ufb_books_test.cpp replays it).  Layout: "UFBREC3\0", n_taxa, then records: 'D' + state (at the start of
// a climb), 'J' + one job ..., 'E' + state (after the worker was joined); little-endian, no padding.
namespace rec {
inline void put(std::FILE *f, const void *p, size_t n) { if (n && std::fwrite(p, 1, n, f) != n) std::perror("ufb record"); }
template <class T> inline void put1(std::FILE *f, T v) { put(f, &v, sizeof v); }
inline bool get(std::FILE *f, void *p, size_t n) { return n == 0 || std::fread(p, 1, n, f) == n; }
template <class T> inline bool get1(std::FILE *f, T &v) { return get(f, &v, sizeof v); }

// The state in a record: boot_trees in full, the non-zero reference counts, and DIGESTS of the stored topologies and of the
// topology map (count + FNV-1a over the entries in key order) -- the maps of a long run hold tens of thousands of topologies;
// a replay carries them from climb to climb itself and is checked against the digests.
struct StateRecord {
  std::vector<int64_t> boot_trees;
  std::vector<std::pair<int64_t, int32_t>> refs;     // (tree, count), count != 0
  int64_t n_refs = 0, n_store = 0, n_topo = 0;
  uint64_t h_store = 0, h_topo = 0;
  bool operator==(const StateRecord &o) const
  {
    return boot_trees == o.boot_trees && refs == o.refs && n_store == o.n_store && n_topo == o.n_topo && h_store == o.h_store && h_topo == o.h_topo;
  }
};
inline uint64_t fnv(uint64_t h, const void *p, size_t n)
{
  const unsigned char *c = static_cast<const unsigned char *>(p);
  for (size_t i = 0; i < n; i++) { h ^= c[i]; h *= 1099511628211ull; }
  return h;
}
inline StateRecord state_of(const Deferred &d, size_t n_trees)
{
  StateRecord r;
  r.boot_trees = d.boot_trees;
  const size_t nr = std::min(n_trees, d.refs.size());          // (the worker grows refs in strides: only the saved trees' entries mean something)
  r.n_refs = (int64_t)n_trees;
  for (size_t t = 0; t < nr; t++) if (d.refs[t] != 0) r.refs.emplace_back((int64_t)t, d.refs[t]);
  std::vector<int64_t> keys;
  for (const auto &kv : d.store) keys.push_back(kv.first);
  std::sort(keys.begin(), keys.end());
  r.n_store = (int64_t)keys.size();
  r.h_store = 1469598103934665603ull;
  for (int64_t k : keys) {
    const std::vector<int32_t> &b = d.store.at(k);
    r.h_store = fnv(r.h_store, &k, sizeof k);
    r.h_store = fnv(r.h_store, b.data(), b.size() * sizeof(int32_t));
  }
  std::vector<const std::pair<const std::string, int64_t> *> topo;
  for (const auto &kv : d.topo_index) topo.push_back(&kv);
  std::sort(topo.begin(), topo.end(), [](const auto *x, const auto *y) { return x->first < y->first; });
  r.n_topo = (int64_t)topo.size();
  r.h_topo = 1469598103934665603ull;
  for (const auto *kv : topo) {
    r.h_topo = fnv(r.h_topo, kv->first.data(), kv->first.size());
    r.h_topo = fnv(r.h_topo, &kv->second, sizeof kv->second);
  }
  return r;
}
inline void write_state(std::FILE *f, char tag, const Deferred &d, size_t n_trees)
{
  const StateRecord r = state_of(d, n_trees);
  put1<char>(f, tag);
  put1<int64_t>(f, (int64_t)r.boot_trees.size());
  put(f, r.boot_trees.data(), r.boot_trees.size() * sizeof(int64_t));
  put1<int64_t>(f, r.n_refs);
  put1<int64_t>(f, (int64_t)r.refs.size());
  for (const auto &kv : r.refs) { put1<int64_t>(f, kv.first); put1<int32_t>(f, kv.second); }
  put1<int64_t>(f, r.n_store); put1<uint64_t>(f, r.h_store);
  put1<int64_t>(f, r.n_topo); put1<uint64_t>(f, r.h_topo);
}
inline bool read_state(std::FILE *f, StateRecord &r)          // (the tag has been read)
{
  int64_t n = 0, nz = 0;
  r = StateRecord();
  if (!get1(f, n)) return false;
  r.boot_trees.resize((size_t)n);
  if (!get(f, r.boot_trees.data(), (size_t)n * sizeof(int64_t))) return false;
  if (!get1(f, r.n_refs) || !get1(f, nz)) return false;
  for (int64_t i = 0; i < nz; i++) {
    int64_t t = 0;
    int32_t v = 0;
    if (!get1(f, t) || !get1(f, v)) return false;
    r.refs.emplace_back(t, v);
  }
  return get1(f, r.n_store) && get1(f, r.h_store) && get1(f, r.n_topo) && get1(f, r.h_topo);
}
template <class Plan>
void write_job(std::FILE *f, const std::vector<LogEntry> &log, const std::vector<int32_t> &back, int32_t epoch, const std::vector<Plan> &plans)
{
  put1<char>(f, 'J');
  put1<int32_t>(f, epoch);
  put1<int32_t>(f, (int32_t)back.size());
  put(f, back.data(), back.size() * sizeof(int32_t));
  put1<int32_t>(f, (int32_t)plans.size());
  for (const Plan &pl : plans) {
    const int32_t h[6] = {pl.rec, pl.n_p, pl.n_total, pl.maxtrav, pl.mintrav_q, pl.walked ? 1 : 0};
    put(f, h, sizeof h);
    put1<int32_t>(f, (int32_t)pl.cands.size());
    for (const auto &c : pl.cands) put1<int32_t>(f, (int32_t)c.q);
  }
  put1<int64_t>(f, (int64_t)log.size());
  put(f, log.data(), log.size() * sizeof(LogEntry));
}
template <class Plan>
bool read_job(std::FILE *f, std::vector<LogEntry> &log, std::vector<int32_t> &back, int32_t &epoch, std::vector<Plan> &plans)      // (tag read)
{
  int32_t n = 0;
  if (!get1(f, epoch) || !get1(f, n)) return false;
  back.resize((size_t)n);
  if (!get(f, back.data(), (size_t)n * sizeof(int32_t)) || !get1(f, n)) return false;
  plans.assign((size_t)n, Plan());
  for (Plan &pl : plans) {
    int32_t h[6], nc = 0;
    if (!get(f, h, sizeof h) || !get1(f, nc)) return false;
    pl.rec = h[0]; pl.n_p = h[1]; pl.n_total = h[2]; pl.maxtrav = h[3]; pl.mintrav_q = h[4]; pl.walked = h[5] != 0;
    pl.cands.resize((size_t)nc);
    for (auto &c : pl.cands) { int32_t q = 0; if (!get1(f, q)) return false; c.q = q; }
  }
  int64_t nl = 0;
  if (!get1(f, nl)) return false;
  log.resize((size_t)nl);
  return get(f, log.data(), (size_t)nl * sizeof(LogEntry));
}
}  // namespace rec

}  // namespace books
}  // namespace mpf
