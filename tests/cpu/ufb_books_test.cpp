// CPU-only check of the deferred half of the online UFBoot bookkeeping (mpboot_amd/host/ufb_books.hpp -- the very code
// libmpfitch.so runs): built by tests/test_ufb_books.py with g++ -fsanitize=thread.
//   ufb_books_test replay <file>   : a stream recorded on the GPU box (MPF_UFB_RECORD) -- every climb's jobs are worked off
//                                    (a) inline on this thread, (b) through LogWorker on a second thread, (c) through LogWorker with
//                                    a queue that is kept short (submit / get contend), and all three must end in the recorded state
//   ufb_books_test random <seed> <batches> : random trees, plans and logs (no recording needed): inline == worker
// Exit code 0 = all equal (and the thread sanitizer, if compiled in, saw no race: it makes the exit code non-zero itself).
#include <cstdio>
#include <cstring>
#include <random>

#include "../../mpboot_amd/host/ufb_books.hpp"

using namespace mpf::books;

struct Cand { int q = 0; };
struct Plan { int rec = 0, n_p = 0, n_total = 0, maxtrav = 0, mintrav_q = 2; bool walked = true; std::vector<Cand> cands; };
struct JobData { std::vector<LogEntry> log; std::vector<int32_t> back; int32_t epoch = 0; std::vector<Plan> plans; };

static bool same(const Deferred &a, const Deferred &b, size_t n_trees, const char *what)
{
  bool ok = a.boot_trees == b.boot_trees;
  if (!ok) std::fprintf(stderr, "%s: boot_trees differ\n", what);
  for (size_t t = 0; t < n_trees; t++) {
    const int32_t ra = t < a.refs.size() ? a.refs[t] : 0, rb = t < b.refs.size() ? b.refs[t] : 0;
    if (ra != rb) { std::fprintf(stderr, "%s: refs[%zu] %d vs %d\n", what, t, ra, rb); ok = false; break; }
  }
  if (a.store.size() != b.store.size()) { std::fprintf(stderr, "%s: %zu vs %zu stored topologies\n", what, a.store.size(), b.store.size()); ok = false; }
  for (const auto &kv : a.store) {
    auto it = b.store.find(kv.first);
    if (it == b.store.end() || it->second != kv.second) { std::fprintf(stderr, "%s: stored topology of tree %lld differs\n", what, (long long)kv.first); ok = false; break; }
  }
  if (a.topo_index != b.topo_index) { std::fprintf(stderr, "%s: topology maps differ (%zu vs %zu)\n", what, a.topo_index.size(), b.topo_index.size()); ok = false; }
  return ok;
}

static void run_inline(int n, Deferred &d, const std::vector<JobData> &jobs)
{
  DrainScratch sc;
  for (const JobData &j : jobs) drain<Plan>(n, d, j.log, j.back, j.epoch, j.plans, sc);
}

static void run_worker(int n, Deferred &d, const std::vector<JobData> &jobs, bool crowd)
{
  LogWorker<Plan> w;
  w.n_taxa = n;
  w.d = &d;
  w.pin = false;
  size_t k = 0;
  for (const JobData &j : jobs) {
    LogWorker<Plan>::Job *jb = w.get();
    jb->log = j.log;
    jb->back = j.back;
    jb->epoch = j.epoch;
    jb->plans = j.plans;
    if (!w.submit(jb)) { drain<Plan>(n, d, jb->log, jb->back, jb->epoch, jb->plans, w.sc); jb->log.clear(); w.spare.push_back(jb); }
    if (crowd && (++k % 7) == 0) { w.finish(); }       // (join and start again in mid-climb: the hand-over of the state both ways)
  }
  w.finish();
}

static bool matches(const rec::StateRecord &want, const Deferred &got, const char *what)
{
  const rec::StateRecord r = rec::state_of(got, (size_t)want.n_refs);
  if (r == want) return true;
  std::fprintf(stderr, "%s: boot_trees %s, refs %s (%zu vs %zu non-zero), stored topologies %lld vs %lld (%s), topology map %lld vs %lld (%s)\n", what,
               r.boot_trees == want.boot_trees ? "equal" : "DIFFER", r.refs == want.refs ? "equal" : "DIFFER", r.refs.size(), want.refs.size(),
               (long long)r.n_store, (long long)want.n_store, r.h_store == want.h_store ? "equal" : "DIFFER", (long long)r.n_topo, (long long)want.n_topo,
               r.h_topo == want.h_topo ? "equal" : "DIFFER");
  return false;
}

static int replay_file(const char *path)
{
  std::FILE *f = std::fopen(path, "rb");
  if (!f) { std::perror(path); return 2; }
  char magic[8];
  int32_t n = 0;
  if (!rec::get(f, magic, 8) || std::memcmp(magic, "UFBREC3", 8) != 0 || !rec::get1(f, n)) { std::fprintf(stderr, "%s: not a recording\n", path); return 2; }
  int climbs = 0;
  size_t batches = 0, entries = 0;
  // the state is carried from climb to climb (a record holds digests of the large maps): three copies, one per way through
  Deferred cur[3];
  bool have_state = false, in_climb = false, ok = true;
  std::vector<JobData> jobs;
  rec::StateRecord st;
  char tag;
  while (rec::get1(f, tag)) {
    if (tag == 'D') {
      if (!rec::read_state(f, st)) { std::fprintf(stderr, "truncated recording\n"); return 2; }
      const bool empty = st.n_store == 0 && st.n_topo == 0 && st.refs.empty() && std::all_of(st.boot_trees.begin(), st.boot_trees.end(), [](int64_t t) { return t < 0; });
      if (empty) {                                 // a tracker that has just been attached
        for (Deferred &d : cur) { d = Deferred(); d.boot_trees = st.boot_trees; }
        have_state = true;
      } else if (!have_state) {
        std::fprintf(stderr, "the recording does not start from an empty tracker\n");
        return 2;
      } else if (!matches(st, cur[0], "start of a climb vs the state carried over (was the tracker used between the recorded climbs?)")) return 1;
      jobs.clear();
      in_climb = true;
    } else if (tag == 'J') {
      jobs.emplace_back();
      JobData &j = jobs.back();
      if (!in_climb || !rec::read_job<Plan>(f, j.log, j.back, j.epoch, j.plans)) { std::fprintf(stderr, "truncated recording\n"); return 2; }
      entries += j.log.size();
    } else if (tag == 'E') {
      if (!in_climb || !rec::read_state(f, st)) { std::fprintf(stderr, "truncated recording\n"); return 2; }
      run_inline(n, cur[0], jobs);
      run_worker(n, cur[1], jobs, false);
      run_worker(n, cur[2], jobs, true);
      ok = matches(st, cur[0], "recorded vs inline") && ok;
      ok = matches(st, cur[1], "recorded vs worker") && ok;
      ok = matches(st, cur[2], "recorded vs worker, joined in mid-climb") && ok;
      climbs++;
      batches += jobs.size();
      in_climb = false;
    } else { std::fprintf(stderr, "bad tag %d\n", (int)tag); return 2; }
  }
  std::fclose(f);
  std::printf("%s: %d taxa, %d climbs, %zu batches, %zu log entries replayed three ways: %s\n", path, n, climbs, batches, entries, ok ? "equal" : "DIFFERENT");
  return ok && climbs > 0 ? 0 : 1;
}

// ---- random streams
static std::vector<int32_t> random_tree(int n, std::mt19937_64 &g)
{
  // records 3v + s; tips 1..n use record 3v, inner nodes n+1..2n-2 use all three.  Grown by random insertion.
  std::vector<int32_t> bk(3 * (size_t)(2 * n - 1) + 3, -1);
  auto hk = [&](int a, int b) { bk[(size_t)a] = b; bk[(size_t)b] = a; };
  int inner = n + 1;
  hk(3 * 1, 3 * inner); hk(3 * 2, 3 * inner + 1); hk(3 * 3, 3 * inner + 2);
  std::vector<int> edges = {3 * 1, 3 * 2, 3 * 3};   // one record per branch
  for (int t = 4; t <= n; t++) {
    inner++;
    const int e = edges[(size_t)(g() % edges.size())], o = bk[(size_t)e];
    hk(e, 3 * inner); hk(o, 3 * inner + 1); hk(3 * t, 3 * inner + 2);
    edges.push_back(3 * inner + 1);
    edges.push_back(3 * t);
  }
  return bk;
}

static int random_streams(uint64_t seed, int n_batches)
{
  std::mt19937_64 g(seed);
  const int n = 8 + (int)(g() % 60), B = 5 + (int)(g() % 40);
  std::vector<int32_t> bk = random_tree(n, g);
  std::vector<JobData> jobs;
  int64_t next_tree = 0;
  int32_t epoch = 1;
  for (int k = 0; k < n_batches; k++) {
    JobData j;
    j.back = bk;
    j.epoch = epoch;
    const int np = 1 + (int)(g() % 6);
    for (int p = 0; p < np; p++) {
      Plan pl;
      // a prune record whose both sides exist on this tree
      for (;;) {
        const int v = n + 1 + (int)(g() % (uint64_t)(n - 2));
        pl.rec = 3 * v + (int)(g() % 3);
        pl.maxtrav = 1 + (int)(g() % 6);
        std::vector<int32_t> qp, qq;
        enumerate_side(n, bk, pl.rec, 1, pl.maxtrav, qp);
        if (!is_tip(bk[(size_t)pl.rec], n)) enumerate_side(n, bk, bk[(size_t)pl.rec], 2, pl.maxtrav, qq);
        pl.n_p = (int)qp.size();
        pl.n_total = (int)(qp.size() + qq.size());
        if (pl.n_total > 0) break;
      }
      // the visit's log: the current tree, then some candidates, each accepted by a few samples; then the end mark
      const int n_trees = 1 + (int)(g() % 4);
      for (int t = 0; t < n_trees; t++) {
        const uint32_t cand = t == 0 ? 0xFFFFFFFFu : (uint32_t)(g() % (uint64_t)pl.n_total);
        const int64_t tree = next_tree++;
        const int hits = (int)(g() % 4);
        for (int h = 0; h < hits; h++) j.log.push_back(LogEntry{(uint32_t)(g() % (uint64_t)B), cand, tree, (int32_t)j.plans.size()});
      }
      j.log.push_back(LogEntry{0xFFFFFFFFu, 0u, 0, (int32_t)j.plans.size()});
      j.plans.push_back(pl);
    }
    jobs.push_back(std::move(j));
    if (g() % 3 == 0) {                            // the climb moves on: another topology (a random SPR = take any tree; the epoch changes)
      bk = random_tree(n, g);
      epoch++;
    }
  }
  Deferred start;
  start.boot_trees.assign((size_t)B, -1);
  Deferred a = start, b = start, c = start;
  run_inline(n, a, jobs);
  run_worker(n, b, jobs, false);
  run_worker(n, c, jobs, true);
  const size_t nt = (size_t)next_tree;
  const bool ok = same(a, b, nt, "inline vs worker") && same(a, c, nt, "inline vs worker, joined in mid-climb");
  size_t pointed = 0;
  for (int64_t t : a.boot_trees) pointed += t >= 0;
  std::printf("seed %llu: %d taxa, %d samples, %d batches, %lld trees, %zu samples point at one, %zu topologies stored: %s\n", (unsigned long long)seed, n, B, n_batches,
              (long long)next_tree, pointed, a.store.size(), ok ? "equal" : "DIFFERENT");
  return ok ? 0 : 1;
}

// ---- canonical form: the same unrooted topology under any numbering of its inner nodes and any rotation of their records
// gives the same key; another topology (checked by its set of bipartitions) gives another
static std::vector<int32_t> relabel(int n, const std::vector<int32_t> &bk, std::mt19937_64 &g)
{
  const int first = n + 1, last = 2 * n - 2;
  std::vector<int> perm((size_t)(last - first + 1));
  for (size_t i = 0; i < perm.size(); i++) perm[i] = first + (int)i;
  std::shuffle(perm.begin(), perm.end(), g);
  std::vector<int> rot(perm.size());
  for (int &r : rot) r = (int)(g() % 3);
  auto map = [&](int rec) {
    const int v = rec / 3, s = rec % 3;
    if (v <= n) return rec;
    return 3 * perm[(size_t)(v - first)] + (s + rot[(size_t)(v - first)]) % 3;
  };
  std::vector<int32_t> out(bk.size(), -1);
  for (int v = 1; v <= last; v++)
    for (int sl = 0; sl < (v <= n ? 1 : 3); sl++) {
      const int r = 3 * v + sl;
      out[(size_t)map(r)] = map(bk[(size_t)r]);
    }
  return out;
}
static void splits_of(int n, const std::vector<int32_t> &bk, int rec, std::vector<std::vector<int>> &out, std::vector<int> &tips)
{
  // tips behind record `rec` (looking away from bk[rec])
  if (is_tip(rec, n)) { tips.push_back(rec / 3); return; }
  std::vector<int> mine;
  splits_of(n, bk, bk[(size_t)nx(rec)], out, mine);
  splits_of(n, bk, bk[(size_t)nx(nx(rec))], out, mine);
  std::sort(mine.begin(), mine.end());
  if (mine.size() > 1 && (int)mine.size() < n - 1) out.push_back(mine);
  tips.insert(tips.end(), mine.begin(), mine.end());
}
static std::vector<std::vector<int>> bipartitions(int n, const std::vector<int32_t> &bk)
{
  std::vector<std::vector<int>> out;
  std::vector<int> tips;
  splits_of(n, bk, bk[3], out, tips);              // everything behind tip 1's neighbour: sides not containing tip 1
  std::sort(out.begin(), out.end());
  return out;
}
static int canon_check(uint64_t seed, int rounds)
{
  std::mt19937_64 g(seed);
  CanonScratch sc;
  std::string k0, k1;
  int same_seen = 0, differ_seen = 0;
  for (int r = 0; r < rounds; r++) {
    const int n = 4 + (int)(g() % 60);
    const std::vector<int32_t> a = random_tree(n, g);
    canonical_topology(n, a, k0, sc);
    const std::vector<int32_t> b = relabel(n, a, g);
    canonical_topology(n, b, k1, sc);
    if (k0 != k1 || bipartitions(n, a) != bipartitions(n, b)) { std::fprintf(stderr, "round %d: a relabelled tree got another key\n", r); return 1; }
    same_seen++;
    const std::vector<int32_t> c = random_tree(n, g);
    canonical_topology(n, c, k1, sc);
    const bool same_topology = bipartitions(n, a) == bipartitions(n, c);
    if ((k0 == k1) != same_topology) { std::fprintf(stderr, "round %d: keys %s, topologies %s\n", r, k0 == k1 ? "equal" : "differ", same_topology ? "equal" : "differ"); return 1; }
    differ_seen += !same_topology;
  }
  std::printf("canonical form: %d relabelled pairs equal, %d different topologies told apart\n", same_seen, differ_seen);
  return 0;
}

int main(int argc, char **argv)
{
  if (argc >= 4 && !std::strcmp(argv[1], "canon")) return canon_check(std::strtoull(argv[2], nullptr, 10), std::atoi(argv[3]));
  if (argc >= 3 && !std::strcmp(argv[1], "replay")) return replay_file(argv[2]);
  if (argc >= 4 && !std::strcmp(argv[1], "random")) return random_streams(std::strtoull(argv[2], nullptr, 10), std::atoi(argv[3]));
  std::fprintf(stderr, "usage: ufb_books_test replay <file> | random <seed> <batches> | canon <seed> <rounds>\n");
  return 2;
}
