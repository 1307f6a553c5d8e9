// climb.hip -- k_climb: the SPR hill climb of pllOptimizeSprParsimony (reference sprparsimony.cpp:3295-3316) as ONE
// persistent gfx950 kernel per sweep segment (interface and rationale: climb.hpp).
//
// Lane layout ("quad" tiles): a wavefront covers 16 * VW words of every state row; lane = 4 * w + g holds, for word
// group w, the KS states [g * KS, (g + 1) * KS) -- DNA: one state per lane, the four lanes of a DPP quad are the four
// states of a word; protein: five states per lane.  The only cross-state step of Fitch's rule, any = OR_k(a_k & b_k),
// is two quad_perm DPP ORs; everything else is per lane.  With one word per lane group a 1000 x 50 000 alignment is cut
// into 98 tiles, i.e. a climb's dependent chain of vector operations is spread over 98 CUs instead of 25.
//
// One step of the loop (all workgroups in lockstep, no communication except (5)):
//   (1) enumerate: for the next B prune nodes, both sides (rearrangeParsimony, :2259-2376): candidate counts per part
//       (gap end x first-level child) from a lane-parallel walk of the radius-6 neighbourhood in the LDS topology, and
//       the vectors the scans will read; the stale ones among them are claimed for the refresh;
//   (2) closure: the stale inputs of claimed vectors, transitively (lanes follow chains, a shared worklist for forks);
//   (3) refresh: dataflow over the claimed vectors -- a wave runs a chain with the running result in registers
//       (newviewParsimonyIterativeFast, :554-878), joins of two stale inputs are handed over through LDS counters;
//   (4) scan: one wave per part walks the DFS of addTraverseParsimony (:2208-2218) with the up-vector in registers and
//       leaves this tile's share of every candidate's length in LDS, in the reference's order;
//   (5) exchange: lengths summed over tiles (64-bit atomic add: low 40 bits value, high bits arrivals; polled);
//   (6) decide: testInsertParsimony's bookkeeping (:2168-2176) and the sweep's accept rule (:3306-3311) with the lcg64 tie
//       stream, the topology edit (removeNodeParsimony / restoreTreeRearrangeParsimony) and the invalidation it causes.
#include "climb.hpp"

#include "../../include/mpfitch.h"

namespace mpf {

namespace {

constexpr int kNW = 8;                       // waves per workgroup
constexpr int kThreads = kNW * 64;
constexpr int kMaxB = 8;                     // prune nodes per step
constexpr int kMaxUnits = 2 * kMaxB;         // (prune node, side)
constexpr int kMaxParts = 4 * kMaxUnits;     // (prune node, side, gap end, first-level child)
constexpr int kDepth = 6;                    // deepest radius
constexpr uint32_t kNone16 = 0xFFFFu;
// heap lanes (root 1, children 2h / 2h + 1) below the root's first / second child
constexpr unsigned long long kSub0 = (1ull << 2) | (3ull << 4) | (0xFull << 8) | (0xFFull << 16) | (0xFFFFull << 32);
constexpr unsigned long long kSub1 = (1ull << 3) | (3ull << 6) | (0xFull << 12) | (0xFFull << 24) | (0xFFFFull << 48);
constexpr unsigned long long kValMask = (1ull << 40) - 1ull;

typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t rfl(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

__device__ __forceinline__ uint32_t quad_or(uint32_t v)
{
  v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]
  v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
  return v;
}

__device__ __forceinline__ uint32_t wave_total(uint32_t v)
{
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, false);  // row_half_mirror
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, false);  // row_mirror
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);  // row_bcast:15
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);  // row_bcast:31
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// one atomic per wave whose result every lane needs: issued by ALL lanes (lane 0 adds v, the others 0) and read back from
// lane 0.  The obvious form -- `if (lane == 0) old = atomic(...); old = readfirstlane(old);` -- is not safe here: hipcc
// (ROCm 7.2) threads the lanes that skip the branch past it into the next loop iteration with their own constant, so that
// they reach the readfirstlane without lane 0 (seen in the ISA of the scan's task loop: an endless loop of lanes 1..63).
__device__ __forceinline__ uint32_t wave_fetch_add(uint32_t *p, uint32_t v, int lane)
{
  return rfl(__hip_atomic_fetch_add(p, lane == 0 ? v : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
}
__device__ __forceinline__ uint32_t wave_fetch_sub(uint32_t *p, uint32_t v, int lane)
{
  return rfl(__hip_atomic_fetch_sub(p, lane == 0 ? v : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
}

#define MPF_B3_ANDOR 0xEA   // (a & b) | c
#define MPF_B3_FITCH 0xD4   // c ? (a & b) : (a | b)
__device__ __forceinline__ uint32_t b3_andor(uint32_t a, uint32_t b, uint32_t c) { return (uint32_t)__builtin_amdgcn_bitop3_b32((int)a, (int)b, (int)c, MPF_B3_ANDOR); }
__device__ __forceinline__ uint32_t b3_fitch(uint32_t a, uint32_t b, uint32_t any) { return (uint32_t)__builtin_amdgcn_bitop3_b32((int)a, (int)b, (int)any, MPF_B3_FITCH); }

// next record of an inner node's ring, in vector ids (tips 0..n-1, inner record 3v+s -> n + 3(v-n-1) + s)
__device__ __forceinline__ uint32_t nxc(uint32_t c, uint32_t n)
{
  const uint32_t s = (c - n) % 3u;
  return s == 2u ? c - 2u : c + 1u;
}

template <int KS, int VW>
struct QT { uint32_t v[KS][VW]; };

template <int KS, int VW>
__device__ __forceinline__ void qload(QT<KS, VW> &t, __amdgpu_buffer_rsrc_t rsrc, const uint32_t (&voff)[KS], uint32_t soff)
{
#pragma unroll
  for (int k = 0; k < KS; k++) {
    if constexpr (VW == 1) {
      t.v[k][0] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff[k], soff, 0);
    } else if constexpr (VW == 2) {
      const v2u x = __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff[k], soff, 0);
      t.v[k][0] = x[0]; t.v[k][1] = x[1];
    } else {
      const v4u x = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[k], soff, 0);
      t.v[k][0] = x[0]; t.v[k][1] = x[1]; t.v[k][2] = x[2]; t.v[k][3] = x[3];
    }
  }
}

template <int KS, int VW>
__device__ __forceinline__ void qstore(const QT<KS, VW> &t, __amdgpu_buffer_rsrc_t rsrc, const uint32_t (&voff)[KS], uint32_t soff)
{
#pragma unroll
  for (int k = 0; k < KS; k++) {
    if constexpr (VW == 1) {
      __builtin_amdgcn_raw_buffer_store_b32(t.v[k][0], rsrc, voff[k], soff, 0);
    } else if constexpr (VW == 2) {
      v2u x; x[0] = t.v[k][0]; x[1] = t.v[k][1];
      __builtin_amdgcn_raw_buffer_store_b64(x, rsrc, voff[k], soff, 0);
    } else {
      v4u x; x[0] = t.v[k][0]; x[1] = t.v[k][1]; x[2] = t.v[k][2]; x[3] = t.v[k][3];
      __builtin_amdgcn_raw_buffer_store_b128(x, rsrc, voff[k], soff, 0);
    }
  }
}

// c = fitch(a, b) (reference sprparsimony.cpp:737-776); returns the sites of this lane group's words with an empty
// intersection -- the same number in all four lanes of a quad
template <int KS, int VW>
__device__ __forceinline__ uint32_t q_fitch(QT<KS, VW> &c, const QT<KS, VW> &a, const QT<KS, VW> &b)
{
  uint32_t cost = 0;
#pragma unroll
  for (int j = 0; j < VW; j++) {
    uint32_t t = a.v[0][j] & b.v[0][j];
#pragma unroll
    for (int k = 1; k < KS; k++) t = b3_andor(a.v[k][j], b.v[k][j], t);
    const uint32_t any = quad_or(t);
#pragma unroll
    for (int k = 0; k < KS; k++) c.v[k][j] = b3_fitch(a.v[k][j], b.v[k][j], any);
    cost += (uint32_t)__builtin_popcount(~any);
  }
  return cost;
}

// sites where the subtree vector s has no state in common with fitch(u, d) (evaluateParsimonyIterativeFast, :1108-1124,
// on the node the insertion would create)
template <int KS, int VW>
__device__ __forceinline__ uint32_t q_join(const QT<KS, VW> &u, const QT<KS, VW> &d, const QT<KS, VW> &s)
{
  uint32_t cost = 0;
#pragma unroll
  for (int j = 0; j < VW; j++) {
    uint32_t t = u.v[0][j] & d.v[0][j];
#pragma unroll
    for (int k = 1; k < KS; k++) t = b3_andor(u.v[k][j], d.v[k][j], t);
    const uint32_t any = quad_or(t);
    uint32_t hit = b3_fitch(u.v[0][j], d.v[0][j], any) & s.v[0][j];
#pragma unroll
    for (int k = 1; k < KS; k++) hit = b3_andor(b3_fitch(u.v[k][j], d.v[k][j], any), s.v[k][j], hit);
    hit = quad_or(hit);
    cost += (uint32_t)__builtin_popcount(~hit);
  }
  return cost;
}

struct Unit { uint16_t x, s, xa, xb, mt, p; };

// the control block every workgroup keeps in LDS (identical everywhere: it is computed from exchanged sums only)
struct Sh {
  uint32_t pos, B, Beff, epoch, exit_reason, ncand, since_move, steps, xgen, n_moves, err, consumed;
  uint32_t last_ncand[3];
  uint32_t best, randomMP, iter_hits;
  int32_t ins, rem;
  unsigned long long rng, hits, n_tests, n_ops, draws, n_nodes;
  uint32_t wtail, rtail, rhead, ndone, nops, task, ok, trace_n;
  Unit unit[kMaxUnits];
  uint32_t pcnt[kMaxParts], poff[kMaxParts];
  uint32_t pn_off[kMaxB], pn_cnt[kMaxB], pn_np[kMaxB], pn_p[kMaxB];
};

template <int KS, int VW>
struct Kx {
  uint16_t *bk;      // back links (vector ids)
  uint8_t *valid;    // the vector in HBM is the Fitch vector of the current tree
  uint32_t *cl;      // claim word: epoch << 8 | stale inputs not yet recomputed
  uint16_t *W, *R;   // worklist of the closure / the invalidation, ready list of the refresh
  uint32_t *cost;    // per candidate: this tile's share, after the exchange the length
  uint16_t *cq;      // per candidate: the insertion branch
  uint32_t *pend;    // [wave][depth][KS * VW][64] up-vectors of second children waiting for their turn
  uint2 *frames;     // [wave][8]
  uint32_t *sct;     // this tile's subtree scores (global)
  uint32_t n, ns, SW4;
  int lane, wave;
  __amdgpu_buffer_rsrc_t rsrc;
  uint32_t voff[KS];
  bool cnt_lane;     // this lane's popcounts count (first lane of a quad, inside the row)
  bool st_lane;      // this lane's words exist
};

template <int KS, int VW>
__device__ __forceinline__ void ld(const Kx<KS, VW> &K, QT<KS, VW> &t, uint32_t cid) { qload<KS, VW>(t, K.rsrc, K.voff, cid * K.SW4); }

template <int KS, int VW>
__device__ __forceinline__ uint32_t ld_sct(const Kx<KS, VW> &K, uint32_t cid)
{
  return cid < K.n ? 0u : __hip_atomic_load(K.sct + cid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// a vector the scans of this step read: if it is stale, claim it for the refresh (once per step)
template <int KS, int VW>
__device__ __forceinline__ void require(const Kx<KS, VW> &K, Sh &sh, uint32_t c, uint32_t epoch)
{
  if (c >= K.n && !K.valid[c]) {
    const uint32_t old = __hip_atomic_fetch_max(&K.cl[c], epoch << 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if ((old >> 8) != epoch) {
      const uint32_t slot = __hip_atomic_fetch_add(&sh.wtail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      K.W[slot] = (uint16_t)c;
    }
  }
}

// ---- (1) one side of one prune node: which scans exist (rearrangeParsimony's tests, :2304-2310, :2330-2347), how many
// insertion tests each part holds, which vectors they read
template <int KS, int VW>
__device__ __forceinline__ void enum_unit(const Kx<KS, VW> &K, Sh &sh, const ClimbParams &P, uint32_t u)
{
  const uint32_t n = K.n, epoch = sh.epoch;
  const int lane = K.lane;
  const uint32_t j = u >> 1, side = u & 1u;
  const uint32_t p = rfl((uint32_t)P.order[sh.pos - 1u + j]);
  const uint32_t q = rfl((uint32_t)K.bk[p]);
  const uint32_t x = side ? q : p, s = side ? p : q;
  const uint32_t mt = side ? 2u : 1u;                   // the q side does not test the first level (mintrav2 = 2)
  uint32_t xa = 0, xb = 0;
  bool app = false;
  if (x >= n) {
    xa = rfl((uint32_t)K.bk[nxc(x, n)]);
    xb = rfl((uint32_t)K.bk[nxc(nxc(x, n), n)]);
    if (!side) {
      app = xa >= n || xb >= n;
    } else {
      bool da = false, db = false;
      if (xa >= n) da = rfl((uint32_t)K.bk[nxc(xa, n)]) >= n || rfl((uint32_t)K.bk[nxc(nxc(xa, n), n)]) >= n;
      if (xb >= n) db = rfl((uint32_t)K.bk[nxc(xb, n)]) >= n || rfl((uint32_t)K.bk[nxc(nxc(xb, n), n)]) >= n;
      app = da || db;
    }
  }
  uint32_t cnt[4] = {0u, 0u, 0u, 0u};
  if (app) {
    const int h = lane;                                  // heap index of an expansion; lane 0 idles
    const int d = h ? 31 - __builtin_clz((unsigned)h) : -1;
    for (uint32_t e = 0; e < 2u; e++) {
      const uint32_t a = e ? xb : xa, other = e ? xa : xb;
      if (a < n) continue;                               // a tip has nothing behind it
      uint32_t node = a;
      bool ok = h >= 1;
      for (int l = 0; l < kDepth - 1; l++) {
        if (l < d && ok) {
          const uint32_t bit = ((uint32_t)h >> (d - 1 - l)) & 1u;
          const uint32_t r1 = nxc(node, n);
          node = K.bk[bit ? nxc(r1, n) : r1];
          ok = node >= n;
        }
      }
      const bool ex = ok && d < (int)P.maxtrav;
      uint32_t c1 = 0, c2 = 0;
      if (ex) {
        const uint32_t r1 = nxc(node, n);
        c1 = K.bk[r1];
        c2 = K.bk[nxc(r1, n)];
      }
      const bool tested = ex && (uint32_t)(d + 1) >= mt;
      const unsigned long long Tm = __ballot((int)tested);
      const uint32_t root = (uint32_t)((Tm >> 1) & 1ull);
      const uint32_t c0n = root + 2u * (uint32_t)__builtin_popcountll(Tm & kSub0);
      const uint32_t c1n = root + 2u * (uint32_t)__builtin_popcountll(Tm & kSub1);
      cnt[2 * e] = c0n;
      cnt[2 * e + 1] = c1n;
      const bool walk0 = c0n > 0, walk1 = c1n > 0;
      const bool need = ex && (h == 1 ? (walk0 || walk1) : (((kSub0 >> h) & 1ull) ? walk0 : walk1));
      if (need) {
        require<KS, VW>(K, sh, c1, epoch);
        require<KS, VW>(K, sh, c2, epoch);
      }
      if (lane == 0 && (walk0 || walk1)) require<KS, VW>(K, sh, other, epoch);
    }
    if (lane == 0 && (cnt[0] | cnt[1] | cnt[2] | cnt[3])) {
      require<KS, VW>(K, sh, s, epoch);                  // the pruned subtree, and both ends of the prune branch for the base length
      require<KS, VW>(K, sh, x, epoch);
    }
  }
  if (lane == 0) {
    Unit un;
    un.x = (uint16_t)x; un.s = (uint16_t)s; un.xa = (uint16_t)xa; un.xb = (uint16_t)xb; un.mt = (uint16_t)mt; un.p = (uint16_t)p;
    sh.unit[u] = un;
#pragma unroll
    for (int i = 0; i < 4; i++) sh.pcnt[4u * u + (uint32_t)i] = cnt[i];
  }
}

// ---- (2) wave 0: output layout of the step, then the closure of stale inputs
template <int KS, int VW>
__device__ __forceinline__ void plan_and_discover(const Kx<KS, VW> &K, Sh &sh)
{
  const int lane = K.lane;
  const uint32_t n = K.n, epoch = sh.epoch, B = sh.B;
  // candidates are laid out part after part = the reference's order (p side: first gap end, its first child's subtree, ...)
  const uint32_t nparts = 8u * B;
  const uint32_t c = (uint32_t)lane < nparts ? sh.pcnt[lane] : 0u;
  uint32_t incl = c;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t t = (uint32_t)__shfl_up((int)incl, o, 64);
    if (lane >= o) incl += t;
  }
  // speculation is cut where the step's candidate buffer ends (one prune node never exceeds it)
  const unsigned long long fits = __ballot((lane & 7) == 7 && (uint32_t)lane < nparts && incl <= kClimbCap);
  const uint32_t Beff = (uint32_t)__builtin_popcountll(fits);
  const uint32_t tot_eff = (uint32_t)__shfl((int)incl, (int)(8u * Beff) - 1, 64);
  const uint32_t excl = incl - c;
  const uint32_t e4 = (uint32_t)__shfl((int)incl, (lane & ~7) + 3, 64), e8 = (uint32_t)__shfl((int)incl, (lane & ~7) + 7, 64);
  if ((uint32_t)lane < nparts) {
    const bool live = (uint32_t)(lane >> 3) < Beff;
    sh.poff[lane] = excl;
    if (!live) sh.pcnt[lane] = 0u;
    if ((lane & 7) == 0 && live) {
      const int j = lane >> 3;
      sh.pn_off[j] = excl;
      sh.pn_cnt[j] = e8 - excl;
      sh.pn_np[j] = e4 - excl;
      sh.pn_p[j] = sh.unit[2 * j].p;
    }
  }
  if (lane == 0) { sh.Beff = Beff; sh.ncand = Beff ? tot_eff : 0u; }
  // closure: every lane follows one chain of stale inputs; forks go to the shared worklist
  uint32_t head = 0, tail = sh.wtail, nops = 0;
  bool have = false;
  uint32_t item = 0;
  for (uint32_t round = 0;; round++) {
    if (round > K.ns) { if (lane == 0) sh.err = 3u; break; }
    {
      const unsigned long long need = __ballot((int)!have);
      const uint32_t rank = (uint32_t)__builtin_popcountll(need & ((1ull << lane) - 1ull));
      const uint32_t avail = tail - head;
      if (!have && rank < avail) { item = K.W[head + rank]; have = true; }
      const uint32_t want = (uint32_t)__builtin_popcountll(need);
      head += want < avail ? want : avail;
    }
    const unsigned long long act = __ballot((int)have);
    if (!act) break;
    nops += (uint32_t)__builtin_popcountll(act);
    bool push = false;
    uint32_t pv = 0;
    if (have) {
      const uint32_t r = item;
      const uint32_t r1 = nxc(r, n);
      const uint32_t a = K.bk[r1], b = K.bk[nxc(r1, n)];
      const bool sa = a >= n && !K.valid[a], sb = b >= n && !K.valid[b];
      bool wa = false, wb = false;
      if (sa) wa = (__hip_atomic_fetch_max(&K.cl[a], epoch << 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >> 8) != epoch;
      if (sb) wb = (__hip_atomic_fetch_max(&K.cl[b], epoch << 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >> 8) != epoch;
      const uint32_t ns = (sa ? 1u : 0u) + (sb ? 1u : 0u);
      if (ns) {
        K.cl[r] = (epoch << 8) | ns;                     // (r is claimed already: other lanes' fetch_max leave the word as it is)
      } else {
        const uint32_t slot = __hip_atomic_fetch_add(&sh.rtail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        K.R[slot] = (uint16_t)r;                         // both inputs valid: a chain starts here
      }
      if (wa) { item = a; if (wb) { push = true; pv = b; } }
      else if (wb) item = b;
      else have = false;
    }
    const unsigned long long pm = __ballot((int)push);
    if (push) K.W[tail + (uint32_t)__builtin_popcountll(pm & ((1ull << lane) - 1ull))] = (uint16_t)pv;
    tail += (uint32_t)__builtin_popcountll(pm);
  }
  if (lane == 0) sh.nops = nops;
}

// ---- (3) dataflow refresh of the claimed vectors
template <int KS, int VW>
__device__ __forceinline__ void refresh(const Kx<KS, VW> &K, Sh &sh)
{
  const uint32_t n = K.n, epoch = sh.epoch, nops = sh.nops;
  const int lane = K.lane;
  if (nops == 0) return;
  QT<KS, VW> c, ta, tb;
  for (;;) {
    const uint32_t idx = wave_fetch_add(&sh.rhead, 1u, lane);
    uint32_t r = kNone16, spins = 0;
    for (;;) {
      r = *(volatile uint16_t *)&K.R[idx];
      if (r != kNone16) break;
      if (*(volatile uint32_t *)&sh.ndone >= nops) break;
      if (++spins > (1u << 22)) { sh.err = 2u; break; }   // (bounded like every wait in this kernel)
      __builtin_amdgcn_s_sleep(1);
    }
    r = rfl(r);
    if (r == kNone16) break;
    if (lane == 0) K.R[idx] = (uint16_t)kNone16;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    uint32_t prev = kNone16, sc_prev = 0;
    for (uint32_t link = 0;; link++) {
      if (link > K.ns) { if (lane == 0) sh.err = 5u; break; }
      const uint32_t r1 = nxc(r, n);
      const uint32_t a = rfl((uint32_t)K.bk[r1]), b = rfl((uint32_t)K.bk[nxc(r1, n)]);
      uint32_t sa, sb;
      if (prev == a) {
        ta = c; sa = sc_prev;
        ld<KS, VW>(K, tb, b); sb = ld_sct<KS, VW>(K, b);
      } else if (prev == b) {
        tb = c; sb = sc_prev;
        ld<KS, VW>(K, ta, a); sa = ld_sct<KS, VW>(K, a);
      } else {
        ld<KS, VW>(K, ta, a); ld<KS, VW>(K, tb, b);
        sa = ld_sct<KS, VW>(K, a); sb = ld_sct<KS, VW>(K, b);
      }
      const uint32_t cost = q_fitch<KS, VW>(c, ta, tb);
      if (K.st_lane) qstore<KS, VW>(c, K.rsrc, K.voff, r * K.SW4);
      const uint32_t sc = wave_total(K.cnt_lane ? cost : 0u) + rfl(sa) + rfl(sb);   // tr->parsimonyScore[p], :874 (this tile's share)
      if (lane == 0) {
        __hip_atomic_store(K.sct + r, sc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        K.valid[r] = 1;
      }
      // the vectors that take this one as an input: the two other records of the node behind it
      const uint32_t w = rfl((uint32_t)K.bk[r]);
      uint32_t nr = 0, rdy0 = 0, rdy1 = 0;
      if (w >= n) {
        uint32_t cand = w;
#pragma unroll
        for (int i = 0; i < 2; i++) {
          cand = nxc(cand, n);
          const uint32_t v = rfl(*(volatile uint32_t *)&K.cl[cand]);
          const uint32_t vc = rfl((uint32_t) * (volatile uint8_t *)&K.valid[cand]);
          if ((v >> 8) == epoch && !vc) {
            bool ready;
            if ((v & 0xFFu) == 1u) {
              ready = true;                              // its other input is valid (or was finished before): go on in registers
            } else {
              // a join of two stale inputs: whoever arrives second goes on, reading the first one's result from memory
              __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
              const uint32_t old = wave_fetch_sub(&K.cl[cand], 1u, lane);
              ready = (old & 0xFFu) == 1u;
              if (ready) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            }
            if (ready) { if (nr == 0) rdy0 = cand; else rdy1 = cand; nr++; }
          }
        }
      }
      if (nr == 2) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) {
          const uint32_t slot = __hip_atomic_fetch_add(&sh.rtail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          K.R[slot] = (uint16_t)rdy1;
        }
      }
      if (lane == 0) __hip_atomic_fetch_add(&sh.ndone, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (nr == 0) break;
      prev = r; sc_prev = sc; r = rdy0;
    }
  }
}

// ---- (4) one part of a scan: addTraverseParsimony + testInsertParsimony (:2208-2218, :2106-2160) below one first-level
// child of one gap end, both children of a node expanded together (one vector read per insertion test)
template <int KS, int VW>
__device__ __forceinline__ void scan_part(const Kx<KS, VW> &K, Sh &sh, uint32_t pi, uint32_t maxtrav)
{
  const uint32_t n = K.n;
  const int lane = K.lane;
  const Unit un = sh.unit[pi >> 2];
  const uint32_t e = (pi >> 1) & 1u, child_mask = 1u << (pi & 1u);
  const uint32_t a = rfl((uint32_t)(e ? un.xb : un.xa)), other = rfl((uint32_t)(e ? un.xa : un.xb));
  const uint32_t mt = rfl((uint32_t)un.mt);
  const uint32_t out_base = rfl(sh.poff[pi]);
  uint2 *stk = K.frames + K.wave * 8;
  uint32_t *pend = K.pend + (size_t)K.wave * (kDepth * KS * VW * 64);
  QT<KS, VW> sv, par, u1, u2, d1, d2;
  ld<KS, VW>(K, sv, rfl((uint32_t)un.s));
  ld<KS, VW>(K, par, other);
  uint32_t k = 0;
  int sp = 0;
  uint32_t node = a, d = 0;
  auto emit = [&](uint32_t cst, uint32_t cid) {
    if (lane == 0) { K.cost[out_base + k] = cst; K.cq[out_base + k] = (uint16_t)cid; }
    k++;
  };
  for (uint32_t it = 0;; it++) {
    if (it > 256u) { if (lane == 0) sh.err = 6u; break; }
    const uint32_t r1 = nxc(node, n);
    const uint32_t c1 = rfl((uint32_t)K.bk[r1]), c2 = rfl((uint32_t)K.bk[nxc(r1, n)]);
    ld<KS, VW>(K, d1, c1);
    ld<KS, VW>(K, d2, c2);
    const uint32_t dd = d + 1u;
    const bool test = dd >= mt, deeper = dd < maxtrav;
    q_fitch<KS, VW>(u1, par, d2);
    q_fitch<KS, VW>(u2, par, d1);
    const bool own1 = dd > 1u || (child_mask & 1u), own2 = dd > 1u || (child_mask & 2u);
    uint32_t tot = 0;
    if (test) {
      uint32_t cst = q_join<KS, VW>(u1, d1, sv) | (q_join<KS, VW>(u2, d2, sv) << 16);
      cst = K.cnt_lane ? cst : 0u;
      tot = wave_total(cst);
    }
    if (own2 && deeper && c2 >= n) {
#pragma unroll
      for (int kk = 0; kk < KS; kk++)
#pragma unroll
        for (int jj = 0; jj < VW; jj++) pend[((dd - 1u) * (KS * VW) + (uint32_t)(kk * VW + jj)) * 64u + (uint32_t)lane] = u2.v[kk][jj];
    }
    if (own2) { stk[sp] = make_uint2(c2 | (dd << 24), tot >> 16); sp++; }
    if (test && own1) emit(tot & 0xFFFFu, c1);
    if (own1 && deeper && c1 >= n) { par = u1; node = c1; d = dd; continue; }
    bool more = false;
    while (sp > 0) {
      sp--;
      const uint2 fr = stk[sp];
      const uint32_t fx = rfl(fr.x);
      const uint32_t q = fx & 0xFFFFFFu, dq = fx >> 24;
      if (dq >= mt) emit(rfl(fr.y), q);
      if (dq < maxtrav && q >= n) {
#pragma unroll
        for (int kk = 0; kk < KS; kk++)
#pragma unroll
          for (int jj = 0; jj < VW; jj++) par.v[kk][jj] = pend[((dq - 1u) * (KS * VW) + (uint32_t)(kk * VW + jj)) * 64u + (uint32_t)lane];
        node = q; d = dq; more = true;
        break;
      }
    }
    if (!more) break;
  }
  if (lane == 0 && k != sh.pcnt[pi]) sh.err = 100u + pi;   // the walk and the enumeration disagree: never on a consistent tree
}

// debugging aid (option "climb_trace"): workgroup 0 leaves where it is in pinned host memory, so that a launch that does not
// come back can be diagnosed from the host
__device__ __forceinline__ void beat(const ClimbParams &P, uint32_t tile, int tid, uint32_t slot, uint32_t v)
{
  if (P.beat && tile == 0 && tid == 0) __hip_atomic_store(P.beat + slot, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__device__ __forceinline__ double tie_draw(unsigned long long &st)
{
  st = st * 0x27bb2ee687b0b0fdULL + 3037000493ULL;       // sprng/lcg64.c:220
  return (double)st * 5.4210108624275222e-20;            // :268
}

// ---- (6) wave 0: the reference's bookkeeping over the step's prune nodes, the move, its invalidation
template <int KS, int VW>
__device__ __forceinline__ void decide(const Kx<KS, VW> &K, Sh &sh, const ClimbParams &P, uint32_t tile)
{
  const int lane = K.lane;
  const uint32_t n = K.n;
  uint32_t best = sh.best, randomMP = sh.randomMP, iter_hits = sh.iter_hits;
  unsigned long long rng = sh.rng, hits = sh.hits, draws = sh.draws, tests = 0;
  int32_t ins = sh.ins, rem = sh.rem;
  const bool rnd = P.tie_mode == (uint32_t)MPF_TIE_RANDOM;
  const uint32_t Beff = sh.Beff;
  bool moved = false;
  uint32_t j = 0;
  for (; j < Beff && !moved; j++) {
    const uint32_t off = sh.pn_off[j], nt = sh.pn_cnt[j], np = sh.pn_np[j], pcid = sh.pn_p[j];
    tests += nt;
    if (rnd) { ins = rem = -1; hits = 1; }
    int32_t sel = -1;
    uint32_t mn = 0xFFFFFFFFu;
    for (uint32_t base = 0; base < nt; base += 64u) {
      const uint32_t ci = base + (uint32_t)lane;
      const uint32_t m = ci < nt ? K.cost[off + ci] : 0xFFFFFFFFu;
      if (P.trace) {
        uint32_t t = m;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const uint32_t x = (uint32_t)__shfl_xor((int)t, o, 64); t = x < t ? x : t; }
        mn = t < mn ? t : mn;
      }
      unsigned long long mask = __ballot((int)(m <= best));
      while (mask) {
        const int l = __builtin_ctzll(mask);
        mask &= mask - 1ull;
        const uint32_t v = (uint32_t)__builtin_amdgcn_readlane((int)m, l);
        if (v > best) continue;                          // best has fallen meanwhile
        const int32_t cidx = (int32_t)(base + (uint32_t)l);
        if (rnd) {
          if (v < best) hits = 1; else hits++;
          bool take = v < best;
          if (!take) { draws++; take = tie_draw(rng) <= 1.0 / (double)hits; }
          if (take) { best = v; sel = cidx; }
        } else if (v < best) {
          best = v; sel = cidx;
        }
      }
    }
    if (sel >= 0) {
      ins = (int32_t)K.cq[off + (uint32_t)sel];
      rem = (uint32_t)sel < np ? (int32_t)pcid : (int32_t)K.bk[pcid];
    }
    bool accept;
    if (rnd) {
      if (best == randomMP) iter_hits++;
      if (best < randomMP) iter_hits = 1;
      accept = best < randomMP;
      if (!accept && best == randomMP) { draws++; accept = tie_draw(rng) <= 1.0 / (double)iter_hits; }
      accept = accept && rem >= 0 && ins >= 0;
    } else {
      accept = best < randomMP;
    }
    if (P.trace && tile == 0 && lane == 0 && sh.trace_n < P.trace_cap) {
      uint32_t *t = P.trace + 8u * sh.trace_n;
      t[0] = sh.pos + j; t[1] = pcid; t[2] = nt; t[3] = np; t[4] = mn; t[5] = best; t[6] = (uint32_t)sel; t[7] = accept ? 1u : 0u;
      sh.trace_n++;
    }
    if (accept) {
      moved = true;
      randomMP = best;
      if (tile == 0 && lane == 0) {
        uint32_t *mv = P.moves + 3u * sh.n_moves;
        mv[0] = (uint32_t)rem; mv[1] = (uint32_t)ins; mv[2] = best;
      }
    }
  }
  const uint32_t consumed = j;
  const uint32_t n_moves = sh.n_moves + (moved ? 1u : 0u);
  if (moved) {
    // removeNodeParsimony + restoreTreeRearrangeParsimony (:2245-2257, :2379-2384), link by link as the host mirror does
    const uint32_t p = (uint32_t)rem, q = (uint32_t)ins;
    const uint32_t p1 = nxc(p, n), p2 = nxc(p1, n);
    const uint32_t a = rfl((uint32_t)K.bk[p1]), b = rfl((uint32_t)K.bk[p2]);
    const uint32_t r = rfl((uint32_t)K.bk[q]);
    if (lane == 0) {
      K.bk[a] = (uint16_t)b; K.bk[b] = (uint16_t)a;
      K.bk[p1] = (uint16_t)q; K.bk[q] = (uint16_t)p1;
      K.bk[p2] = (uint16_t)r; K.bk[r] = (uint16_t)p2;
    }
    // every vector whose subtree contains an edited node is stale: the three of each edited node and, walking outwards, the
    // two outward-looking ones of every node reached, as far as they were valid
    const uint32_t einv = sh.epoch + 1u;
    uint32_t five = a;
    five = lane / 3 == 1 ? b : five;
    five = lane / 3 == 2 ? p : five;
    five = lane / 3 == 3 ? q : five;
    five = lane / 3 == 4 ? r : five;
    bool have = false;
    uint32_t item = 0;
    if (lane < 15 && five >= n) {
      const uint32_t rec = five - (five - n) % 3u + (uint32_t)(lane % 3);
      K.valid[rec] = 0;
      const uint32_t w = K.bk[rec];
      if (w >= n) { have = true; item = w; }
    }
    uint32_t head = 0, tail = 0;
    for (uint32_t round = 0;; round++) {
      if (round > K.ns) { if (lane == 0) sh.err = 4u; break; }
      {
        const unsigned long long need = __ballot((int)!have);
        const uint32_t rank = (uint32_t)__builtin_popcountll(need & ((1ull << lane) - 1ull));
        const uint32_t avail = tail - head;
        if (!have && rank < avail) { item = K.W[head + rank]; have = true; }
        const uint32_t want = (uint32_t)__builtin_popcountll(need);
        head += want < avail ? want : avail;
      }
      if (!__ballot((int)have)) break;
      bool push = false;
      uint32_t pv = 0;
      if (have) {
        const uint32_t o1 = nxc(item, n), o2 = nxc(o1, n);
        const bool w1 = (__hip_atomic_fetch_max(&K.cl[o1], einv << 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >> 8) != einv;
        const bool w2 = (__hip_atomic_fetch_max(&K.cl[o2], einv << 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >> 8) != einv;
        const bool v1 = w1 && K.valid[o1], v2 = w2 && K.valid[o2];
        const uint32_t u1 = K.bk[o1], u2 = K.bk[o2];
        if (v1) K.valid[o1] = 0;
        if (v2) K.valid[o2] = 0;
        const bool g1 = v1 && u1 >= n, g2 = v2 && u2 >= n;
        if (g1) { item = u1; if (g2) { push = true; pv = u2; } }
        else if (g2) item = u2;
        else have = false;
      }
      const unsigned long long pm = __ballot((int)push);
      if (push) K.W[tail + (uint32_t)__builtin_popcountll(pm & ((1ull << lane) - 1ull))] = (uint16_t)pv;
      tail += (uint32_t)__builtin_popcountll(pm);
    }
  }
  if (lane == 0) {
    sh.best = best; sh.randomMP = randomMP; sh.iter_hits = iter_hits;
    sh.rng = rng; sh.hits = hits; sh.draws = draws; sh.ins = ins; sh.rem = rem;
    sh.n_tests += tests; sh.n_nodes += consumed;
    sh.n_moves = n_moves;
    sh.consumed = consumed;
    const uint32_t pos = sh.pos + consumed;
    sh.pos = pos;
    sh.steps++;
    sh.epoch += 2u;
    const uint32_t since = moved ? 0u : sh.since_move + consumed;
    sh.since_move = since;
    // a batch is wasted behind the first accepted move; after a step without one the next looks twice as far ahead
    uint32_t B = moved ? P.batch_min : sh.B * 2u;
    B = B > P.batch_max ? P.batch_max : B;
    B = B < 1u ? 1u : B;
    sh.B = B;
    uint32_t reason = CLIMB_RUNNING;
    if (pos > P.total) reason = CLIMB_SWEEP_END;
    else if (n_moves >= P.max_moves) reason = CLIMB_MOVES_FULL;
    else if (P.idle_limit && since >= P.idle_limit) reason = CLIMB_IDLE;
    if (consumed == 0u || sh.steps > 4u * P.total + 16u) sh.err = sh.err ? sh.err : 7u;
    if (sh.err) reason = CLIMB_ERROR;
    sh.exit_reason = reason;
  }
}

template <int KS, int VW>
__global__ __launch_bounds__(kThreads) void k_climb(ClimbParams P)
{
  extern __shared__ __align__(16) unsigned char smem[];
  const int tid = (int)threadIdx.x, lane = tid & 63;
  const int wave = (int)rfl((uint32_t)(tid >> 6));
  const uint32_t tile = blockIdx.x, n = P.n, ns = P.nslots, T = P.tiles;
  // ---- carve the workgroup's LDS
  Sh &sh = *reinterpret_cast<Sh *>(smem);
  size_t at = (sizeof(Sh) + 15) & ~(size_t)15;
  Kx<KS, VW> K;
  K.cl = reinterpret_cast<uint32_t *>(smem + at); at += (((size_t)ns * 4) + 15) & ~(size_t)15;
  K.cost = reinterpret_cast<uint32_t *>(smem + at); at += (size_t)kClimbCap * 4;
  K.pend = reinterpret_cast<uint32_t *>(smem + at); at += (size_t)kNW * kDepth * KS * VW * 64 * 4;
  K.frames = reinterpret_cast<uint2 *>(smem + at); at += (size_t)kNW * 8 * sizeof(uint2);
  K.bk = reinterpret_cast<uint16_t *>(smem + at); at += (((size_t)ns * 2) + 15) & ~(size_t)15;
  K.W = reinterpret_cast<uint16_t *>(smem + at); at += (((size_t)ns * 2) + 15) & ~(size_t)15;
  K.R = reinterpret_cast<uint16_t *>(smem + at); at += (((size_t)(ns + 16) * 2) + 15) & ~(size_t)15;
  K.cq = reinterpret_cast<uint16_t *>(smem + at); at += (size_t)kClimbCap * 2;
  K.valid = reinterpret_cast<uint8_t *>(smem + at);
  K.n = n; K.ns = ns; K.lane = lane; K.wave = wave;
  K.SW4 = (uint32_t)(4 * KS) * P.Wp * 4u;
  K.sct = P.sct + (size_t)tile * ns;
  K.rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)P.vec, 0, 0x7FFFFFFF, 0x00020000);
  {
    const uint32_t w = (uint32_t)lane >> 2, g = (uint32_t)lane & 3u;
    uint32_t word0 = (tile * 16u + w) * (uint32_t)VW;
    K.st_lane = word0 < P.Wp;
    if (!K.st_lane) word0 = P.Wp - (uint32_t)VW;         // lanes past the row end load real data and contribute nothing
    K.cnt_lane = K.st_lane && g == 0u;
#pragma unroll
    for (int k = 0; k < KS; k++) K.voff[k] = ((g * (uint32_t)KS + (uint32_t)k) * P.Wp + word0) * 4u;
  }
  // ---- the launch's state: topology, all inner vectors stale (the kernel keeps its own per-tile subtree scores)
  for (uint32_t i = (uint32_t)tid; i < ns; i += kThreads) {
    K.bk[i] = P.bk[i];
    K.valid[i] = i < n ? 1 : 0;
    K.cl[i] = 0u;
  }
  for (uint32_t i = (uint32_t)tid; i < ns + 16u; i += kThreads) K.R[i] = (uint16_t)kNone16;
  if (tid == 0) {
    const ClimbHeader h = *P.hdr;
    sh.pos = h.pos; sh.B = h.batch ? h.batch : P.batch_min; sh.epoch = 1u; sh.exit_reason = CLIMB_RUNNING;
    sh.since_move = h.since_move; sh.steps = 0; sh.xgen = 0; sh.n_moves = 0; sh.err = 0; sh.trace_n = 0;
    sh.last_ncand[0] = sh.last_ncand[1] = sh.last_ncand[2] = 0;
    sh.best = h.best; sh.randomMP = h.randomMP; sh.iter_hits = h.iter_hits; sh.ins = h.insert_cid; sh.rem = h.remove_cid;
    sh.rng = h.rng; sh.hits = h.hits; sh.n_tests = 0; sh.n_ops = 0; sh.draws = 0; sh.n_nodes = 0;
    if (sh.B > (uint32_t)kMaxB) sh.B = kMaxB;
    // every workgroup must be resident before anyone waits for anyone: arrive, then wait for the others -- not for ever
    __hip_atomic_fetch_add(&P.hdr->arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    uint32_t ok = 1;
    for (;;) {
      if (__hip_atomic_load(&P.hdr->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { ok = 0; break; }
      if (__hip_atomic_load(&P.hdr->arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= T) break;
      if (__builtin_amdgcn_s_memrealtime() - t0 > 30ull * 100000ull) {      // 30 ms of the 100 MHz clock
        __hip_atomic_store(&P.hdr->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = 0;
        break;
      }
      __builtin_amdgcn_s_sleep(8);
    }
    // (a workgroup that saw everybody arrive may still be overtaken by another one's time-out: look once more)
    if (ok && __hip_atomic_load(&P.hdr->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) &&
        __hip_atomic_load(&P.hdr->arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < T) ok = 0;
    sh.ok = ok;
  }
  __syncthreads();
  if (!sh.ok) {
    if (tile == 0 && tid == 0) P.hdr->reason = CLIMB_ABORT;
    return;
  }

  for (;;) {
    // ---- step set-up
    if (tid == 0) {
      const uint32_t left = P.total - sh.pos + 1u;
      if (sh.B > left) sh.B = left;
      sh.wtail = 0; sh.rtail = 0; sh.rhead = 0; sh.ndone = 0; sh.nops = 0; sh.task = 0;
    }
    __syncthreads();
    const uint32_t B = sh.B;
    beat(P, tile, tid, 0, sh.steps); beat(P, tile, tid, 2, sh.pos); beat(P, tile, tid, 3, B); beat(P, tile, tid, 1, 1);
    // ---- (1)
    for (uint32_t u = (uint32_t)wave; u < 2u * B; u += kNW) enum_unit<KS, VW>(K, sh, P, u);
    __syncthreads();
    beat(P, tile, tid, 1, 2);
    // ---- (2)
    if (wave == 0) plan_and_discover<KS, VW>(K, sh);
    __syncthreads();
    beat(P, tile, tid, 4, sh.ncand); beat(P, tile, tid, 5, sh.nops); beat(P, tile, tid, 6, sh.rtail); beat(P, tile, tid, 1, 3);
    // ---- (3)
    refresh<KS, VW>(K, sh);
    __syncthreads();
    beat(P, tile, tid, 7, sh.ndone); beat(P, tile, tid, 1, 4);
    // ---- (4)
    const uint32_t ncand = sh.ncand;
    {
      const uint32_t nparts = 8u * sh.Beff;
      for (;;) {
        const uint32_t t = wave_fetch_add(&sh.task, 1u, lane);
        if (P.beat && tile == 0 && lane == 0) __hip_atomic_store(P.beat + 16 + wave, t | 0x100u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (t >= nparts) break;
        if (sh.pcnt[t] == 0u) continue;
        scan_part<KS, VW>(K, sh, t, P.maxtrav);
      }
      if (P.beat && tile == 0 && lane == 0) __hip_atomic_store(P.beat + 16 + wave, 0xFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __syncthreads();
    beat(P, tile, tid, 1, 5);
    // ---- (5) lengths = sum over tiles of (subtree scores at both ends of the prune branch + join cost)
    if (ncand) {
      const uint32_t slot = sh.xgen % 3u;
      unsigned long long *gs = P.gsum + (size_t)slot * kClimbCap;
      for (uint32_t c = (uint32_t)tid; c < ncand; c += kThreads) {
        uint32_t j = 0;
        while (j + 1u < sh.Beff && c >= sh.pn_off[j + 1u]) j++;
        const uint32_t p = sh.pn_p[j], q = K.bk[p];
        const uint32_t val = K.cost[c] + ld_sct<KS, VW>(K, p) + ld_sct<KS, VW>(K, q);
        __hip_atomic_fetch_add(gs + c, (1ull << 40) | (unsigned long long)val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      for (uint32_t c = (uint32_t)tid; c < ncand; c += kThreads) {
        unsigned long long v;
        uint32_t spins = 0;
        for (;;) {
          v = __hip_atomic_load(gs + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if ((uint32_t)(v >> 40) >= T) break;
          if ((++spins & 1023u) == 0u) {
            // a tile that never arrives (it cannot happen once every workgroup is resident) must not hang the GPU: whoever
            // notices first tells everybody through the header, and the launch ends with CLIMB_ERROR
            if (__hip_atomic_load(&P.hdr->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { sh.err = 1u; break; }
            if (spins > (1u << 20)) { __hip_atomic_store(&P.hdr->abort, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); sh.err = 1u; break; }
          }
          __builtin_amdgcn_s_sleep(2);
        }
        K.cost[c] = (uint32_t)(v & kValMask);
      }
      __syncthreads();
      // the slot used one exchange ago has been read by everybody who got here: its turn comes again two exchanges on
      if (tile == sh.xgen % T) {
        const uint32_t zs = (sh.xgen + 2u) % 3u;
        unsigned long long *gz = P.gsum + (size_t)zs * kClimbCap;
        for (uint32_t c = (uint32_t)tid; c < sh.last_ncand[zs]; c += kThreads) __hip_atomic_store(gz + c, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __syncthreads();
      if (tid == 0) { sh.last_ncand[slot] = ncand; sh.xgen++; }
    }
    __syncthreads();
    beat(P, tile, tid, 1, 6);
    // ---- (6)
    if (wave == 0) decide<KS, VW>(K, sh, P, tile);
    __syncthreads();
    beat(P, tile, tid, 8, sh.err);
    if (sh.exit_reason != CLIMB_RUNNING) break;
  }
  beat(P, tile, tid, 1, 9);
  // ---- hand the state back
  if (tile == 0) {
    for (uint32_t i = (uint32_t)tid; i < ns; i += kThreads) P.bk[i] = K.bk[i];
    if (tid == 0) {
      ClimbHeader *h = P.hdr;
      h->rng = sh.rng; h->hits = sh.hits; h->best = sh.best; h->randomMP = sh.randomMP; h->iter_hits = sh.iter_hits;
      h->pos = sh.pos; h->insert_cid = sh.ins; h->remove_cid = sh.rem; h->n_moves = sh.n_moves; h->reason = sh.exit_reason;
      h->err = sh.err; h->steps = sh.steps; h->n_tests = sh.n_tests; h->n_ops = sh.n_ops; h->draws = sh.draws;
      h->n_scanned_nodes = sh.n_nodes; h->since_move = sh.since_move; h->batch = sh.B;
      h->pad[0] = sh.trace_n;
    }
  }
}

template <int KS, int VW>
size_t lds_bytes(uint32_t ns)
{
  size_t at = (sizeof(Sh) + 15) & ~(size_t)15;
  at += (((size_t)ns * 4) + 15) & ~(size_t)15;
  at += (size_t)kClimbCap * 4;
  at += (size_t)kNW * kDepth * KS * VW * 64 * 4;
  at += (size_t)kNW * 8 * sizeof(uint2);
  at += (((size_t)ns * 2) + 15) & ~(size_t)15;
  at += (((size_t)ns * 2) + 15) & ~(size_t)15;
  at += (((size_t)(ns + 16) * 2) + 15) & ~(size_t)15;
  at += (size_t)kClimbCap * 2;
  at += ns;
  return (at + 15) & ~(size_t)15;
}

template <int KS, int VW>
hipError_t launch_t(hipStream_t st, const ClimbParams &p)
{
  const size_t lds = lds_bytes<KS, VW>(p.nslots);
  static thread_local int attr_dev = -1;
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (lds > 64 * 1024 || attr_dev != dev) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_climb<KS, VW>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_dev = dev;
  }
  hipLaunchKernelGGL((k_climb<KS, VW>), dim3(p.tiles), dim3(kThreads), lds, st, p);
  return hipGetLastError();
}

}  // namespace

static inline uint32_t slots_of(int n) { return (uint32_t)n + 3u * (uint32_t)(n - 1); }

int climb_tiles(const Geometry &g, int vw) { return (g.Wp + 16 * vw - 1) / (16 * vw); }

size_t climb_lds_bytes(const Geometry &g, int n_taxa, int vw)
{
  const uint32_t ns = slots_of(n_taxa);
  if (g.S == 4) return vw == 1 ? lds_bytes<1, 1>(ns) : vw == 2 ? lds_bytes<1, 2>(ns) : lds_bytes<1, 4>(ns);
  return lds_bytes<5, 1>(ns);
}

bool climb_supported(const Geometry &g, int n_taxa, int maxtrav)
{
  if (g.sankoff || g.big) return false;
  if (g.S != 4 && g.S != 20) return false;
  if (maxtrav < 1 || maxtrav > kDepth) return false;
  if (slots_of(n_taxa) + 16u >= 0xFFFFu) return false;
  if (climb_tiles(g, 1) >= (1 << 20)) return false;
  return climb_lds_bytes(g, n_taxa, 1) <= 150 * 1024;
}

hipError_t launch_climb(hipStream_t st, const Geometry &g, int vw, const ClimbParams &p)
{
  if (g.S == 4) {
    if (vw == 1) return launch_t<1, 1>(st, p);
    if (vw == 2) return launch_t<1, 2>(st, p);
    return launch_t<1, 4>(st, p);
  }
  return launch_t<5, 1>(st, p);
}

}  // namespace mpf
