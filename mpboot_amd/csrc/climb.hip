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

#ifndef MPF_CLIMB_MAXB
#define MPF_CLIMB_MAXB 16
#endif
constexpr int kMaxB = MPF_CLIMB_MAXB;                    // prune nodes per step (their 4 x 16 parts are the lanes of one wave in plan_and_discover)
constexpr int kMaxUnits = 2 * kMaxB;         // (prune node, side)
constexpr int kMaxParts = 2 * kMaxUnits;     // (prune node, side, gap end): one DFS program each
constexpr int kDepth = 6;                    // deepest radius
constexpr int kProgStride = 64;              // program entries per part: at most 63 expansions at radius 6
constexpr uint32_t kNone16 = 0xFFFFu;
constexpr unsigned long long kValMask = (1ull << 40) - 1ull;
// heap lanes (root 1, children 2h / 2h + 1) in the subtree of the root's first child
constexpr unsigned long long kLeftSub = (1ull << 2) | (3ull << 4) | (0xFull << 8) | (0xFFull << 16) | (0xFFFFull << 32);
// claim word of a vector: epoch << 17 | index among the step's refresh ops << 3 | stale inputs not yet recomputed
constexpr int kEpochShift = 17;
constexpr uint32_t kIdxMask = 0x3FFFu;
constexpr uint32_t kEpochLimit = 32000u;     // (15 bits; a launch that gets there hands back and is started again)
// Refresh programs: the first kLcap ops of a step's closure get descriptors (consumers, operand slots); larger closures (the
// first step of a launch recomputes ~n vectors) run on the plain dataflow path.
constexpr uint32_t kLcap = 256;
constexpr uint32_t kNoSlot = 0xFFu;
template <int KS, int VW> struct Cfg {
  static constexpr int R = KS * VW;                                   // registers per vector tile
  // waves per workgroup: sixteen where the tiles are small (the scan's programs and the refresh's chains are dealt to them),
  // eight where a wave's parked up-vectors would not fit the LDS sixteen times
  // (DNA on 128-word tiles -- many climbs side by side on one chip, 13 workgroups each at C3 --: four, or the parked up-vectors
  //  would not fit beside the control state)
  static constexpr int NW = R <= 2 ? 16 : (KS == 1 && VW == 8) ? 4 : 8;
  static constexpr int NT = NW * 64;
  static constexpr int PF = R <= 2 ? 8 : 4;                           // expansions whose child vectors are requested together
  // operand slots of the refresh (vector + per-lane subtree scores) and the parked up-vectors of the scan are never alive
  // together: one LDS region serves both
  static constexpr size_t kSlotBytes = (size_t)(R + 1) * 64 * 4;
  // k_climb_many on the word-major shape: TWELVE waves (three a SIMD, 168 registers each) -- a climb that is one workgroup is alone
  // on its CU, and what it waits for most is its own LDS, scalar and L2 round trips: more waves in flight is what helps
  static constexpr int NW_MANY = KS == 4 ? 12 : NW;
  static constexpr int NT_MANY = NW_MANY * 64;
  // the scans' parked up-vectors: [wave][depth 1..5][R][64]
  __host__ __device__ static constexpr size_t pend_bytes(int nw) { return (size_t)nw * 5 * R * 64 * 4; }
  // (64 KB where the control state leaves them, down to 40 KB -- or the parked vectors' size -- at a thousand taxa: region_bytes)
  __host__ __device__ static constexpr size_t region_max(int nw) { return pend_bytes(nw) > 65536 ? pend_bytes(nw) : 65536; }
  __host__ __device__ static constexpr size_t region_min(int nw) { return pend_bytes(nw) > 40960 ? pend_bytes(nw) : 40960; }
};
constexpr size_t kLdsBudget = 158 * 1024;      // of the CU's 160 KB (all of it dynamic: the kernels have no static LDS)

#include "quadtile.hpp"

struct Unit { uint16_t x, s, xa, xb, mt, p; };

// the control block every workgroup keeps in LDS (identical everywhere: it is computed from exchanged sums only)
struct Sh {
  uint32_t pos, B, Beff, epoch, exit_reason, ncand, since_move, steps, xgen, n_moves, err, consumed;
  uint32_t last_ncand[3];
  uint32_t best, randomMP, iter_hits;
  int32_t ins, rem;
  unsigned long long rng, hits, n_tests, n_ops, draws, n_nodes;
  uint32_t wtail, rtail, rhead, ndone, nops, task, ok, trace_n, use_static, pn_base[kMaxB];
  uint32_t xcc, xm;                          // this workgroup's XCD and how many of the launch's workgroups share it
  uint32_t startMP, sweeps;                  // ClimbParams::sweeps_inside: the sweep under way started at this length; sweeps begun inside the launch
  uint32_t gap_ema;                          // recent distance between accepted moves in prune nodes, x 8 (decide: the batch after a move)
  uint32_t qtail;                            // vectors the step's scans read, listed by the enumeration (validity looked at later)
  uint32_t moved, einv, inv5[5];             // the move decide_select applied: stamp and touched nodes for the invalidation walk
  Unit unit[kMaxUnits];
  uint32_t pcnt[kMaxParts], poff[kMaxParts], pE[kMaxParts];
  uint32_t pn_off[kMaxB], pn_cnt[kMaxB], pn_np[kMaxB], pn_p[kMaxB];
  uint32_t ntasks;
  uint8_t tl[2 * kMaxParts];                  // scan tasks of the step: part << 1 | half
  unsigned long long tph[16], tlast;          // time per phase (100 MHz ticks), workgroup 0
  uint32_t c_rounds, c_inv, c_chains, c_parts, c_inv2, c_dynops;
  unsigned long long clk0, rt0;
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
  uint2 *prog;       // [part][64] DFS programs of the step's scans
  uint32_t *stage;   // [slots][R + 1][64] operand slots of the refresh: vector registers + per-lane subtree scores
  uint2 *D;          // [kLcap] refresh op: r | slot of operand a << 16 | slot of operand b << 24 ; a | b << 16
  uint2 *CONS;       // [kLcap][2] who consumes an op's result: consumer idx | which << 8 | its stale inputs << 9 | slot of its other operand << 16 ; consumer's vector | other operand's vector << 16
  uint32_t *NC;      // [kLcap] number of consumers registered
  uint32_t *PEND;    // [kLcap] stale inputs not yet recomputed (joins)
  uint16_t *OL;      // [kLcap] the first refresh ops, by index
  uint16_t *ord;     // [total] the sweep's visiting order
  uint16_t *Q;       // [ns] what the enumeration lists (lives in the stage / pend region, idle between decide and refresh)
  uint2 *SD;         // [kLcap] (over CONS) the refresh of a step as ONE sequence (schedule_private): r | how operand a comes << 16 | operand b << 18 ; a | b << 16
  uint32_t *PEND0;   // [kLcap] PEND as the link pass left it  } a workgroup that works through several tiles per step runs the
  uint16_t *R0;      // [kLcap] the chain starts of the link pass } same refresh once per tile: what a run consumes is put back
  uint16_t *TF;      // [kLcap] k_climb_many: the step (its epoch) whose refresh tile k of this workgroup has been taken through
  unsigned long long pre, ancl, lsub;   // heap-index relations of this lane (enumeration)
  uint32_t n, ns, SW4;
  uint32_t slots;    // operand slots the region holds
  int lane, wave;
  __amdgpu_buffer_rsrc_t rsrc, rsrc_s;   // the vector store; this tile's per-lane subtree scores [vector][16 word groups]
  uint32_t voff[KS], svoff;
  uint32_t zero;     // 0, opaque to the compiler (keeps per-lane LDS adds from being folded into a wave reduction)
  bool cnt_lane;     // this lane's popcounts count (first lane of a quad, inside the row)
  bool st_lane;      // this lane's words exist
};

template <int KS, int VW>
__device__ __forceinline__ void ld(const Kx<KS, VW> &K, QT<KS, VW> &t, uint32_t cid) { qload<KS, VW>(t, K.rsrc, K.voff, cid * K.SW4); }

// tr->parsimonyScore[] (reference sprparsimony.cpp:874) of this tile, kept PER LANE: every lane of a quad holds the mutations
// of its word group in the subtree, so a refresh op adds three registers and no reduction across lanes is needed until a
// prune branch's base length is wanted
template <int KS, int VW>
__device__ __forceinline__ uint32_t ld_sl(const Kx<KS, VW> &K, uint32_t cid)
{
  return cid < K.n ? 0u : __builtin_amdgcn_raw_buffer_load_b32(K.rsrc_s, K.svoff, cid * (kWordMajor<KS> ? 256u : 64u), 0);
}
template <int KS, int VW>
__device__ __forceinline__ void st_sl(const Kx<KS, VW> &K, uint32_t cid, uint32_t v)
{
  if (K.cnt_lane) __builtin_amdgcn_raw_buffer_store_b32(v, K.rsrc_s, K.svoff, cid * (kWordMajor<KS> ? 256u : 64u), 0);
}

// a vector the scans of this step read: listed once per step (the claim word's epoch de-duplicates).  Whether it is stale is
// looked at AFTER the enumeration (filter_required): the enumeration of a step runs beside the invalidation walk of the move
// before it, which is still clearing validity flags.
template <int KS, int VW>
__device__ __forceinline__ void require(const Kx<KS, VW> &K, Sh &sh, uint32_t c, uint32_t epoch)
{
  if (c < K.n) return;                                   // tips are never stale
  const uint32_t old = __hip_atomic_fetch_max(&K.cl[c], epoch << kEpochShift, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  if ((old >> kEpochShift) != epoch) {
    const uint32_t slot = __hip_atomic_fetch_add(&sh.qtail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    K.Q[slot] = (uint16_t)c;
  }
}

// the stale ones among the listed vectors become the closure's work list (all waves; order is free: every workgroup decides
// from the exchanged sums only, never from the order in which it refreshed its own tile)
template <int KS, int VW>
__device__ __forceinline__ void filter_required(const Kx<KS, VW> &K, Sh &sh, int tid, int nthreads)
{
  const uint32_t nq = sh.qtail;
  for (uint32_t base = 0; base < nq; base += (uint32_t)nthreads) {
    const uint32_t i = base + (uint32_t)tid;
    const uint32_t c = i < nq ? (uint32_t)K.Q[i] : 0u;
    const bool stale = i < nq && !K.valid[c];
    const unsigned long long m = __ballot((int)stale);
    if (m) {
      const uint32_t at = wave_fetch_add(&sh.wtail, (uint32_t)__builtin_popcountll(m), K.lane);
      if (stale) K.W[at + (uint32_t)__builtin_popcountll(m & ((1ull << K.lane) - 1ull))] = (uint16_t)c;
    }
  }
}

// ---- (1) one side of one prune node: which scans exist (rearrangeParsimony's tests, :2304-2310, :2330-2347) and, per gap
// end, the DFS of addTraverseParsimony (:2208-2218) as a program.  Lane h is the heap index of an expansion (root 1 = the gap
// end itself, children 2h / 2h + 1, depth <= 5 at radius 6) and walks down from the root by the bits of h -- at most six
// dependent LDS rounds for a whole neighbourhood.  Pre-order positions and the reference's candidate indices follow from
// ballots and fixed heap relations: an expansion's entry goes to prog[#expansions before it in pre-order]; its first
// candidate's index is 2 * #tested expansions before it - #tested ancestors whose LEFT subtree holds it (those have emitted
// one candidate so far, the others two), the second one follows the left subtree's candidates.
template <int KS, int VW>
__device__ __forceinline__ void enum_unit(const Kx<KS, VW> &K, Sh &sh, const ClimbParams &P, uint32_t u)
{
  const uint32_t n = K.n, epoch = sh.epoch;
  const int lane = K.lane;
  const uint32_t j = u >> 1, side = u & 1u;
  const uint32_t p = rfl((uint32_t)K.ord[sh.pos - 1u + j]);
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
  uint32_t cnt[2] = {0u, 0u}, ne[2] = {0u, 0u};
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
      const unsigned long long Em = __ballot((int)ex), Tm = __ballot((int)tested);
      const uint32_t part = 2u * u + e;
      cnt[e] = 2u * (uint32_t)__builtin_popcountll(Tm);
      ne[e] = cnt[e] ? (uint32_t)__builtin_popcountll(Em) | ((uint32_t)__builtin_popcountll(Em & kLeftSub) << 8) : 0u;   // expansions | of them below the root's first child
      if (ex && cnt[e]) {
        const uint32_t pos = (uint32_t)__builtin_popcountll(Em & K.pre);
        const uint32_t k1 = 2u * (uint32_t)__builtin_popcountll(Tm & K.pre) - (uint32_t)__builtin_popcountll(Tm & K.ancl);
        const uint32_t k2 = k1 + 1u + 2u * (uint32_t)__builtin_popcountll(Tm & K.lsub);
        const uint32_t save2 = (h < 32 && ((Em >> (2 * h + 1)) & 1ull)) ? 1u : 0u;   // the second child is expanded later: keep its up-vector
        const uint32_t fp = (h > 1 && (h & 1)) ? 1u : 0u;                            // this node IS a second child: its up-vector waits in LDS
        K.prog[part * kProgStride + pos] =
            make_uint2(c1 | (c2 << 16), (uint32_t)(d + 1) | (tested ? 16u : 0u) | (save2 << 5) | (fp << 6) | (k1 << 8) | (k2 << 16));
        if (tested) { K.cq[part * 128u + k1] = (uint16_t)c1; K.cq[part * 128u + k2] = (uint16_t)c2; }   // the insertion branches by candidate index
        require<KS, VW>(K, sh, c1, epoch);
        require<KS, VW>(K, sh, c2, epoch);
      }
      if (lane == 0 && cnt[e]) require<KS, VW>(K, sh, other, epoch);
    }
    if (lane == 0 && (cnt[0] | cnt[1])) {
      require<KS, VW>(K, sh, s, epoch);                  // the pruned subtree, and both ends of the prune branch for the base length
      require<KS, VW>(K, sh, x, epoch);
    }
  }
  if (lane == 0) {
    Unit un;
    un.x = (uint16_t)x; un.s = (uint16_t)s; un.xa = (uint16_t)xa; un.xb = (uint16_t)xb; un.mt = (uint16_t)mt; un.p = (uint16_t)p;
    sh.unit[u] = un;
    sh.pcnt[2u * u] = cnt[0]; sh.pcnt[2u * u + 1u] = cnt[1];
    sh.pE[2u * u] = ne[0]; sh.pE[2u * u + 1u] = ne[1];
  }
}

// ---- (2) wave 0: output layout of the step, then the closure of stale inputs
template <int KS, int VW>
__device__ __forceinline__ void plan_and_discover(const Kx<KS, VW> &K, Sh &sh, const bool no_slots)
{
  const int lane = K.lane;
  const uint32_t n = K.n, epoch = sh.epoch, B = sh.B;
  // candidates are laid out part after part = the reference's order (p side: first gap end, second gap end; then the q side)
  const uint32_t nparts = 4u * B;
  // prefix sums over the parts without LDS traffic: the four parts of a prune node are the four lanes of a DPP quad, the (at
  // most eight) prune-node totals go through the scalar unit
  const uint32_t c = (uint32_t)lane < nparts ? sh.pcnt[lane] : 0u;
  const uint32_t q0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)c, 0x00, 0xF, 0xF, false);   // quad_perm [0,0,0,0]
  const uint32_t q1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)c, 0x55, 0xF, 0xF, false);   // quad_perm [1,1,1,1]
  const uint32_t q2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)c, 0xAA, 0xF, 0xF, false);   // quad_perm [2,2,2,2]
  const uint32_t q3 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)c, 0xFF, 0xF, 0xF, false);   // quad_perm [3,3,3,3]
  const uint32_t qsum = q0 + q1 + q2 + q3;
  const int lq = lane & 3;
  const uint32_t in_quad = (lq > 0 ? q0 : 0u) + (lq > 1 ? q1 : 0u) + (lq > 2 ? q2 : 0u);
  uint32_t pn_tot[kMaxB], pn_pre[kMaxB + 1];
  pn_pre[0] = 0;
#pragma unroll
  for (int j = 0; j < kMaxB; j++) {
    pn_tot[j] = (uint32_t)__builtin_amdgcn_readlane((int)qsum, 4 * j);
    pn_pre[j + 1] = pn_pre[j] + pn_tot[j];
  }
  // speculation is cut where the step's candidate buffer ends (one prune node never exceeds it)
  uint32_t Beff = 0, tot_eff = 0;
#pragma unroll
  for (int j = 0; j < kMaxB; j++)
    if ((uint32_t)j < B && pn_pre[j + 1] <= kClimbCap) { Beff = (uint32_t)j + 1u; tot_eff = pn_pre[j + 1]; }
  uint32_t my_pre = 0;
#pragma unroll
  for (int j = 0; j < kMaxB; j++) my_pre = (lane >> 2) == j ? pn_pre[j] : my_pre;
  const uint32_t excl = my_pre + in_quad;
  const uint32_t e2 = my_pre + q0 + q1, e4 = my_pre + qsum;
  if ((uint32_t)lane < nparts) {
    const bool live = (uint32_t)(lane >> 2) < Beff;
    sh.poff[lane] = excl;
    if (!live) { sh.pcnt[lane] = 0u; sh.pE[lane] = 0u; }
    if ((lane & 3) == 0 && live) {
      const int j = lane >> 2;
      sh.pn_off[j] = excl;
      sh.pn_cnt[j] = e4 - excl;
      sh.pn_np[j] = e2 - excl;
      sh.pn_p[j] = sh.unit[2 * j].p;
    }
  }
  if (lane == 0) { sh.Beff = Beff; sh.ncand = Beff ? tot_eff : 0u; }
  {
    // the scan's task list: one entry per program, two for a long one (scan_part); dealt to the waves round-robin
    const bool live = (uint32_t)lane < nparts && (uint32_t)(lane >> 2) < Beff;
    const uint32_t E = live ? (sh.pE[lane] & 0xFFu) : 0u;
    const uint32_t nt = E == 0u ? 0u : E >= 6u ? 2u : 1u;
    const unsigned long long m1 = __ballot((int)(nt >= 1u)), m2 = __ballot((int)(nt == 2u));
    const unsigned long long lt = (1ull << lane) - 1ull;
    const uint32_t at = (uint32_t)__builtin_popcountll(m1 & lt) + (uint32_t)__builtin_popcountll(m2 & lt);
    if (nt >= 1u) sh.tl[at] = (uint8_t)(lane << 1);
    if (nt == 2u) sh.tl[at + 1u] = (uint8_t)((lane << 1) | 1);
    if (lane == 0) sh.ntasks = (uint32_t)__builtin_popcountll(m1) + (uint32_t)__builtin_popcountll(m2);
  }
  // closure: every lane follows one chain of stale inputs; forks go to the shared worklist.  Every vector gets an index (the
  // order in which the lanes take them up); the first kLcap are listed for the link pass below.
  // (a lane walks several links between two looks at the worklist: the bookkeeping of a round costs as much as a link)
  uint32_t head = 0;
  bool have = false;
  uint32_t item = 0;
  uint32_t round = 0;
  if (lane == 0) sh.nops = 0;
  for (;; round++) {
    if (round > K.ns) { if (lane == 0) sh.err = 3u; break; }
    const uint32_t tail = *(volatile uint32_t *)&sh.wtail;
    {
      const unsigned long long need = __ballot((int)!have);
      const uint32_t rank = (uint32_t)__builtin_popcountll(need & ((1ull << lane) - 1ull));
      const uint32_t avail = tail - head;
      if (!have && rank < avail) { item = K.W[head + rank]; have = true; }
      const uint32_t want = (uint32_t)__builtin_popcountll(need);
      head += want < avail ? want : avail;
    }
    if (!__ballot((int)have)) break;
#pragma unroll 1
    for (int step = 0; step < 4; step++) {
      if (have) {
        const uint32_t r = item;
        uint32_t r1, r2;
        ring2(r, n, r1, r2);
        // two LDS round trips per link: (1) the inputs' ids and this op's index, (2) the inputs' claims and validity -- every
        // read unconditional (tips count as valid vectors), nothing waits in between
        const uint32_t a = K.bk[r1], b = K.bk[r2];
        const uint32_t idx = __hip_atomic_fetch_add(&sh.nops, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        // (a claim on a vector that turns out to be valid, or on a tip, means nothing: only stale vectors are ever looked up
        //  by their claim)
        const uint32_t oa = __hip_atomic_fetch_max(&K.cl[a], epoch << kEpochShift, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const uint32_t ob = __hip_atomic_fetch_max(&K.cl[b], epoch << kEpochShift, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const uint32_t va = K.valid[a], vb = K.valid[b];
        const bool sa = !va, sb = !vb;
        const bool wa = sa & ((oa >> kEpochShift) != epoch), wb = sb & ((ob >> kEpochShift) != epoch);
        const uint32_t ns = (sa ? 1u : 0u) + (sb ? 1u : 0u);
        // (r is claimed already: other lanes' fetch_max leave the word as it is)
        K.cl[r] = (epoch << kEpochShift) | ((idx < kIdxMask ? idx : kIdxMask) << 3) | ns;
        if (idx < kLcap) K.OL[idx] = (uint16_t)r;
        if (!ns) {
          const uint32_t slot = __hip_atomic_fetch_add(&sh.rtail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          K.R[slot] = (uint16_t)r;                       // both inputs valid: a chain starts here
        }
        if (wa) {
          item = a;
          if (wb) {
            const uint32_t slot = __hip_atomic_fetch_add(&sh.wtail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            K.W[slot] = (uint16_t)b;
          }
        } else if (wb) item = b;
        else have = false;
      }
    }
  }
  const uint32_t nops = *(volatile uint32_t *)&sh.nops;
  if (lane == 0) { sh.n_ops += nops; sh.c_rounds += round; sh.c_chains += sh.rtail; }
  // link pass: every op of a closure that fits gets a descriptor, a slot for each operand that is valid already (staged into
  // LDS before the chains start), and registers with its stale inputs as their consumer -- what a chain needs to go from one
  // link to the next is then ONE entry, read a link ahead
  const bool linkable = nops > 0 && nops <= kLcap;
  uint32_t mode = linkable ? 1u : 0u;
  if (linkable) {
    uint32_t slot_base = 0, nstart = 0;
    for (uint32_t base = 0; base < nops; base += 64u) {
      const uint32_t i = base + (uint32_t)lane;
      const bool act = i < nops;
      uint32_t r = 0, a = 0, b = 0;
      bool sa = false, sb = false;
      if (act) {
        r = K.OL[i];
        const uint32_t r1 = nxc(r, n);
        a = K.bk[r1];
        b = K.bk[nxc(r1, n)];
        sa = a >= n && !K.valid[a];
        sb = b >= n && !K.valid[b];
      }
      const uint32_t nst = (sa ? 1u : 0u) + (sb ? 1u : 0u);
      // an operand gets a slot if it is valid (staged from memory before the chains start) or if the op is a join of two
      // stale inputs (whoever arrives first leaves its result there for the other one)
      const bool wa = act && (!sa || nst == 2u), wb = act && (!sb || nst == 2u);
      const unsigned long long ma = __ballot((int)wa), mb = __ballot((int)wb);
      const unsigned long long lt = (1ull << lane) - 1ull;
      uint32_t s0 = slot_base + (uint32_t)__builtin_popcountll(ma & lt) + (uint32_t)__builtin_popcountll(mb & lt);
      uint32_t slotA = kNoSlot, slotB = kNoSlot;
      if (wa) { slotA = s0 < K.slots ? s0 : kNoSlot; s0++; }
      if (wb) slotB = s0 < K.slots ? s0 : kNoSlot;
      slot_base += (uint32_t)__builtin_popcountll(ma) + (uint32_t)__builtin_popcountll(mb);
      if (act) {
        K.D[i] = make_uint2(r | (slotA << 16) | (slotB << 24), a | (b << 16));
        K.PEND[i] = nst;
        if (sa) {
          const uint32_t ik = (K.cl[a] >> 3) & kIdxMask;
          const uint32_t sl = __hip_atomic_fetch_add(&K.NC[ik], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (sl < 2u) K.CONS[2u * ik + sl] = make_uint2(i | (0u << 8) | (nst << 9) | (slotB << 16) | (slotA << 24), r | (b << 16));
        }
        if (sb) {
          const uint32_t ik = (K.cl[b] >> 3) & kIdxMask;
          const uint32_t sl = __hip_atomic_fetch_add(&K.NC[ik], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (sl < 2u) K.CONS[2u * ik + sl] = make_uint2(i | (1u << 8) | (nst << 9) | (slotA << 16) | (slotB << 24), r | (a << 16));
        }
      }
      // chains start at the ops without a stale input
      const unsigned long long ms = __ballot((int)(act && nst == 0u));
      if (act && nst == 0u) K.R[nstart + (uint32_t)__builtin_popcountll(ms & lt)] = (uint16_t)i;
      nstart += (uint32_t)__builtin_popcountll(ms);
    }
    if (lane == 0) sh.rtail = nstart;
    // (refresh_private stages nothing: there the closure keeps its descriptors however many operands it has)
    if (slot_base > K.slots && !no_slots) {
      // slots ran out: this step's refresh takes the plain path, whose chain starts are vectors, not op indices
      for (uint32_t k = (uint32_t)lane; k < nstart; k += 64u) K.R[k] = K.OL[K.R[k]];
      mode = 0u;
    }
  }
  if (lane == 0) {
    sh.use_static = mode;
    if (nops) { if (mode) sh.c_parts++; else { sh.c_inv2++; sh.c_dynops += nops; } }
  }
}

// ---- (3) refresh of the claimed vectors (newviewParsimonyIterativeFast, :554-878, on exactly the stale vectors the scans
// read).  Closures that fit the link pass: every wave first copies the VALID operands of its share of the ops into their LDS
// slots -- all those loads are in flight together, one memory round trip for the whole refresh instead of one per link of a
// chain -- then the chains run: the running result and its per-lane scores stay in registers, the next link's consumer entry
// and its other operand are requested while the current link is combined.  Where two stale inputs meet, whoever arrives
// second goes on (the first one's result is read back from memory: joins are one op in ten).
template <int KS, int VW>
__device__ __forceinline__ void refresh_static(const Kx<KS, VW> &K, Sh &sh, bool prof, const uint32_t nw)
{
  constexpr int R = Cfg<KS, VW>::R;
  const uint32_t nops = sh.nops;
  const int lane = K.lane;
  // -- stage
  for (uint32_t base = (uint32_t)K.wave * 4u; base < nops; base += nw * 4u) {
    QT<KS, VW> ta[4], tb[4];
    uint32_t la[4], lb[4], da[4], db[4];
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
      const uint32_t i = base + (uint32_t)kk;
      da[kk] = db[kk] = kNoSlot;
      if (i < nops) {
        const uint2 d = K.D[i];
        const uint32_t dx = rfl(d.x), dy = rfl(d.y), join = rfl(K.PEND[i]) == 2u ? 1u : 0u;
        da[kk] = join ? kNoSlot : (dx >> 16) & 0xFFu;
        db[kk] = join ? kNoSlot : dx >> 24;
        if (da[kk] != kNoSlot) { ld<KS, VW>(K, ta[kk], dy & 0xFFFFu); la[kk] = ld_sl<KS, VW>(K, dy & 0xFFFFu); }
        if (db[kk] != kNoSlot) { ld<KS, VW>(K, tb[kk], dy >> 16); lb[kk] = ld_sl<KS, VW>(K, dy >> 16); }
      }
    }
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
      if (da[kk] != kNoSlot) {
        uint32_t *sp = K.stage + (size_t)da[kk] * ((R + 1) * 64) + lane;
#pragma unroll
        for (int k = 0; k < R; k++) sp[k * 64] = ta[kk].v[k / VW][k % VW];
        sp[R * 64] = la[kk];
      }
      if (db[kk] != kNoSlot) {
        uint32_t *sp = K.stage + (size_t)db[kk] * ((R + 1) * 64) + lane;
#pragma unroll
        for (int k = 0; k < R; k++) sp[k * 64] = tb[kk].v[k / VW][k % VW];
        sp[R * 64] = lb[kk];
      }
    }
  }
  __syncthreads();
  if (prof && lane == 0) sh.tph[8] += __builtin_amdgcn_s_memrealtime() - sh.tlast;
  // -- chains
  auto operand = [&](QT<KS, VW> &t, uint32_t &sl, uint32_t slot, uint32_t cid) {
    if (slot != kNoSlot) {
      const uint32_t *sp = K.stage + (size_t)slot * ((R + 1) * 64) + lane;
#pragma unroll
      for (int k = 0; k < R; k++) t.v[k / VW][k % VW] = sp[k * 64];
      sl = sp[R * 64];
    } else {
      // (rare: a join's other input, or no slot left.  The loads are waited for right here: left pending, the wait would be
      //  placed behind the merge of the two paths and the usual path -- LDS only -- would sit out its own older STORES with
      //  it: gfx950 counts loads and stores in one in-order counter)
      ld<KS, VW>(K, t, cid);
      sl = ld_sl<KS, VW>(K, cid);
#pragma unroll
      for (int k = 0; k < R; k++) asm volatile("" : "+v"(t.v[k / VW][k % VW]));
      asm volatile("" : "+v"(sl));
    }
  };
  QT<KS, VW> c, ta, tb;
  uint32_t la = 0, lb = 0, lc = 0;
  for (uint32_t si = (uint32_t)K.wave;; si += nw) {       // chain starts are dealt round-robin (late ones -- rare -- land behind the list)
    uint32_t st = kNone16, spins = 0;
    for (;;) {
      st = *(volatile uint16_t *)&K.R[si];                // (entries beyond the starts written so far read "none")
      if (st != kNone16) break;
      if (*(volatile uint32_t *)&sh.ndone >= nops) break;
      if (++spins > (1u << 22)) { sh.err = 2u; break; }   // (bounded like every wait in this kernel)
      __builtin_amdgcn_s_sleep(1);
    }
    st = rfl(st);
    if (st == kNone16) break;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const unsigned long long ct0 = prof ? __builtin_amdgcn_s_memrealtime() : 0ull;
    uint32_t nlinks = 0;
    uint32_t i = st, r;
    {
      const uint2 d = K.D[i];
      const uint32_t dx = rfl(d.x), dy = rfl(d.y);
      r = dx & 0xFFFFu;
      operand(ta, la, (dx >> 16) & 0xFFu, dy & 0xFFFFu);
      operand(tb, lb, dx >> 24, dy >> 16);
    }
    bool pre = false;                                     // the consumer entries of op i are in flight already
    uint32_t ncv = 0;
    uint2 e0 = make_uint2(0u, 0u), e1 = make_uint2(0u, 0u);
    for (uint32_t link = 0;; link++) {
      if (link > kLcap) { if (lane == 0) sh.err = 5u; break; }
      // who waits for this result
      if (!pre) { ncv = K.NC[i]; e0 = K.CONS[2u * i]; e1 = K.CONS[2u * i + 1u]; }
      const uint32_t nc = rfl(ncv), x0 = rfl(e0.x), y0 = rfl(e0.y);
      // The usual link: ONE consumer whose other input is valid and staged.  Then the next link is known before this one is
      // combined: its other operand and ITS consumer entries are requested now and arrive behind the arithmetic -- a chain
      // costs one LDS round trip per link, hidden.
      const bool fast = nc == 1u && ((x0 >> 9) & 3u) == 1u && ((x0 >> 16) & 0xFFu) != kNoSlot;
      QT<KS, VW> pt;
      uint32_t pl = 0, ncv2 = 0;
      uint2 f0 = make_uint2(0u, 0u), f1 = make_uint2(0u, 0u);
      if (fast) {
        const uint32_t j = x0 & 0xFFu;
        const uint32_t *sp = K.stage + (size_t)((x0 >> 16) & 0xFFu) * ((R + 1) * 64) + lane;
#pragma unroll
        for (int k = 0; k < R; k++) pt.v[k / VW][k % VW] = sp[k * 64];
        pl = sp[R * 64];
        ncv2 = K.NC[j]; f0 = K.CONS[2u * j]; f1 = K.CONS[2u * j + 1u];
      }
      const uint32_t cost = q_fitch<KS, VW>(c, ta, tb);
      lc = cost + la + lb;
      if (K.st_lane) qstore<KS, VW>(c, K.rsrc, K.voff, r * K.SW4);
      st_sl<KS, VW>(K, r, lc);
      if (lane == 0) K.valid[r] = 1;
      if (fast) {
        if (lane == 0) __hip_atomic_fetch_add(&sh.ndone, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        nlinks++;
        i = x0 & 0xFFu;
        r = y0 & 0xFFFFu;
        if (((x0 >> 8) & 1u) == 0u) { ta = c; la = lc; tb = pt; lb = pl; }
        else { tb = c; lb = lc; ta = pt; la = pl; }
        ncv = ncv2; e0 = f0; e1 = f1;
        pre = true;
        continue;
      }
      pre = false;
      uint32_t nxt = kNone16, nx_x = 0, nx_y = 0;
#pragma unroll
      for (int u = 0; u < 2; u++) {
        if ((uint32_t)u < nc) {
          const uint32_t ex_ = rfl(u ? e1.x : e0.x), ey_ = rfl(u ? e1.y : e0.y);
          const uint32_t j = ex_ & 0xFFu;
          bool ready = ((ex_ >> 9) & 3u) == 1u;          // its other input is valid: go on in registers
          if (!ready) {
            // a join of two stale inputs: both sides leave their result in their slot and count down; whoever finds the
            // other one's count goes on with it (LDS only: writes and the atomic of a wave arrive in order)
            uint32_t *mp = K.stage + (size_t)(ex_ >> 24) * ((R + 1) * 64) + lane;
#pragma unroll
            for (int k = 0; k < R; k++) mp[k * 64] = c.v[k / VW][k % VW];
            mp[R * 64] = lc;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const uint32_t old = wave_fetch_sub(&K.PEND[j], 1u, lane);
            ready = old == 1u;
          }
          if (ready) {
            if (nxt == kNone16) { nxt = j; nx_x = ex_; nx_y = ey_; }
            else {
              // (a vector two ops of the step wait for: the second one becomes a chain start and reads this result from memory)
              __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
              if (lane == 0) {
                const uint32_t slot = __hip_atomic_fetch_add(&sh.rtail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                K.R[slot] = (uint16_t)j;
              }
            }
          }
        }
      }
      if (lane == 0) __hip_atomic_fetch_add(&sh.ndone, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      nlinks++;
      if (nxt == kNone16) break;
      // the next link: this result is one operand, the other one sits in its slot (or in memory: a join, or no slot left)
      const uint32_t which = (nx_x >> 8) & 1u, oslot = (nx_x >> 16) & 0xFFu, ocid = nx_y >> 16;
      i = nxt;
      r = nx_y & 0xFFFFu;
      if (which == 0u) { ta = c; la = lc; operand(tb, lb, oslot, ocid); }
      else { tb = c; lb = lc; operand(ta, la, oslot, ocid); }
    }
    if (prof && lane == 0) { sh.tph[9] += __builtin_amdgcn_s_memrealtime() - ct0; sh.tph[10] += nlinks; sh.tph[11]++; }
  }
}

// the plain dataflow path (closures beyond kLcap ops: the first step of a launch): operands from memory, consumers found through
// the topology and the claim words
template <int KS, int VW>
__device__ __forceinline__ void refresh_dynamic(const Kx<KS, VW> &K, Sh &sh)
{
  const uint32_t n = K.n, epoch = sh.epoch, nops = sh.nops;
  const int lane = K.lane;
  QT<KS, VW> c, ta, tb;
  for (;;) {
    const uint32_t idx = wave_fetch_add(&sh.rhead, 1u, lane);
    uint32_t r = kNone16, spins = 0;
    for (;;) {
      r = *(volatile uint16_t *)&K.R[idx];
      if (r != kNone16) break;
      if (*(volatile uint32_t *)&sh.ndone >= nops) break;
      if (++spins > (1u << 22)) { sh.err = 2u; break; }
      __builtin_amdgcn_s_sleep(1);
    }
    r = rfl(r);
    if (r == kNone16) break;
    if (lane == 0) K.R[idx] = (uint16_t)kNone16;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    uint32_t prev = kNone16, lprev = 0;
    for (uint32_t link = 0;; link++) {
      if (link > K.ns) { if (lane == 0) sh.err = 5u; break; }
      const uint32_t r1 = nxc(r, n);
      const uint32_t a = rfl((uint32_t)K.bk[r1]), b = rfl((uint32_t)K.bk[nxc(r1, n)]);
      uint32_t la, lb;
      if (prev == a) { ta = c; la = lprev; } else { ld<KS, VW>(K, ta, a); la = ld_sl<KS, VW>(K, a); }
      if (prev == b) { tb = c; lb = lprev; } else { ld<KS, VW>(K, tb, b); lb = ld_sl<KS, VW>(K, b); }
      const uint32_t cost = q_fitch<KS, VW>(c, ta, tb);
      const uint32_t lc = cost + la + lb;
      if (K.st_lane) qstore<KS, VW>(c, K.rsrc, K.voff, r * K.SW4);
      st_sl<KS, VW>(K, r, lc);
      if (lane == 0) K.valid[r] = 1;
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
          if ((v >> kEpochShift) == epoch && !vc) {
            bool ready;
            if ((v & 7u) == 1u) {
              ready = true;
            } else {
              __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
              const uint32_t old = wave_fetch_sub(&K.cl[cand], 1u, lane);
              ready = (old & 7u) == 1u;
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
      prev = r; lprev = lc; r = rdy0;
    }
  }
}

// ---- (3') a workgroup that holds SEVERAL tiles (fewer workgroups than tiles; k_climb_many: all of them).  Run tile after tile, the
// cooperative refresh above pays its chain of dependent links -- a memory round trip, LDS hand-overs, two barriers -- once per
// tile, and most waves watch (measured at C3 on 13 tiles of 128 words: 27 us per tile whatever its width, 350 of a step's 800 us).
// Instead: the closure's ops are put into ONE order in which every op comes after its stale inputs (wave 0, once per step), and
// every wave takes whole tiles of its own through that sequence, alone: nothing to wait for, nothing to hand over.  An operand
// that was valid before the step -- or was written at least a block of four ops ago by this very wave: a wave's accesses to an
// address arrive in program order -- is requested with its block's other operands in one go; the result of the op just before
// stays in registers; what is left (the far input of a join, now and then) is read where it is needed.
template <int KS, int VW>
__device__ __forceinline__ void schedule_private(const Kx<KS, VW> &K, Sh &sh)
{
  const uint32_t n = K.n, nops = rfl(sh.nops), nstart = rfl(sh.rtail);
  if (nops <= 64u) {
    // the usual closure: an op per lane, its descriptor, consumers and counts in registers; the walk itself runs on the scalar unit
    // (v_readlane / v_writelane with a scalar lane index) -- a twentieth of a microsecond per op where the LDS form below, one
    // dependent round trip after the other, takes a third
    const int lane = K.lane;
    const bool act = (uint32_t)lane < nops;
    const uint2 d = act ? K.D[lane] : make_uint2(0u, 0u);
    const uint32_t ncl = act ? (K.NC[lane] < 2u ? K.NC[lane] : 2u) : 0u;
    const uint32_t c0 = act ? K.CONS[2 * lane].x : 0u, c1 = act ? K.CONS[2 * lane + 1].x : 0u;
    int pend = act ? (int)K.PEND[lane] : 0;
    const uint32_t a = d.y & 0xFFFFu, b = d.y >> 16;
    uint32_t ia = 0xFFu, ib = 0xFFu;                   // the op that makes a stale operand
    if (act && a >= n && !K.valid[a]) ia = (K.cl[a] >> 3) & 0x3Fu;
    if (act && b >= n && !K.valid[b]) ib = (K.cl[b] >> 3) & 0x3Fu;
    const int meta = (int)(ncl | (ia << 8) | (ib << 16));
    int stackv = (uint32_t)lane < nstart ? (int)K.R[nstart - 1u - (uint32_t)lane] : 0;
    int posv = 0, outx = 0, outy = 0;
    uint32_t sp = nstart, cnt = 0;
    while (sp) {
      sp--;
      const uint32_t i = (uint32_t)__builtin_amdgcn_readlane(stackv, (int)sp);
      const uint32_t dx = (uint32_t)__builtin_amdgcn_readlane((int)d.x, (int)i), dy = (uint32_t)__builtin_amdgcn_readlane((int)d.y, (int)i);
      const uint32_t mt = (uint32_t)__builtin_amdgcn_readlane(meta, (int)i);
      uint32_t fl[2];
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const uint32_t iv = (mt >> (8 + 8 * u)) & 0xFFu;
        uint32_t f = 1u;
        if (iv != 0xFFu) {
          const uint32_t at = (uint32_t)__builtin_amdgcn_readlane(posv, (int)iv);
          f = at + 1u == cnt ? 0u : at < (cnt & ~3u) ? 1u : 2u;
        }
        fl[u] = f;
      }
      posv = wlane((int)cnt, (int)i, posv);
      outx = wlane((int)((dx & 0xFFFFu) | (fl[0] << 16) | (fl[1] << 18)), (int)cnt, outx);
      outy = wlane((int)dy, (int)cnt, outy);
      cnt++;
      if (cnt > nops) break;
      const uint32_t nc = mt & 0xFFu;
      const uint32_t e0 = (uint32_t)__builtin_amdgcn_readlane((int)c0, (int)i), e1 = (uint32_t)__builtin_amdgcn_readlane((int)c1, (int)i);
#pragma unroll
      for (int u = 0; u < 2; u++) {
        if ((uint32_t)u < nc) {
          const uint32_t ex_ = u ? e1 : e0, j = ex_ & 0xFFu;
          bool ready = ((ex_ >> 9) & 3u) == 1u;
          if (!ready) {
            const int left = __builtin_amdgcn_readlane(pend, (int)j) - 1;
            pend = wlane(left, (int)j, pend);
            ready = left == 0;
          }
          if (ready) { stackv = wlane((int)j, (int)sp, stackv); sp++; }
        }
      }
    }
    if (act) K.SD[lane] = make_uint2((uint32_t)outx, (uint32_t)outy);
    if (cnt != nops && lane == 0) sh.err = sh.err ? sh.err : 6u;
    return;
  }
  // (larger closures: one lane, LDS round trips.  The sequence is written beside the consumer entries it is made from -- into the
  //  region, idle between the enumeration's list and the scans -- and moved over them at the end)
  uint2 *seq = reinterpret_cast<uint2 *>(K.stage);
  if (K.lane == 0) {
  uint32_t sp = 0, cnt = 0;
  for (uint32_t k = 0; k < nstart; k++) K.R0[sp++] = K.R[nstart - 1u - k];
  while (sp) {
    const uint32_t i = K.R0[--sp];
    const uint2 d = K.D[i];
    const uint32_t a = d.y & 0xFFFFu, b = d.y >> 16;
    uint32_t fl[2];
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const uint32_t v = u ? b : a;
      uint32_t f = 1u;                                   // from memory, with the block's requests
      if (v >= n && !K.valid[v]) {
        const uint32_t at = K.PEND0[(K.cl[v] >> 3) & kIdxMask];      // where its op stands in the sequence (it does: it came first)
        f = at + 1u == cnt ? 0u : at < (cnt & ~3u) ? 1u : 2u;
      }
      fl[u] = f;
    }
    K.PEND0[i] = cnt;
    seq[cnt] = make_uint2((d.x & 0xFFFFu) | (fl[0] << 16) | (fl[1] << 18), d.y);
    cnt++;
    if (cnt > nops) break;
    const uint32_t nc = K.NC[i] < 2u ? K.NC[i] : 2u;
    for (uint32_t u = 0; u < nc; u++) {
      const uint32_t ex_ = K.CONS[2u * i + u].x, j = ex_ & 0xFFu;
      bool ready = ((ex_ >> 9) & 3u) == 1u;
      if (!ready) { const uint32_t left = K.PEND[j] - 1u; K.PEND[j] = left; ready = left == 0u; }
      if (ready) K.R0[sp++] = (uint16_t)j;
    }
  }
  if (cnt != nops) sh.err = sh.err ? sh.err : 6u;
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  for (uint32_t k = (uint32_t)K.lane; k < nops; k += 64u) K.SD[k] = seq[k];
}

template <int KS, int VW>
__device__ __forceinline__ void refresh_private(const Kx<KS, VW> &K, Sh &sh)
{
  const uint32_t nops = sh.nops;
  const int lane = K.lane;
  QT<KS, VW> c;
  uint32_t lc = 0;
  for (uint32_t k0 = 0; k0 < nops; k0 += 4u) {
    QT<KS, VW> A[4], B[4];
    uint32_t la[4], lb[4], sx[4], sy[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const uint32_t k = k0 + (uint32_t)i;
      const uint2 sd = K.SD[k < nops ? k : nops - 1u];
      sx[i] = rfl(sd.x); sy[i] = rfl(sd.y);
      la[i] = lb[i] = 0;
      if (k < nops) {
        if (((sx[i] >> 16) & 3u) == 1u) { ld<KS, VW>(K, A[i], sy[i] & 0xFFFFu); la[i] = ld_sl<KS, VW>(K, sy[i] & 0xFFFFu); }
        if (((sx[i] >> 18) & 3u) == 1u) { ld<KS, VW>(K, B[i], sy[i] >> 16); lb[i] = ld_sl<KS, VW>(K, sy[i] >> 16); }
      }
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const uint32_t k = k0 + (uint32_t)i;
      if (k < nops) {
        const uint32_t fa = (sx[i] >> 16) & 3u, fb = (sx[i] >> 18) & 3u, r = sx[i] & 0xFFFFu;
        if (fa == 0u) { A[i] = c; la[i] = lc; }
        else if (fa == 2u) { ld<KS, VW>(K, A[i], sy[i] & 0xFFFFu); la[i] = ld_sl<KS, VW>(K, sy[i] & 0xFFFFu); }
        if (fb == 0u) { B[i] = c; lb[i] = lc; }
        else if (fb == 2u) { ld<KS, VW>(K, B[i], sy[i] >> 16); lb[i] = ld_sl<KS, VW>(K, sy[i] >> 16); }
        QT<KS, VW> o;
        const uint32_t cost = q_fitch<KS, VW>(o, A[i], B[i]);
        c = o;
        lc = cost + la[i] + lb[i];
        if (K.st_lane) qstore<KS, VW>(c, K.rsrc, K.voff, r * K.SW4);
        st_sl<KS, VW>(K, r, lc);
        if (lane == 0) K.valid[r] = 1;
      }
    }
  }
}

template <int KS, int VW>
__device__ __forceinline__ void refresh(const Kx<KS, VW> &K, Sh &sh, bool prof, const uint32_t nw)
{
  if (sh.nops == 0) return;
  if (sh.use_static) refresh_static<KS, VW>(K, sh, prof, nw);
  else refresh_dynamic<KS, VW>(K, sh);
}

// ---- (4) one gap end of a scan: the program of (1) run front to back -- addTraverseParsimony + testInsertParsimony
// (:2208-2218, :2106-2160), both children of a node expanded together (one vector read per insertion test).  The child
// vectors of PF expansions are requested at once, so a part pays for one memory round trip per PF expansions; the
// up-vector runs in registers from an expansion to its first child, second children find theirs in LDS.
template <int KS, int VW>
__device__ __forceinline__ void scan_part(const Kx<KS, VW> &K, Sh &sh, uint32_t pi, uint32_t half)
{
  constexpr int R = Cfg<KS, VW>::R;
  constexpr int PF = Cfg<KS, VW>::PF;
  const int lane = K.lane;
  const Unit un = sh.unit[pi >> 1];
  const uint32_t e = pi & 1u;
  const uint32_t other = rfl((uint32_t)(e ? un.xa : un.xb));
  const uint32_t pe = rfl(sh.pE[pi]), E = pe & 0xFFu, EL = pe >> 8;
  // a long program is shared by two waves: both run the root entry, one goes on below its first child, the other one below
  // the second (whose up-vector the root entry parks like any second child's)
  const bool split = E >= 6u;
  const uint32_t lo = !split ? 1u : half ? 1u + EL : 1u, hi = !split ? E : half ? E : 1u + EL;
  const uint32_t root_mask = !split ? 3u : half ? 2u : 1u;       // which of the root's two candidates this wave books
  // this lane's address of candidate 0 (quad leaders inside the row add their counts, everybody else adds nothing)
  uint32_t *cbase = K.cost + rfl(sh.poff[pi]);
  const uint2 *prog = K.prog + pi * kProgStride;
  uint32_t *pend = K.pend + (size_t)K.wave * (5 * R * 64) + lane;     // slots for child depths 1..5
  QT<KS, VW> sv, par, u1, u2, d1[PF], d2[PF];
  uint32_t ex[PF], ey[PF];
  ld<KS, VW>(K, sv, rfl((uint32_t)un.s));
  ld<KS, VW>(K, par, other);
  // entry 0 first, then [lo, hi)
  const uint32_t n_run = 1u + (hi - lo);
  for (uint32_t blk = 0; blk < n_run; blk += (uint32_t)PF) {
    uint2 en[PF];
#pragma unroll
    for (int i = 0; i < PF; i++) {
      uint32_t q = blk + (uint32_t)i;
      q = q < n_run ? q : n_run - 1u;                                 // (clamped: the requests stay unconditional)
      en[i] = prog[q == 0u ? 0u : lo + q - 1u];
    }
#pragma unroll
    for (int i = 0; i < PF; i++) {
      ex[i] = rfl(en[i].x);
      ey[i] = rfl(en[i].y);
      ld<KS, VW>(K, d1[i], ex[i] & 0xFFFFu);
      ld<KS, VW>(K, d2[i], ex[i] >> 16);
    }
#pragma unroll
    for (int i = 0; i < PF; i++) {
      const uint32_t q = blk + (uint32_t)i;
      if (q < n_run) {
        const uint32_t y = ey[i], dd = y & 15u;
        if (q > 0u) {
          if (y & 64u) {
#pragma unroll
            for (int k = 0; k < R; k++) par.v[k / VW][k % VW] = pend[((dd - 2u) * R + (uint32_t)k) * 64u];
          } else {
            par = u1;
          }
        }
        q_fitch<KS, VW>(u1, par, d2[i]);
        q_fitch<KS, VW>(u2, par, d1[i]);
        if (y & 16u) {
          const uint32_t m = q == 0u ? root_mask : 3u;
          // counts leave as one LDS add per candidate from the 16 quad leaders (no reduction across lanes, nothing to wait for)
          const uint32_t j1 = (m & 1u) ? q_join<KS, VW>(u1, d1[i], sv) : 0u;      // (all lanes: the quad ORs need their neighbours)
          const uint32_t j2 = (m & 2u) ? q_join<KS, VW>(u2, d2[i], sv) : 0u;
          if constexpr (kWordMajor<KS>) {
            // (64 counting lanes: summed across the wave first -- 64 adds to one LDS word would queue up)
            const uint32_t t1 = (m & 1u) ? wave_total(K.cnt_lane ? j1 : 0u) : 0u, t2 = (m & 2u) ? wave_total(K.cnt_lane ? j2 : 0u) : 0u;
            if (lane == 0) {
              if (m & 1u) __hip_atomic_fetch_add(cbase + ((y >> 8) & 0xFFu), t1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              if (m & 2u) __hip_atomic_fetch_add(cbase + ((y >> 16) & 0xFFu), t2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
          } else
          if (K.cnt_lane) {
            if (m & 1u) __hip_atomic_fetch_add(cbase + ((y >> 8) & 0xFFu) + K.zero, j1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (m & 2u) __hip_atomic_fetch_add(cbase + ((y >> 16) & 0xFFu) + K.zero, j2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        }
        if (y & 32u) {
#pragma unroll
          for (int k = 0; k < R; k++) pend[((dd - 1u) * R + (uint32_t)k) * 64u] = u2.v[k / VW][k % VW];
        }
      }
    }
  }
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
  bool moved = false, cut = false;
  const uint32_t stop_len = P.stop_len;
  uint32_t j = 0;
  for (; j < Beff && !moved; j++) {
    const uint32_t off = sh.pn_off[j], nt = sh.pn_cnt[j], np = sh.pn_np[j], pcid = sh.pn_p[j];
    if (stop_len) {
      // a tree the tracker would have to book: hand back in front of this prune node, nothing of it consumed
      unsigned long long any = 0ull;
      for (uint32_t base = 0; base < nt; base += 64u) {
        const uint32_t ci = base + (uint32_t)lane;
        any |= __ballot((int)((ci < nt ? K.cost[off + ci] : 0xFFFFFFFFu) <= stop_len));
      }
      if (any) { cut = true; break; }
    }
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
      // the insertion branch: named by the enumeration, per part and candidate index
      uint32_t part = 4u * j;
      while (part < 4u * j + 3u && off + (uint32_t)sel >= sh.poff[part] + sh.pcnt[part]) part++;
      ins = (int32_t)K.cq[part * 128u + (off + (uint32_t)sel - sh.poff[part])];
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
    // (the invalidation this edit causes is walked after the barrier behind this function, beside the next step's enumeration:
    //  invalidate_walk)
    if (lane == 0) { sh.inv5[0] = a; sh.inv5[1] = b; sh.inv5[2] = p; sh.inv5[3] = q; sh.inv5[4] = r; sh.einv = sh.epoch + 1u; }
  }
  if (lane == 0) {
    sh.moved = moved ? 1u : 0u;
    if (sh.ncand) { sh.last_ncand[sh.xgen % 3u] = sh.ncand; sh.xgen++; }
    sh.best = best; sh.randomMP = randomMP; sh.iter_hits = iter_hits;
    sh.rng = rng; sh.hits = hits; sh.draws = draws; sh.ins = ins; sh.rem = rem;
    sh.n_tests += tests; sh.n_nodes += consumed;
    sh.n_moves = n_moves;
    sh.consumed = consumed;
    const uint32_t pos = sh.pos + consumed;
    sh.pos = pos;
    sh.steps++;
    sh.epoch += 2u;
    const uint32_t gap = sh.since_move + consumed;      // prune nodes since the move before (this step's included)
    const uint32_t since = moved ? 0u : gap;
    sh.since_move = since;
    // a batch is wasted behind the first accepted move; after a step without one the next looks twice as far ahead -- and after a
    // move that was sixteen or more prune nodes away half as far as that (moves come in stretches of similar density: a sweep near an
    // optimum does not start from two prune nodes again after each of its rare moves)
    // (round 6) ... and after a nearer one about three quarters of the recent average distance between moves: a climb from a
    // random tree moves at every first or second prune node (batch_min it is), one from a tree perturbed by 498 NNIs at every
    // sixth, where restarting from two prune nodes cost a step in three for nothing (C3: 798 -> ~620 steps per such climb)
    // (round 5 took this distance AFTER the counter had been reset: it never saw more than the step's own prune nodes)
    if (moved) sh.gap_ema = (3u * sh.gap_ema + 8u * (gap > 64u ? 64u : gap) + 2u) >> 2;
    uint32_t b_near = (sh.gap_ema * P.near_q / 4u + 4u) >> 3;
    b_near = b_near < P.batch_min ? P.batch_min : b_near;
    uint32_t B = moved ? (gap >= 16u ? gap / 2u : b_near) : sh.B * 2u;
    B = B > P.batch_max ? P.batch_max : B;
    B = B < 1u ? 1u : B;
    if (pos <= P.total && B > P.total - pos + 1u) B = P.total - pos + 1u;      // (not beyond the end of the sweep)
    sh.B = B;
    // the next step's counters (its enumeration starts right behind the barrier that follows)
    sh.rtail = 0; sh.rhead = 0; sh.ndone = 0; sh.nops = 0; sh.task = 0; sh.qtail = 0;
    uint32_t reason = CLIMB_RUNNING;
    if (pos > P.total) reason = CLIMB_SWEEP_END;
    else if (n_moves >= P.max_moves || sh.epoch > kEpochLimit) reason = CLIMB_MOVES_FULL;
    else if (P.idle_limit && since >= P.idle_limit) reason = CLIMB_IDLE;
    if (cut) reason = CLIMB_CUTOFF;
    if ((consumed == 0u && !cut) || sh.steps > 4u * P.total * (sh.sweeps + 1u) + 16u) sh.err = sh.err ? sh.err : 7u;
    if (sh.err) reason = CLIMB_ERROR;
    sh.exit_reason = reason;
  }
}

// ---- (6b) wave 0, beside the next step's enumeration: every vector whose subtree contains an edited node is stale -- the three
// of each edited node and, walking outwards, the two outward-looking ones of every node reached, as far as they were valid.
// (The enumeration stamps claim words with the NEXT epoch meanwhile; the stamp of this walk only saves second visits, validity
//  decides.)
template <int KS, int VW>
__device__ __forceinline__ void invalidate_walk(const Kx<KS, VW> &K, Sh &sh)
{
  const int lane = K.lane;
  const uint32_t n = K.n, einv = sh.einv;
  uint32_t five = sh.inv5[0];
  five = lane / 3 == 1 ? sh.inv5[1] : five;
  five = lane / 3 == 2 ? sh.inv5[2] : five;
  five = lane / 3 == 3 ? sh.inv5[3] : five;
  five = lane / 3 == 4 ? sh.inv5[4] : five;
  bool have = false;
  uint32_t item = 0;
  if (lane < 15 && five >= n) {
    const uint32_t rec = five - (five - n) % 3u + (uint32_t)(lane % 3);
    K.valid[rec] = 0;
    const uint32_t w = K.bk[rec];
    if (w >= n) { have = true; item = w; }
  }
  uint32_t head = 0;
  uint32_t round = 0;
  if (lane == 0) sh.wtail = 0;
  for (;; round++) {
    if (round > K.ns) { if (lane == 0) sh.err = 4u; break; }
    const uint32_t tail = *(volatile uint32_t *)&sh.wtail;
    {
      const unsigned long long need = __ballot((int)!have);
      const uint32_t rank = (uint32_t)__builtin_popcountll(need & ((1ull << lane) - 1ull));
      const uint32_t avail = tail - head;
      if (!have && rank < avail) { item = K.W[head + rank]; have = true; }
      const uint32_t want = (uint32_t)__builtin_popcountll(need);
      head += want < avail ? want : avail;
    }
    if (!__ballot((int)have)) break;
#pragma unroll 1
    for (int step = 0; step < 4; step++) {
      if (have) {
        uint32_t o1, o2;
        ring2(item, n, o1, o2);
        // (one LDS round trip per link: claims, validity and the records behind go out together)
        const uint32_t c1 = __hip_atomic_fetch_max(&K.cl[o1], einv << kEpochShift, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const uint32_t c2 = __hip_atomic_fetch_max(&K.cl[o2], einv << kEpochShift, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const uint32_t x1 = K.valid[o1], x2 = K.valid[o2];
        const uint32_t u1 = K.bk[o1], u2 = K.bk[o2];
        const bool v1 = ((c1 >> kEpochShift) != einv) & (x1 != 0u), v2 = ((c2 >> kEpochShift) != einv) & (x2 != 0u);
        if (v1) K.valid[o1] = 0;
        if (v2) K.valid[o2] = 0;
        const bool g1 = v1 & (u1 >= n), g2 = v2 & (u2 >= n);
        if (g1) {
          item = u1;
          if (g2) {
            const uint32_t slot = __hip_atomic_fetch_add(&sh.wtail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            K.W[slot] = (uint16_t)u2;
          }
        } else if (g2) item = u2;
        else have = false;
      }
    }
  }
  if (lane == 0) sh.c_inv += round;
}

// everything in a workgroup's LDS but the stage / pend region (the kernel carves in this order)
// (maxb = ClimbParams::batch_max: a step has at most 4 * maxb scan programs -- their entries and candidate lists are the largest
//  arrays here, 48 KB at sixteen prune nodes per step, 24 KB at the plain climb's eight)
__host__ __device__ inline size_t lds_fixed_bytes(uint32_t ns, uint32_t maxb)
{
  const size_t parts = 4 * (size_t)(maxb < 1u ? 1u : maxb > (uint32_t)kMaxB ? (uint32_t)kMaxB : maxb);
  size_t at = (sizeof(Sh) + 15) & ~(size_t)15;
  at += (((size_t)ns * 4) + 15) & ~(size_t)15;
  at += parts * kProgStride * sizeof(uint2);
  at += (size_t)kLcap * sizeof(uint2);
  at += (size_t)kLcap * 2 * sizeof(uint2);
  at += (size_t)kLcap * 4;
  at += (size_t)kLcap * 4;
  at += (size_t)kClimbCap * 4;
  at += (((size_t)ns * 2) + 15) & ~(size_t)15;
  at += (((size_t)ns * 2) + 15) & ~(size_t)15;
  at += (((size_t)(ns + 16) * 2) + 15) & ~(size_t)15;
  at += parts * 128 * 2;
  at += (size_t)kLcap * 2;
  at += (((size_t)(ns / 2 + 1) * 2) + 15) & ~(size_t)15;      // the visiting order: 2n - 2 entries
  at += ns;
  at = (at + 15) & ~(size_t)15;
  at += (size_t)kLcap * 4 + (size_t)kLcap * 2;               // PEND0, R0 (multi-tile workgroups)
  at += (size_t)kLcap * 2;                                    // TF (k_climb_many: a tile's refresh of this step is through)
  return (at + 15) & ~(size_t)15;
}

// the stage / pend region of a launch: what the budget leaves, between the configuration's bounds (host and kernel agree by
// computing it from the same number of vector slots)
template <int KS, int VW>
__host__ __device__ inline size_t region_bytes(uint32_t ns, uint32_t maxb, int nw)
{
  const size_t fixed = lds_fixed_bytes(ns, maxb);
  size_t r = kLdsBudget > fixed ? (kLdsBudget - fixed) & ~(size_t)15 : 0;
  r = r > Cfg<KS, VW>::region_max(nw) ? Cfg<KS, VW>::region_max(nw) : r;
  r = r < Cfg<KS, VW>::region_min(nw) ? Cfg<KS, VW>::region_min(nw) : r;
  return r;
}

// The body of a climb's workgroup: workgroup `tile` of the climb's T (k_climb: the launch's grid; k_climb_many: ONE workgroup per
// climb, every workgroup of the launch another climb with its own ClimbParams).
// MODE 0: a workgroup per tile (k_climb as launched by default: nothing of the several-tiles machinery is compiled in -- its
// registers are the sixteen-wave shapes' scarce resource); 1: fewer workgroups than tiles, tile after tile (option climb_groups);
// 2: k_climb_many (the waves take tiles of their own through the refresh: refresh_private)
template <int KS, int VW, int MODE>
__device__ __forceinline__ void climb_body(const ClimbParams &P, const uint32_t tile, const uint32_t T)
{
  constexpr uint32_t kNW = MODE == 2 ? Cfg<KS, VW>::NW_MANY : Cfg<KS, VW>::NW, kThreads = kNW * 64u;
  extern __shared__ __align__(16) unsigned char smem[];
  const int tid = (int)threadIdx.x, lane = tid & 63;
  const int wave = (int)rfl((uint32_t)(tid >> 6));
  // workgroup `tile` of T; it owns the tiles tile, tile + T, ... of the alignment's TT (T == TT: exactly one)
  const uint32_t n = P.n, ns = P.nslots, TT = P.tiles;
  constexpr bool PRIV = MODE == 2;
  const uint32_t nmine = MODE == 0 ? 1u : tile < TT ? (TT - tile + T - 1u) / T : 0u;
  // ---- carve the workgroup's LDS
  Sh &sh = *reinterpret_cast<Sh *>(smem);
  size_t at = (sizeof(Sh) + 15) & ~(size_t)15;
  Kx<KS, VW> K;
  K.cl = reinterpret_cast<uint32_t *>(smem + at); at += (((size_t)ns * 4) + 15) & ~(size_t)15;
  const uint32_t maxb = P.batch_max < 1u ? 1u : P.batch_max > (uint32_t)kMaxB ? (uint32_t)kMaxB : P.batch_max;
  K.prog = reinterpret_cast<uint2 *>(smem + at); at += (size_t)(4u * maxb) * kProgStride * sizeof(uint2);
  K.stage = reinterpret_cast<uint32_t *>(smem + at);
  K.pend = reinterpret_cast<uint32_t *>(smem + at);
  K.Q = reinterpret_cast<uint16_t *>(smem + at); at += region_bytes<KS, VW>(ns, maxb, (int)kNW);       // (ns entries of 2 bytes: fits the region for every ns the other arrays allow)
  K.D = reinterpret_cast<uint2 *>(smem + at); at += (size_t)kLcap * sizeof(uint2);
  K.CONS = reinterpret_cast<uint2 *>(smem + at); at += (size_t)kLcap * 2 * sizeof(uint2);
  K.NC = reinterpret_cast<uint32_t *>(smem + at); at += (size_t)kLcap * 4;
  K.PEND = reinterpret_cast<uint32_t *>(smem + at); at += (size_t)kLcap * 4;
  K.cost = reinterpret_cast<uint32_t *>(smem + at); at += (size_t)kClimbCap * 4;
  K.bk = reinterpret_cast<uint16_t *>(smem + at); at += (((size_t)ns * 2) + 15) & ~(size_t)15;
  K.W = reinterpret_cast<uint16_t *>(smem + at); at += (((size_t)ns * 2) + 15) & ~(size_t)15;
  K.R = reinterpret_cast<uint16_t *>(smem + at); at += (((size_t)(ns + 16) * 2) + 15) & ~(size_t)15;
  K.cq = reinterpret_cast<uint16_t *>(smem + at); at += (size_t)(4u * maxb) * 128 * 2;
  K.OL = reinterpret_cast<uint16_t *>(smem + at); at += (size_t)kLcap * 2;
  K.ord = reinterpret_cast<uint16_t *>(smem + at); at += (((size_t)P.total * 2) + 15) & ~(size_t)15;
  K.valid = reinterpret_cast<uint8_t *>(smem + at); at = (at + ns + 15) & ~(size_t)15;
  K.PEND0 = reinterpret_cast<uint32_t *>(smem + at); at += (size_t)kLcap * 4;
  K.R0 = reinterpret_cast<uint16_t *>(smem + at); at += (size_t)kLcap * 2;
  K.TF = reinterpret_cast<uint16_t *>(smem + at);
  K.SD = K.CONS;                                        // (the consumer entries are done with once the sequence stands; the region is not free: early waves' scans park vectors there while late waves still refresh)
  K.n = n; K.ns = ns; K.lane = lane; K.wave = wave;
  { const size_t sl = region_bytes<KS, VW>(ns, maxb, (int)kNW) / Cfg<KS, VW>::kSlotBytes; K.slots = sl < 254 ? (uint32_t)sl : 254u; }
  K.SW4 = (uint32_t)(kWordMajor<KS> ? 4 : 4 * KS) * P.Wp * 4u;
  K.svoff = kWordMajor<KS> ? (uint32_t)lane * 4u : ((uint32_t)lane >> 2) * 4u;
  asm volatile("v_mov_b32 %0, 0" : "=v"(K.zero));
  K.rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)P.vec, 0, 0x7FFFFFFF, 0x00020000);
  // what depends on the tile of sites at hand: its rows of per-lane subtree scores, the lanes' offsets into a vector, which lanes count
  auto set_tile = [&](uint32_t t) {
    constexpr uint32_t kRow = kWordMajor<KS> ? 64u : 16u;      // score words per vector and tile
    K.rsrc_s = __builtin_amdgcn_make_buffer_rsrc((void *)(P.sct + (size_t)t * ns * kRow), 0, (int)(ns * kRow * 4u), 0x00020000);
    if constexpr (kWordMajor<KS>) {
      // a lane = VW words with their four states (rows of the store: state k at k * Wp)
      uint32_t word0 = (t * 64u + (uint32_t)lane) * (uint32_t)VW;
      K.st_lane = word0 < P.Wp;
      if (!K.st_lane) word0 = P.Wp - (uint32_t)VW;
      K.cnt_lane = K.st_lane;
#pragma unroll
      for (int k = 0; k < KS; k++) K.voff[k] = ((uint32_t)k * P.Wp + word0) * 4u;
    } else {
    const uint32_t w = (uint32_t)lane >> 2, g = (uint32_t)lane & 3u;
    uint32_t word0 = (t * 16u + w) * (uint32_t)VW;
    K.st_lane = word0 < P.Wp;
    if (!K.st_lane) word0 = P.Wp - (uint32_t)VW;         // lanes past the row end load real data and contribute nothing
    K.cnt_lane = K.st_lane && g == 0u;
#pragma unroll
    for (int k = 0; k < KS; k++) K.voff[k] = ((g * (uint32_t)KS + (uint32_t)k) * P.Wp + word0) * 4u;
    }
  };
  set_tile(tile);
  {
    // heap-index relations of lane h (complete binary tree, root 1): who comes before h in pre-order, which ancestors hold h
    // in their LEFT subtree, who sits in the subtree of h's left child
    const uint32_t h = (uint32_t)lane;
    unsigned long long pre = 0, ancl = 0, lsub = 0;
    if (h >= 1u) {
      const int dh = 31 - __builtin_clz(h);
      const uint32_t hh = h << (5 - dh);
      for (uint32_t g = 1; g < 64u; g++) {
        const int dg = 31 - __builtin_clz(g);
        const uint32_t gg = g << (5 - dg);
        if (g != h && (gg < hh || (gg == hh && dg < dh))) pre |= 1ull << g;
        if (dg < dh && (h >> (dh - dg)) == g && !((h >> (dh - dg - 1)) & 1u)) ancl |= 1ull << g;
        if (dg > dh && (g >> (dg - dh - 1)) == 2u * h) lsub |= 1ull << g;
      }
    }
    K.pre = pre; K.ancl = ancl; K.lsub = lsub;
  }
  // ---- the launch's state: topology, all inner vectors stale (the kernel keeps its own per-tile subtree scores)
  for (uint32_t i = (uint32_t)tid; i < ns; i += kThreads) {
    K.bk[i] = P.bk[i];
    K.valid[i] = i < n ? 1 : 0;
    K.cl[i] = 0u;
  }
  for (uint32_t i = (uint32_t)tid; i < ns + 16u; i += kThreads) K.R[i] = (uint16_t)kNone16;
  for (uint32_t i = (uint32_t)tid; i < kLcap; i += kThreads) K.TF[i] = 0;
  for (uint32_t i = (uint32_t)tid; i < P.total; i += kThreads) K.ord[i] = P.order[i];
  if (tid == 0) {
    const ClimbHeader h = *P.hdr;
    sh.pos = h.pos; sh.B = h.batch ? h.batch : P.batch_min; sh.epoch = 1u; sh.exit_reason = CLIMB_RUNNING;
    sh.gap_ema = 8u * P.batch_min;
    sh.startMP = h.start_mp; sh.sweeps = 0;
    sh.since_move = h.since_move; sh.rtail = 0; sh.steps = 0; sh.xgen = 0; sh.n_moves = 0; sh.err = 0; sh.trace_n = 0;
    sh.last_ncand[0] = sh.last_ncand[1] = sh.last_ncand[2] = 0;
    sh.best = h.best; sh.randomMP = h.randomMP; sh.iter_hits = h.iter_hits; sh.ins = h.insert_cid; sh.rem = h.remove_cid;
    sh.rng = h.rng; sh.hits = h.hits; sh.n_tests = 0; sh.n_ops = 0; sh.draws = 0; sh.n_nodes = 0;
    if (sh.B > maxb) sh.B = maxb;
    if (sh.pos <= P.total && sh.B > P.total - sh.pos + 1u) sh.B = P.total - sh.pos + 1u;
    sh.wtail = 0; sh.rhead = 0; sh.ndone = 0; sh.nops = 0; sh.task = 0; sh.qtail = 0; sh.moved = 0; sh.einv = 0;
    for (int i = 0; i < 16; i++) sh.tph[i] = 0;
    sh.c_rounds = sh.c_inv = sh.c_chains = sh.c_parts = sh.c_inv2 = sh.c_dynops = 0;
    sh.clk0 = __builtin_amdgcn_s_memtime(); sh.rt0 = sh.tlast;
    sh.tlast = __builtin_amdgcn_s_memrealtime();
    // every workgroup must be resident before anyone waits for anyone: arrive, then wait for the others -- not for ever
    // (exchange groups: workgroup b belongs to group b % 8 -- the workgroups the dispatcher deals to one XCD, so a group's
    //  level-1 words normally live in one L2; membership is by NUMBER, not by where a workgroup happens to run: a preempted
    //  workgroup may come back on another XCD, and the protocol must not care)
    sh.xcc = tile & 7u;
    __hip_atomic_fetch_add(&P.hdr->arrive, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    // go or abort is ONE word decided by ONE compare-and-swap: whoever sees everybody arrived proposes "go", whoever runs out of
    // time proposes "abort", the first proposal stands and everybody -- also a workgroup that only becomes resident later --
    // follows it (round 3 let a time-out and the last arrival race: some tiles left, the others waited for them in the exchange)
    uint32_t gate = 0;
    for (;;) {
      gate = __hip_atomic_load(&P.hdr->start_gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (gate) break;
      uint32_t want = 0;
      if (P.fault == 0xFFFFFFFFu) want = 2u;
      else if (__hip_atomic_load(&P.hdr->arrive, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= T) want = 1u;
      else if (__builtin_amdgcn_s_memrealtime() - t0 > 30ull * 100000ull) want = 2u;      // 30 ms of the 100 MHz clock
      if (want) {
        uint32_t expect = 0;
        __hip_atomic_compare_exchange_strong(&P.hdr->start_gate, &expect, want, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        continue;
      }
      __builtin_amdgcn_s_sleep(8);
    }
    const uint32_t ok = gate == 1u ? 1u : 0u;
    sh.ok = ok;
    sh.xm = (T >> 3) + ((tile & 7u) < (T & 7u) ? 1u : 0u);          // workgroups b < T with b % 8 == tile % 8
  }
  __syncthreads();
  if (!sh.ok) {
    if (tile == 0 && tid == 0) P.hdr->reason = CLIMB_ABORT;
    return;
  }

#define MPF_TMARK(i) do { if (tile == 0 && tid == 0) { const unsigned long long now_ = __builtin_amdgcn_s_memrealtime(); sh.tph[i] += now_ - sh.tlast; sh.tlast = now_; } } while (0)
  bool first = true;
  for (;;) {
    // ---- (1) the step's enumeration.  First step of the launch: all waves, right here.  Later steps: it ran behind decide_select
    // of the step before (bottom of the loop), on waves 1.. beside wave 0's invalidation walk.
    if (first) {
      if ((uint32_t)tid < kLcap) K.NC[tid] = 0u;
      for (uint32_t i = (uint32_t)tid; i < kClimbCap; i += kThreads) K.cost[i] = 0u;
      for (uint32_t u = (uint32_t)wave; u < 2u * sh.B; u += kNW) enum_unit<KS, VW>(K, sh, P, u);
      first = false;
    }
    __syncthreads();
    const uint32_t B = sh.B;
    beat(P, tile, tid, 0, sh.steps); beat(P, tile, tid, 2, sh.pos); beat(P, tile, tid, 3, B); beat(P, tile, tid, 1, 1);
    MPF_TMARK(1);
    // the stale ones among the vectors the scans read: the closure's work list (validity is final now: the walk is over)
    filter_required<KS, VW>(K, sh, tid, (int)kThreads);
    __syncthreads();
    beat(P, tile, tid, 1, 2);
    MPF_TMARK(0);
    // ---- (2)
    if (wave == 0) {
      plan_and_discover<KS, VW>(K, sh, PRIV && nmine > 1u);
      if constexpr (PRIV) {
        if (nmine > 1u && sh.use_static) {
          __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
          const unsigned long long ts0 = __builtin_amdgcn_s_memrealtime();
          schedule_private<KS, VW>(K, sh);
          if (lane == 0) sh.tph[15] += __builtin_amdgcn_s_memrealtime() - ts0;
        }
      }
    }
    __syncthreads();
    beat(P, tile, tid, 4, sh.ncand); beat(P, tile, tid, 5, sh.nops); beat(P, tile, tid, 6, sh.rtail); beat(P, tile, tid, 1, 3);
    MPF_TMARK(2);
    // ---- (3) + (4), once per tile of this workgroup.  One tile (the usual launch: a workgroup per tile): as before.  Several:
    // the SAME refresh and the SAME scans run on the next tile's words of every vector -- what a refresh consumes of the closure's
    // lists (PEND's counts, the ready list's appended chain starts; on the plain dataflow path the claim words' counts and the
    // validity flags it raises) is put back in front of every further run, the candidates' counts keep adding up in K.cost, the
    // base lengths in sh.pn_base.
    const uint32_t ncand = sh.ncand;
    const bool multi = MODE != 0 && nmine > 1u;
    const uint32_t nops_step = sh.nops, rtail0 = sh.rtail, use_static = sh.use_static;
    uint32_t *snap = P.snap ? P.snap + (size_t)tile * ((size_t)ns + ns / 4u + 1u) : nullptr;
    // several tiles, and the step's closure has its sequence (or is empty): the waves take tiles of their own through the refresh,
    // then the scans of all tiles are dealt out together (no barrier between one tile and the next)
    // (k_climb_many only: k_climb's sixteen-wave shapes have no registers to spare for it)
    const bool priv = PRIV && multi && (use_static || nops_step == 0u);
    if ((uint32_t)tid < kMaxB) sh.pn_base[tid] = 0u;
    const unsigned long long tsec0 = (multi && !priv && tid == 0) ? __builtin_amdgcn_s_memrealtime() : 0ull;
    if constexpr (PRIV) if (priv) {
      __syncthreads();
      for (uint32_t tk = (uint32_t)wave; tk < nmine; tk += kNW) {
        set_tile(tile + tk * T);
        refresh_private<KS, VW>(K, sh);
        // this tile's share of the two sides of every prune branch (per-lane scores summed over the word groups)
        for (uint32_t j0 = 0; j0 < sh.Beff; j0 += 8u) {
          uint32_t bv[8];
#pragma unroll
          for (int j = 0; j < 8; j++) {
            bv[j] = 0;
            if (j0 + (uint32_t)j < sh.Beff) {
              const uint32_t p = rfl(sh.pn_p[j0 + (uint32_t)j]), q = rfl((uint32_t)K.bk[p]);
              bv[j] = ld_sl<KS, VW>(K, p) + ld_sl<KS, VW>(K, q);
            }
          }
#pragma unroll
          for (int j = 0; j < 8; j++) {
            if (j0 + (uint32_t)j < sh.Beff) {
              const uint32_t tot = wave_total(K.cnt_lane ? bv[j] : 0u);
              if (lane == 0) __hip_atomic_fetch_add(&sh.pn_base[j0 + (uint32_t)j], tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
          }
        }
        // this tile's vectors are as the step needs them: whoever scans it may start (the scans of ALL tiles are one queue, below)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) *(volatile uint16_t *)&K.TF[tk < kLcap ? tk : kLcap - 1u] = (uint16_t)sh.epoch;
      }
      if (nmine > kLcap) __syncthreads();                 // (more tiles than flags: everybody waits for everybody, as before)
      MPF_TMARK(3);
      const uint32_t ntasks = sh.ntasks, items = nmine * ntasks;
      // (the same items as one stream of blocks -- the next block's vectors requested while the one at hand is combined -- was
      //  built and measured slower, 1 470 against 1 290 ms of a C3 climb; HISTORY.md)
      // The (tile, program) items are taken off ONE counter in tile order: a wave whose share of the refresh took a round longer (25
      // tiles on twelve waves: one wave has three, the others two) simply arrives later and takes fewer -- no barrier between the
      // refresh and the scans, only the flag of the tile an item belongs to.
      uint32_t cur = 0xFFFFFFFFu;
      const uint32_t want = sh.epoch & 0xFFFFu;
      for (;;) {
        const uint32_t it = wave_fetch_add(&sh.task, 1u, lane);
        if (it >= items) break;
        const uint32_t tk = it / ntasks, ti = it - tk * ntasks;
        if (tk != cur) {
          if (nmine <= kLcap) {
            uint32_t spins = 0;
            while (rfl((uint32_t) * (volatile uint16_t *)&K.TF[tk]) != want) {
              if (++spins > (1u << 22)) { sh.err = sh.err ? sh.err : 8u; break; }      // (bounded like every wait in this kernel)
              __builtin_amdgcn_s_sleep(1);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
          }
          set_tile(tile + tk * T);
          cur = tk;
        }
        const uint32_t t = rfl((uint32_t)sh.tl[ti]);
        scan_part<KS, VW>(K, sh, t >> 1, t & 1u);
      }
      __syncthreads();
      beat(P, tile, tid, 7, sh.ndone); beat(P, tile, tid, 1, 4);
      for (uint32_t i = (uint32_t)tid; i < sh.rtail; i += kThreads) K.R[i] = (uint16_t)kNone16;
      if ((uint32_t)tid < kLcap) K.NC[tid] = 0u;
    }
    if (!priv) {
    if (multi && nops_step) {
      if (use_static) {
        for (uint32_t i = (uint32_t)tid; i < nops_step; i += kThreads) K.PEND0[i] = K.PEND[i];
        for (uint32_t i = (uint32_t)tid; i < rtail0 && i < kLcap; i += kThreads) K.R0[i] = K.R[i];
      } else {
        // (closures beyond the link pass -- the first step of a launch recomputes every vector --: claim words, validity and the
        //  ready list go to this workgroup's piece of global scratch)
        for (uint32_t i = (uint32_t)tid; i < ns; i += kThreads) snap[i] = K.cl[i];
        for (uint32_t i = (uint32_t)tid; i < (ns + 3u) / 4u; i += kThreads) {
          uint32_t v = 0;
          for (uint32_t k = 0; k < 4u; k++) if (4u * i + k < ns) v |= (uint32_t)K.valid[4u * i + k] << (8u * k);
          snap[ns + i] = v;
        }
        for (uint32_t i = (uint32_t)tid; i < rtail0; i += kThreads) P.snap_r[(size_t)tile * ns + i] = K.R[i];
      }
    }
    __syncthreads();
    for (uint32_t tk = 0; tk < nmine; tk++) {
      if (multi) {
        set_tile(tile + tk * T);
        if (tk > 0u && nops_step) {
          const uint32_t rt_end = sh.rtail;
          __syncthreads();
          if (use_static) {
            for (uint32_t i = (uint32_t)tid; i < nops_step; i += kThreads) K.PEND[i] = K.PEND0[i];
            for (uint32_t i = (uint32_t)tid; i < rt_end; i += kThreads) K.R[i] = i < rtail0 ? K.R0[i] : (uint16_t)kNone16;
          } else {
            for (uint32_t i = (uint32_t)tid; i < ns; i += kThreads) K.cl[i] = snap[i];
            for (uint32_t i = (uint32_t)tid; i < (ns + 3u) / 4u; i += kThreads) {
              const uint32_t v = snap[ns + i];
              for (uint32_t k = 0; k < 4u; k++) if (4u * i + k < ns) K.valid[4u * i + k] = (uint8_t)(v >> (8u * k));
            }
            for (uint32_t i = (uint32_t)tid; i < rt_end; i += kThreads) K.R[i] = i < rtail0 ? P.snap_r[(size_t)tile * ns + i] : (uint16_t)kNone16;
          }
          if (tid == 0) { sh.rtail = rtail0; sh.rhead = 0; sh.ndone = 0; }
          __syncthreads();
        }
      }
      refresh<KS, VW>(K, sh, tile == 0 && wave == 0, kNW);
      __syncthreads();
      MPF_TMARK(3);
      if (tk + 1u == nmine) {
        beat(P, tile, tid, 7, sh.ndone); beat(P, tile, tid, 1, 4);
        // (the refresh's lists are done with: chain starts back to "none", consumer counts to zero -- nothing reads them before the
        //  next closure, which runs several barriers on)
        for (uint32_t i = (uint32_t)tid; i < sh.rtail; i += kThreads) K.R[i] = (uint16_t)kNone16;
        if ((uint32_t)tid < kLcap) K.NC[tid] = 0u;
      }
      // ---- (4)
      {
        // length of the two sides of the prune branch (this tile's share): per-lane scores summed over the word groups -- wave j
        // asks for prune node j's two score rows now and folds them after its scan tasks
        // (workgroups of fewer than kMaxB waves: a wave takes prune nodes wave, wave + kNW)
        constexpr int kBaseRounds = (kMaxB + (int)kNW - 1) / (int)kNW;
        uint32_t bv[kBaseRounds];
#pragma unroll
        for (int rr = 0; rr < kBaseRounds; rr++) {
          const uint32_t j = (uint32_t)wave + (uint32_t)rr * kNW;
          bv[rr] = 0;
          if (j < sh.Beff) {
            const uint32_t p = rfl(sh.pn_p[j]), q = rfl((uint32_t)K.bk[p]);
            bv[rr] = ld_sl<KS, VW>(K, p) + ld_sl<KS, VW>(K, q);
          }
        }
        const uint32_t ntasks = sh.ntasks;
        for (uint32_t ti = (uint32_t)wave; ti < ntasks; ti += kNW) {
          const uint32_t t = rfl((uint32_t)sh.tl[ti]);
          scan_part<KS, VW>(K, sh, t >> 1, t & 1u);
        }
#pragma unroll
        for (int rr = 0; rr < kBaseRounds; rr++) {
          const uint32_t j = (uint32_t)wave + (uint32_t)rr * kNW;
          if (j < sh.Beff) {
            const uint32_t tot = wave_total(K.cnt_lane ? bv[rr] : 0u);
            if (lane == 0) sh.pn_base[j] += tot;            // (wave j alone owns prune node j's word)
          }
        }
      }
      if (multi) {
        __syncthreads();                                   // (the next tile's refresh stages into the region the scans parked their up-vectors in)
        MPF_TMARK(4);
      }
    }
    if (multi && tid == 0) { sh.tph[13] += __builtin_amdgcn_s_memrealtime() - tsec0; sh.tph[14] += 100ull; }   // (steps whose closure took the plain path, tile after tile)
    }
    __syncthreads();
    beat(P, tile, tid, 1, 5);
    MPF_TMARK(4);
    // ---- (5) lengths = sum over tiles of (subtree scores at both ends of the prune branch + join cost)
    if (ncand && T == 1u) {
      // ONE workgroup holds every tile: the sums are complete where they are
      for (uint32_t c = (uint32_t)tid; c < ncand; c += kThreads) {
        uint32_t j = 0;
        while (j + 1u < sh.Beff && c >= sh.pn_off[j + 1u]) j++;
        K.cost[c] += sh.pn_base[j];
      }
    } else if (ncand) {
      const uint32_t slot = sh.xgen % 3u;
      unsigned long long *gs = P.gsum + (size_t)slot * kClimbCap;
      // level 1: the words of this workgroup's group (every eighth workgroup: 12-13 adds per word instead of 98 -- same-address
      // atomics are served one after the other); the returned count tells the group's last workgroup that the total is complete.
      // Level 2: that one forwards it with one more add (count = the group's workgroups) and clears the level-1 word for its next
      // turn, three exchanges on.  All device scope: correctness does not depend on where a workgroup runs.
      unsigned long long *xs = P.xsum + ((size_t)sh.xcc * 3u + slot) * kClimbCap;
      const uint32_t xm = sh.xm;
      const bool withhold = P.fault && P.fault != 0xFFFFFFFFu && sh.steps + 1u == P.fault && tile + 1u == T && T > 1u;
      for (uint32_t c = (uint32_t)tid; c < ncand && !withhold; c += kThreads) {
        uint32_t j = 0;
        while (j + 1u < sh.Beff && c >= sh.pn_off[j + 1u]) j++;
        const uint32_t val = K.cost[c] + sh.pn_base[j];
        if (xm == 1u) {
          __hip_atomic_fetch_add(gs + c, (1ull << 40) | (unsigned long long)val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
          const unsigned long long mine = (1ull << 40) | (unsigned long long)val;
          const unsigned long long old = __hip_atomic_fetch_add(xs + c, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if ((uint32_t)(old >> 40) + 1u == xm) {
            const unsigned long long tot = old + mine;
            __hip_atomic_fetch_add(gs + c, tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_sub(xs + c, tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      }
      for (uint32_t c = (uint32_t)tid; c < ncand; c += kThreads) {
        unsigned long long v, wait0 = 0ull;
        uint32_t spins = 0;
        for (;;) {
          v = __hip_atomic_load(gs + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if ((uint32_t)(v >> 40) >= T) break;
          if ((++spins & 255u) == 0u) {
            // a tile that does not arrive: with every workgroup resident that cannot happen -- but a chip shared with OTHER
            // processes' persistent kernels can take some of this launch's workgroups off their CUs (queue time-slicing) and not
            // bring them all back while the others spin.  Nobody waits for ever: after 100 ms of wall clock whoever notices first
            // tells everybody through the header, the launch ends with CLIMB_ERROR / err 1, and the host -- whose own state is
            // untouched until a launch comes back clean -- runs the segment as host-driven batches instead.
            if (__hip_atomic_load(&P.hdr->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { sh.err = 1u; break; }
            if (wait0 == 0ull) wait0 = __builtin_amdgcn_s_memrealtime();
            else if (__builtin_amdgcn_s_memrealtime() - wait0 > 100ull * 100000ull) {
              uint32_t expect = 0;
              if (__hip_atomic_compare_exchange_strong(&P.hdr->pad2[0], &expect, 0x80000000u | tile, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                P.hdr->pad2[1] = c | (ncand << 16);
                P.hdr->pad2[2] = (uint32_t)(v >> 40) | (sh.xgen << 16);
              }
              __hip_atomic_store(&P.hdr->abort, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              sh.err = 1u;
              break;
            }
          }
          __builtin_amdgcn_s_sleep(2);
        }
        K.cost[c] = (uint32_t)(v & kValMask);
      }
      // the slot used one exchange ago has been read by everybody who got here: its turn comes again two exchanges on (this
      // workgroup's next adds are several barriers away)
      if (tile == sh.xgen % T) {
        const uint32_t zs = (sh.xgen + 2u) % 3u;
        unsigned long long *gz = P.gsum + (size_t)zs * kClimbCap;
        for (uint32_t c = (uint32_t)tid; c < sh.last_ncand[zs]; c += kThreads) __hip_atomic_store(gz + c, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    __syncthreads();
    beat(P, tile, tid, 1, 6);
    MPF_TMARK(5);
    // ---- (6a) the reference's bookkeeping over the step's prune nodes; an accepted move is applied to the topology
    if (wave == 0) decide<KS, VW>(K, sh, P, tile);
    __syncthreads();
    beat(P, tile, tid, 8, sh.err);
    MPF_TMARK(6);
    const bool next_sweep = sh.exit_reason == CLIMB_SWEEP_END && P.sweeps_inside && sh.randomMP < sh.startMP && sh.n_moves < P.max_moves && sh.epoch + 64u < kEpochLimit;
    const bool leave = !next_sweep && sh.exit_reason != CLIMB_RUNNING;
    if (next_sweep) {
      __syncthreads();                              // (everybody has read the words lane 0 is about to change)
      // the next sweep, inside the launch: nodeRectifierPars on the topology as the last move left it -- a pre-order walk from the
      // start tip's neighbour, first child first (reference :2046-2101; Engine::node_rectifier), the inner records in the order and
      // by the record they are entered; one lane, the closure's work list as its stack
      if (tid == 0) {
        uint32_t count = 0, sp = 0;
        K.W[sp++] = K.bk[K.ord[0]];
        while (sp) {
          const uint32_t p = K.W[--sp];
          if (p < n) continue;
          K.ord[n + count] = (uint16_t)p;
          count++;
          const uint32_t p1 = nxc(p, n);
          K.W[sp++] = K.bk[nxc(p1, n)];
          K.W[sp++] = K.bk[p1];
        }
        sh.startMP = sh.randomMP;
        sh.pos = 1u;
        sh.B = P.batch_min < P.total ? P.batch_min : P.total;
        sh.since_move = 0u;
        sh.sweeps++;
        sh.exit_reason = CLIMB_RUNNING;
      }
      __syncthreads();
    }
    if (leave) break;
    // ---- (6b) side by side: wave 0 walks the invalidation the move causes, the other waves clear the candidate sums and enumerate
    // the NEXT step on the edited topology (pure topology work; what it lists is looked at for validity behind the barrier)
    if (wave == 0) {
      if (sh.moved) invalidate_walk<KS, VW>(K, sh);
      if (lane == 0) sh.wtail = 0;                       // (the walk's work list is done with: the filter fills it anew)
    } else {
      for (uint32_t i = (uint32_t)tid - 64u; i < kClimbCap; i += kThreads - 64u) K.cost[i] = 0u;
      for (uint32_t u = (uint32_t)wave - 1u; u < 2u * sh.B; u += kNW - 1u) enum_unit<KS, VW>(K, sh, P, u);
    }
  }
  beat(P, tile, tid, 1, 9);
  // ---- hand the state back
  if (tile == 0) {
    for (uint32_t i = (uint32_t)tid; i < ns; i += kThreads) P.bk[i] = K.bk[i];
    if (P.sweeps_inside) for (uint32_t i = (uint32_t)tid; i < P.total; i += kThreads) P.order[i] = K.ord[i];
    if (tid == 0) {
      ClimbHeader *h = P.hdr;
      h->start_mp = sh.startMP; h->sweeps = sh.sweeps;
      h->rng = sh.rng; h->hits = sh.hits; h->best = sh.best; h->randomMP = sh.randomMP; h->iter_hits = sh.iter_hits;
      h->pos = sh.pos; h->insert_cid = sh.ins; h->remove_cid = sh.rem; h->n_moves = sh.n_moves; h->reason = sh.exit_reason;
      h->err = sh.err; h->steps = sh.steps; h->n_tests = sh.n_tests; h->n_ops = sh.n_ops; h->draws = sh.draws;
      h->n_scanned_nodes = sh.n_nodes; h->since_move = sh.since_move; h->batch = sh.B;
      h->pad[0] = sh.trace_n; h->pad[1] = sh.c_rounds; h->pad[2] = sh.c_inv; h->pad[3] = sh.c_chains;
      for (int i = 0; i < 7; i++) h->tph[i] = sh.tph[i];
      for (int i = 8; i < 16; i++) h->tph[i] = sh.tph[i];
      h->tph[12] = sh.c_parts * 100ull;
      // shader clock in kHz: s_memtime ticks per s_memrealtime tick (100 MHz)
      h->tph[7] = (__builtin_amdgcn_s_memtime() - sh.clk0) * 100000ull / (__builtin_amdgcn_s_memrealtime() - sh.rt0 + 1ull);
    }
  }
}

template <int KS, int VW, bool GROUPS>
__global__ __launch_bounds__((Cfg<KS, VW>::NT)) void k_climb(ClimbParams P) { climb_body<KS, VW, GROUPS ? 1 : 0>(P, blockIdx.x, gridDim.x); }

// MANY climbs in one launch, one resident workgroup each (ClimbParams::groups == 1 semantics: nothing crosses between workgroups,
// so they need not be resident together -- a grid larger than the chip simply runs in turns).  The engines of such a batch share
// the alignment's shape (state rows, tile width); every one has its own vector store, topology, tie stream and result block.
template <int KS, int VW>
__global__ __launch_bounds__((Cfg<KS, VW>::NT_MANY)) void k_climb_many(const ClimbParams *__restrict__ PP)
{
  const ClimbParams P = PP[blockIdx.x];
  climb_body<KS, VW, 2>(P, 0u, 1u);
}

template <int KS, int VW>
size_t lds_bytes(uint32_t ns, uint32_t maxb, bool many)
{
  return lds_fixed_bytes(ns, maxb) + region_bytes<KS, VW>(ns, maxb, many ? Cfg<KS, VW>::NW_MANY : Cfg<KS, VW>::NW);
}

template <int KS, int VW, bool GROUPS>
hipError_t launch_g(hipStream_t st, const ClimbParams &p)
{
  const size_t lds = lds_bytes<KS, VW>(p.nslots, p.batch_max, false);
  static thread_local int attr_dev = -1;
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (lds > 64 * 1024 || attr_dev != dev) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_climb<KS, VW, GROUPS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_dev = dev;
  }
  hipLaunchKernelGGL((k_climb<KS, VW, GROUPS>), dim3(p.groups ? p.groups : p.tiles), dim3(Cfg<KS, VW>::NT), lds, st, p);
  return hipGetLastError();
}

template <int KS, int VW>
hipError_t launch_t(hipStream_t st, const ClimbParams &p)
{
  return p.groups && p.groups < p.tiles ? launch_g<KS, VW, true>(st, p) : launch_g<KS, VW, false>(st, p);
}

template <int KS, int VW>
hipError_t launch_many_t(hipStream_t st, const ClimbParams *d_params, int n_climbs, size_t lds)
{
  static thread_local int attr_dev = -1;
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (lds > 64 * 1024 || attr_dev != dev) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_climb_many<KS, VW>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_dev = dev;
  }
  hipLaunchKernelGGL((k_climb_many<KS, VW>), dim3((unsigned)n_climbs), dim3(Cfg<KS, VW>::NT_MANY), lds, st, d_params);
  return hipGetLastError();
}

}  // namespace

static inline uint32_t slots_of(int n) { return (uint32_t)n + 3u * (uint32_t)(n - 1); }

int climb_tiles(const Geometry &g, int vw) { return (g.Wp + 16 * vw - 1) / (16 * vw); }

size_t climb_lds_bytes(const Geometry &g, int n_taxa, int vw, int batch_max, bool many, bool word_major)
{
  const uint32_t ns = slots_of(n_taxa), mb = (uint32_t)batch_max;
  if (g.S == 4 && word_major && vw == 4) return lds_bytes<4, 1>(ns, mb, many);
  if (g.S == 4) return vw == 1 ? lds_bytes<1, 1>(ns, mb, many) : vw == 2 ? lds_bytes<1, 2>(ns, mb, many) : vw == 4 ? lds_bytes<1, 4>(ns, mb, many) : lds_bytes<1, 8>(ns, mb, many);
  if (g.S == 32) return lds_bytes<8, 1>(ns, mb, many);
  return lds_bytes<5, 1>(ns, mb, many);
}

bool climb_supported(const Geometry &g, int n_taxa, int maxtrav, int batch_max)
{
  if (g.sankoff || g.big) return false;
  if (g.S != 4 && g.S != 20 && g.S != 32) return false;
  if (maxtrav < 1 || maxtrav > kDepth) return false;
  if (slots_of(n_taxa) + 16u >= 0xFFFFu) return false;
  if (climb_tiles(g, 1) >= (1 << 20)) return false;
  return climb_lds_bytes(g, n_taxa, 1, batch_max < 1 ? 1 : batch_max > kMaxB ? kMaxB : batch_max, false, false) <= kLdsBudget;
}

hipError_t launch_climb(hipStream_t st, const Geometry &g, int vw, const ClimbParams &p, bool word_major)
{
  // (the word-major shape: 64-word tiles, a workgroup per tile)
  if (g.S == 4 && word_major && vw == 4 && !(p.groups && p.groups < p.tiles)) return launch_g<4, 1, false>(st, p);
  if (g.S == 4) {
    if (vw == 1) return launch_t<1, 1>(st, p);
    if (vw == 2) return launch_t<1, 2>(st, p);
    if (vw == 8) return launch_t<1, 8>(st, p);
    return launch_t<1, 4>(st, p);
  }
  if (g.S == 32) return launch_t<8, 1>(st, p);           // 32-state data: eight states per lane
  return launch_t<5, 1>(st, p);
}

hipError_t launch_climb_many(hipStream_t st, const Geometry &g, int vw, const ClimbParams *d_params, int n_climbs, size_t lds, bool word_major)
{
  if (n_climbs <= 0) return hipSuccess;
  if (g.S == 4 && word_major) {
    if (vw != 4) return hipErrorInvalidValue;               // (64-word tiles: a word per lane)
    return launch_many_t<4, 1>(st, d_params, n_climbs, lds);
  }
  if (g.S == 4) {
    if (vw == 1) return launch_many_t<1, 1>(st, d_params, n_climbs, lds);
    if (vw == 2) return launch_many_t<1, 2>(st, d_params, n_climbs, lds);
    if (vw == 8) return launch_many_t<1, 8>(st, d_params, n_climbs, lds);
    return launch_many_t<1, 4>(st, d_params, n_climbs, lds);
  }
  if (g.S == 32) return launch_many_t<8, 1>(st, d_params, n_climbs, lds);
  return launch_many_t<5, 1>(st, d_params, n_climbs, lds);
}

}  // namespace mpf
