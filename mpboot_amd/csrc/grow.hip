// grow.hip -- k_grow: the addition loop of _pllMakeParsimonyTreeFast (reference sprparsimony.cpp:3107-3181, stepwiseAddition
// :2977-3019) as ONE persistent gfx950 kernel per tree (interface and rationale: grow.hpp).
//
// One step = one added taxon (all workgroups in lockstep, no communication except (4)):
//   (1) plan:      the skeleton -- nodes with more than S descendants, in pre-order -- by an ordered compaction over the pre-order array;
//   (2) skeleton:  wave 0 walks it top-down from the start tip's vector: per node both children's D are loaded once,
//                  U(child) = fitch(U(node), D(other child)), the children's candidate costs are booked, U of a second child that is
//                  walked later is parked in HBM, U of every cut-off subtree's root is left in HBM with the root on the part list;
//   (3) parts:     the waves take parts off the list and walk them the same way (at most 64 nodes: one ballot gives the walk);
//   (4) exchange:  one row per workgroup (16-bit partial costs in pairs, flag bits of the last root path), an arrival counter,
//                  every workgroup sums all rows; the descent cut's "subtree score > 0" flags are brought up to date from the bits;
//   (5) decide:    stepwiseAddition's bookkeeping over the candidates in pre-order (= the reference's visiting order): descent cut,
//                  random tie rule with the lcg64 stream (:3004), or the first minimum;
//   (6) insert:    the new inner node and the tip enter the rooted tree: pre-order positions shift by two behind the branch, the
//                  subtree below it gets one level deeper, its ancestors two nodes larger (every thread a few entries);
//   (7) path:      D of the new node and of its ancestors, a chain in registers with the siblings' vectors requested ahead.
#include "grow.hpp"

#include "../../include/mpfitch.h"

namespace mpf {

namespace {

#include "quadtile.hpp"

template <int KS, int VW> struct GCfg {
  static constexpr int R = KS * VW;                                   // registers per vector tile
  static constexpr int NW = R <= 2 ? 16 : 8;
  static constexpr int NT = NW * 64;
  static constexpr int PF = R <= 2 ? 4 : 2;                           // expansions whose child vectors are requested together (two sets in rotation)
};

struct GSh {
  uint32_t m, step, ok, err, exit_reason, root, len, S;
  uint32_t nskel, nparts, part_next, nzero, path_n, ins_pos, best, npub, skel_done, pad0[3];
  uint32_t wcnt[16];
  unsigned long long rng, draws;
  unsigned long long tph[8], tlast;
};

template <int KS, int VW>
struct Gx {
  uint16_t *par, *ch1, *ch2, *pos, *sz, *dep, *dcid, *ord, *skl, *parts, *path, *psib;
  uint32_t *cost, *gflag;
  uint4 *prog;       // [waves][64] the walks' programs
  uint8_t *nz, *pflag;
  uint32_t n, N2, SW4;
  int lane, wave;
  __amdgpu_buffer_rsrc_t rsrc;
  uint32_t voff[KS];
  uint32_t zero;
  bool cnt_lane, st_lane;
};

// KS == 4 is the WORD-MAJOR DNA layout: the kernel works on the engine's word-major copy of the vectors (Geometry::shoff: the four
// state words of a 32-site word side by side), a lane holds ALL FOUR states of ONE word.  A vector tile is then one contiguous
// kilobyte -- one buffer_load_dwordx4 per wave instead of four 256-byte row segments -- and Fitch's cross-state OR stays inside
// the lane: half the vector instructions of the quad layout and four times longer bursts (the addition loop reads every down
// vector once per added taxon: at a dozen trees side by side it is bound by what HBM delivers for such gathers).
template <int KS, int VW>
__device__ __forceinline__ void g_ld(const Gx<KS, VW> &G, QT<KS, VW> &t, uint32_t cid)
{
  if constexpr (KS == 4) {
    const v4u x = __builtin_amdgcn_raw_buffer_load_b128(G.rsrc, G.voff[0], cid * G.SW4, 0);
    t.v[0][0] = x[0]; t.v[1][0] = x[1]; t.v[2][0] = x[2]; t.v[3][0] = x[3];
  } else {
    qload<KS, VW>(t, G.rsrc, G.voff, cid * G.SW4);
  }
}
template <int KS, int VW>
__device__ __forceinline__ void g_st(const Gx<KS, VW> &G, const QT<KS, VW> &t, uint32_t cid)
{
  if (!G.st_lane) return;
  if constexpr (KS == 4) {
    v4u x; x[0] = t.v[0][0]; x[1] = t.v[1][0]; x[2] = t.v[2][0]; x[3] = t.v[3][0];
    __builtin_amdgcn_raw_buffer_store_b128(x, G.rsrc, G.voff[0], cid * G.SW4, 0);
  } else {
    qstore<KS, VW>(t, G.rsrc, G.voff, cid * G.SW4);
  }
}
// c = fitch(a, b); returns this lane's count of sites with an empty intersection (quad layout: the same number in the four lanes
// of a quad, counted by the first; word-major: every lane its own word)
template <int KS, int VW>
__device__ __forceinline__ uint32_t v_fitch(QT<KS, VW> &c, const QT<KS, VW> &a, const QT<KS, VW> &b)
{
  if constexpr (KS == 4) {
    const uint32_t any = b3_andor(a.v[3][0], b.v[3][0], b3_andor(a.v[2][0], b.v[2][0], b3_andor(a.v[1][0], b.v[1][0], a.v[0][0] & b.v[0][0])));
#pragma unroll
    for (int k = 0; k < 4; k++) c.v[k][0] = b3_fitch(a.v[k][0], b.v[k][0], any);
    return (uint32_t)__builtin_popcount(~any);
  } else {
    return q_fitch<KS, VW>(c, a, b);
  }
}
template <int KS, int VW>
__device__ __forceinline__ uint32_t v_join(const QT<KS, VW> &u, const QT<KS, VW> &d, const QT<KS, VW> &s)
{
  if constexpr (KS == 4) {
    const uint32_t any = b3_andor(u.v[3][0], d.v[3][0], b3_andor(u.v[2][0], d.v[2][0], b3_andor(u.v[1][0], d.v[1][0], u.v[0][0] & d.v[0][0])));
    uint32_t hit = b3_fitch(u.v[0][0], d.v[0][0], any) & s.v[0][0];
#pragma unroll
    for (int k = 1; k < 4; k++) hit = b3_andor(b3_fitch(u.v[k][0], d.v[k][0], any), s.v[k][0], hit);
    return (uint32_t)__builtin_popcount(~hit);
  } else {
    return q_join<KS, VW>(u, d, s);
  }
}

// a vector tile in a scratch slot (parked up-vectors, the parts' roots): [slot][R][64] words, lane-major
template <int KS, int VW>
__device__ __forceinline__ void s_st(uint32_t *base, uint32_t slot, const QT<KS, VW> &t, int lane)
{
  uint32_t *p = base + (size_t)slot * (size_t)(KS * VW * 64) + lane;
#pragma unroll
  for (int k = 0; k < KS * VW; k++) p[k * 64] = t.v[k / VW][k % VW];
}
template <int KS, int VW>
__device__ __forceinline__ void s_ld(const uint32_t *base, uint32_t slot, QT<KS, VW> &t, int lane)
{
  const uint32_t *p = base + (size_t)slot * (size_t)(KS * VW * 64) + lane;
#pragma unroll
  for (int k = 0; k < KS * VW; k++) t.v[k / VW][k % VW] = p[k * 64];
}

__device__ __forceinline__ double g_tie_draw(unsigned long long &st)
{
  st = st * 0x27bb2ee687b0b0fdULL + 3037000493ULL;       // sprng/lcg64.c:220
  return (double)st * 5.4210108624275222e-20;            // :268
}

__device__ __forceinline__ void publish_parts(GSh &sh, uint32_t np, int lane);

// the walk of the skeleton (SKEL: wave 0, expansions = sh.nskel nodes of G.skl) or of one part (root r, at most 64 nodes).
// Ur = U(r).  Books the candidate costs of every expanded node's two children; U of the first child stays in registers, U of a
// second child that is expanded later waits in `park` at the level of its parent.
// What an expansion needs to know about the tree -- its children's vectors and pre-order positions, where its own U comes from,
// what becomes of the children -- is worked out for 64 expansions at a time, a lane each, into a 16-byte entry of this wave's
// program buffer: the walk itself then costs ONE broadcast LDS read per expansion (the first version asked the tree arrays a
// dozen dependent questions per expansion, each a round trip to LDS through readfirstlane: 0.7 us of the 0.9 us an expansion took).
// The children's vectors of the next PF expansions are requested before the current PF are worked on (two register sets in rotation).
template <int KS, int VW, bool SKEL>
__device__ __forceinline__ void walk(const Gx<KS, VW> &G, GSh &sh, uint32_t r, const QT<KS, VW> &Ur, const QT<KS, VW> &T, uint32_t *park,
                                     uint32_t *ucp)
{
  constexpr int PF = GCfg<KS, VW>::PF;
  const int lane = G.lane;
  const uint32_t S = sh.S;
  const uint32_t depr = rfl((uint32_t)G.dep[r]);
  uint4 *prog = G.prog + (size_t)G.wave * 64;
  uint32_t E, base = 0;
  unsigned long long todo = 0ull;
  if constexpr (SKEL) {
    E = sh.nskel;
  } else {
    base = rfl((uint32_t)G.pos[r]);
    const uint32_t cnt = rfl((uint32_t)G.sz[r]);
    bool inner = false;
    if ((uint32_t)lane < cnt) inner = G.sz[G.ord[base + (uint32_t)lane]] > 1u;
    todo = __ballot((int)inner);
    E = (uint32_t)__builtin_popcountll(todo);
  }
  uint32_t np = SKEL ? sh.nparts : 0u, np_pub = np;
  (void)np_pub;
  QT<KS, VW> u1, u2, par;
  struct Set { QT<KS, VW> d1[PF], d2[PF]; } A, B;
  for (uint32_t q0 = 0; q0 < E; q0 += 64u) {
    const uint32_t EC = E - q0 < 64u ? E - q0 : 64u;                  // expansions of this chunk
    // ---- the chunk's program, a lane per expansion
    {
      uint32_t e = 0, slot = 0;
      bool mine;
      if constexpr (SKEL) {
        mine = (uint32_t)lane < EC;
        if (mine) e = G.skl[q0 + (uint32_t)lane];
        slot = (uint32_t)lane;
      } else {
        mine = ((todo >> lane) & 1ull) != 0ull;                        // (a part is one chunk: E <= 64)
        if (mine) e = G.ord[base + (uint32_t)lane];
        slot = (uint32_t)__builtin_popcountll(todo & ((1ull << lane) - 1ull));
      }
      if (mine) {
        const uint32_t a = G.ch1[e], b = G.ch2[e];
        const uint32_t sza = G.sz[a], szb = G.sz[b];
        uint32_t z = (G.dep[e] - depr) << 8;
        const bool walk_a = sza > 1u && (!SKEL || sza > S);
        if (e == r) {
          z |= 2u;
        } else {
          const uint32_t pe = G.par[e], sib = G.ch1[pe];
          if (sib == e) z |= 1u;                                       // first child: its parent was the expansion just before
          else if (!(G.sz[sib] > 1u && (!SKEL || G.sz[sib] > S))) z |= 32u;   // second child of a parent whose first child is not walked: the parent was the expansion just before, U is still in its second register set
        }
        if (szb > 1u && (!SKEL || szb > S) && walk_a) z |= 4u;         // both children are walked: the second one's U waits in HBM
        if (SKEL && sza > 1u && sza <= S) z |= 8u;                     // a cut-off subtree's root: its U goes on the part list
        if (SKEL && szb > 1u && szb <= S) z |= 16u;
        prog[slot] = make_uint4((uint32_t)G.dcid[a] | ((uint32_t)G.dcid[b] << 16), (uint32_t)G.pos[a] | ((uint32_t)G.pos[b] << 16), z, e);
      }
    }
    // requests of the expansions [blk, blk + PF) of the chunk into one register set (clamped: the requests stay unconditional)
    auto issue = [&](Set &X, uint32_t blk) {
#pragma unroll
      for (int i = 0; i < PF; i++) {
        uint32_t q = blk + (uint32_t)i;
        q = q < EC ? q : EC - 1u;
        const uint32_t x = rfl(prog[q].x);
        g_ld<KS, VW>(G, X.d1[i], x & 0xFFFFu);
        g_ld<KS, VW>(G, X.d2[i], x >> 16);
      }
    };
    auto work = [&](Set &X, uint32_t blk) {
#pragma unroll
      for (int i = 0; i < PF; i++) {
        const uint32_t q = blk + (uint32_t)i;
        if (q < EC) {
          const uint4 en = prog[q];
          const uint32_t y = rfl(en.y), z = rfl(en.z);
          const uint32_t lev = z >> 8;
          if (z & 2u) par = Ur;
          else if (z & 1u) par = u1;
          else if (z & 32u) par = u2;
          else s_ld<KS, VW>(park, lev - 1u, par, lane);
          v_fitch<KS, VW>(u1, par, X.d2[i]);                          // U(a) = fitch(U(e), D(b))
          v_fitch<KS, VW>(u2, par, X.d1[i]);
          const uint32_t j1 = v_join<KS, VW>(u1, X.d1[i], T);
          const uint32_t j2 = v_join<KS, VW>(u2, X.d2[i], T);
          if (G.cnt_lane) {
            __hip_atomic_fetch_add(G.cost + (y & 0xFFFFu) + G.zero, j1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(G.cost + (y >> 16) + G.zero, j2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
          if (z & 4u) {
            if (lev >= (SKEL ? kGrowParkSkel : kGrowParkPart)) { if (lane == 0) sh.err = 3u; }
            else s_st<KS, VW>(park, lev, u2, lane);
          }
          if constexpr (SKEL) {
            if (z & 24u) {
              const uint32_t e = rfl(en.w);
              if (z & 8u) {
                if (np < kGrowMaxParts) { s_st<KS, VW>(ucp, np, u1, lane); if (lane == 0) G.parts[np] = G.ch1[e]; np++; }
                else if (lane == 0) sh.err = 4u;
              }
              if (z & 16u) {
                if (np < kGrowMaxParts) { s_st<KS, VW>(ucp, np, u2, lane); if (lane == 0) G.parts[np] = G.ch2[e]; np++; }
                else if (lane == 0) sh.err = 4u;
              }
            }
          }
        }
      }
    };
    issue(A, 0u);
    for (uint32_t blk = 0; blk < EC; blk += 2u * (uint32_t)PF) {
      issue(B, blk + (uint32_t)PF);
      work(A, blk);
      // (the parts left so far: published with the stores in front of the count.  The release waits for every request that is out --
      //  here those of set B, which the next lines wait for anyway)
      if constexpr (SKEL) { if (np != np_pub) { publish_parts(sh, np, lane); np_pub = np; } }
      issue(A, blk + 2u * (uint32_t)PF);
      work(B, blk + (uint32_t)PF);
    }
  }
  if constexpr (SKEL) { if (lane == 0) sh.nparts = np; }
}

// the parts the skeleton's walk has left so far become visible to the waiting waves: their roots' U (HBM) first, then the count
__device__ __forceinline__ void publish_parts(GSh &sh, uint32_t np, int lane)
{
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  if (lane == 0) __hip_atomic_store(&sh.npub, np, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <int KS, int VW>
__device__ __forceinline__ void grow_body(const GrowParams &P)
{
  constexpr uint32_t kThreads = GCfg<KS, VW>::NT, kNW = GCfg<KS, VW>::NW;
  constexpr int PF = GCfg<KS, VW>::PF;
  constexpr int R = GCfg<KS, VW>::R;
  extern __shared__ __align__(16) unsigned char smem[];
  const int tid = (int)threadIdx.x, lane = tid & 63;
  const int wave = (int)rfl((uint32_t)(tid >> 6));
  const uint32_t tile = blockIdx.x, n = P.n, N2 = 2u * P.n, T = P.tiles;
  GSh &sh = *reinterpret_cast<GSh *>(smem);
  size_t at = (sizeof(GSh) + 15) & ~(size_t)15;
  Gx<KS, VW> G;
  auto carve16 = [&](uint16_t *&p) { p = reinterpret_cast<uint16_t *>(smem + at); at += (((size_t)N2 * 2) + 15) & ~(size_t)15; };
  carve16(G.par); carve16(G.ch1); carve16(G.ch2); carve16(G.pos); carve16(G.sz); carve16(G.dep); carve16(G.dcid);
  carve16(G.ord); carve16(G.skl); carve16(G.parts); carve16(G.path); carve16(G.psib);
  G.cost = reinterpret_cast<uint32_t *>(smem + at); at += (size_t)N2 * 4;
  G.gflag = reinterpret_cast<uint32_t *>(smem + at); at += (((size_t)(N2 / 32 + 2) * 4) + 15) & ~(size_t)15;
  G.nz = reinterpret_cast<uint8_t *>(smem + at); at += ((size_t)N2 + 15) & ~(size_t)15;
  G.pflag = reinterpret_cast<uint8_t *>(smem + at); at += ((size_t)N2 + 15) & ~(size_t)15;
  G.prog = reinterpret_cast<uint4 *>(smem + at);
  G.n = n; G.N2 = N2; G.lane = lane; G.wave = wave;
  asm volatile("v_mov_b32 %0, 0" : "=v"(G.zero));
  G.rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)P.vec, 0, 0x7FFFFFFF, 0x00020000);
  if constexpr (KS == 4) {
    // word-major copy (P.vec points at it): vector = Wp x 16 bytes, this lane's word at 16 bytes x word
    G.SW4 = P.Wp * 16u;
    uint32_t word0 = tile * 64u + (uint32_t)lane;
    G.st_lane = word0 < P.Wp;
    if (!G.st_lane) word0 = P.Wp - 1u;
    G.cnt_lane = G.st_lane;
    G.voff[0] = word0 * 16u;
    G.voff[1] = G.voff[2] = G.voff[3] = 0;
  } else {
    G.SW4 = (uint32_t)(4 * KS) * P.Wp * 4u;
    const uint32_t w = (uint32_t)lane >> 2, g = (uint32_t)lane & 3u;
    uint32_t word0 = (tile * 16u + w) * (uint32_t)VW;
    G.st_lane = word0 < P.Wp;
    if (!G.st_lane) word0 = P.Wp - (uint32_t)VW;         // lanes past the row end load real data and contribute nothing
    G.cnt_lane = G.st_lane && g == 0u;
#pragma unroll
    for (int k = 0; k < KS; k++) G.voff[k] = ((g * (uint32_t)KS + (uint32_t)k) * P.Wp + word0) * 4u;
  }
  // scratch of this tile: parked up-vectors (per wave for the parts, one deep region for the skeleton), U of the parts' roots
  const size_t slot_words = (size_t)R * 64;
  uint32_t *park_tile = P.park + (size_t)tile * (size_t)(kNW * kGrowParkPart + kGrowParkSkel) * slot_words;
  uint32_t *park_w = park_tile + (size_t)wave * kGrowParkPart * slot_words;
  uint32_t *park_s = park_tile + (size_t)kNW * kGrowParkPart * slot_words;
  uint32_t *ucp = P.ucp + (size_t)tile * kGrowMaxParts * slot_words;

  // ---- the start tree
  {
    const uint16_t *in = P.init;
    for (uint32_t i = (uint32_t)tid; i < N2; i += kThreads) {
      G.par[i] = in[i]; G.ch1[i] = in[N2 + i]; G.ch2[i] = in[2 * N2 + i]; G.pos[i] = in[3 * N2 + i]; G.sz[i] = in[4 * N2 + i];
      G.dep[i] = in[5 * N2 + i]; G.dcid[i] = in[6 * N2 + i]; G.ord[i] = in[7 * N2 + i];
      G.nz[i] = 0; G.pflag[i] = 0;
    }
  }
  if (tid == 0) {
    const GrowHeader h = *P.hdr;
    sh.m = P.m0; sh.step = 0; sh.err = 0; sh.exit_reason = GROW_RUNNING; sh.root = P.root_node; sh.len = P.len0;
    sh.nzero = 0; sh.path_n = 0; sh.rng = h.rng; sh.draws = 0;
    for (int i = 0; i < 8; i++) sh.tph[i] = 0;
    sh.tlast = __builtin_amdgcn_s_memrealtime();
    // every workgroup must be resident before anyone waits for anyone (as in k_climb: ONE compare-and-swap decides go or abort)
    __hip_atomic_fetch_add(&P.hdr->arrive, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
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
    sh.ok = gate == 1u ? 1u : 0u;
  }
  __syncthreads();
  if (!sh.ok) {
    if (tile == 0 && tid == 0) P.hdr->reason = GROW_ABORT;
    return;
  }
#define MPF_GMARK(i) do { if (tile == 0 && tid == 0) { const unsigned long long now_ = __builtin_amdgcn_s_memrealtime(); sh.tph[i] += now_ - sh.tlast; sh.tlast = now_; } } while (0)

  // D of the start tree's inner nodes, children first (reverse pre-order), and this tile's "the join costs something" flags:
  // they travel with the first step's exchange like a root path's
  if (wave == 0) {
    uint32_t k = 0;
    for (int i = (int)P.m0 - 1; i >= 0; i--) {
      const uint32_t c = rfl((uint32_t)G.ord[i]);
      if (rfl((uint32_t)G.sz[c]) <= 1u) continue;
      QT<KS, VW> a, b, d;
      g_ld<KS, VW>(G, a, rfl((uint32_t)G.dcid[rfl((uint32_t)G.ch1[c])]));
      g_ld<KS, VW>(G, b, rfl((uint32_t)G.dcid[rfl((uint32_t)G.ch2[c])]));
      const uint32_t cnt = v_fitch<KS, VW>(d, a, b);
      g_st<KS, VW>(G, d, rfl((uint32_t)G.dcid[c]));
      const bool any = __ballot((int)(G.cnt_lane && cnt > 0u)) != 0ull;
      if (lane == 0) { G.path[k] = (uint16_t)c; G.psib[k] = (uint16_t)kGrowNone; G.pflag[k] = any ? 1 : 0; }
      k++;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");          // (a parent further up reads this vector back)
    }
    if (lane == 0) sh.path_n = k;
  }
  __syncthreads();

  QT<KS, VW> Tv;                                   // the tip of this step
  for (;;) {
    const uint32_t step = sh.step, m = sh.m;
    g_ld<KS, VW>(G, Tv, (uint32_t)P.tips[step]);
    // ---- (1) plan
    if (tid == 0) {
      uint32_t S = m / (3u * kNW);
      S = S < 8u ? 8u : S > kGrowParkPart ? kGrowParkPart : S;
      sh.S = S; sh.nparts = 0; sh.part_next = 0; sh.nskel = 0; sh.npub = 0; sh.skel_done = 0;
    }
    for (uint32_t i = (uint32_t)tid; i < m; i += kThreads) G.cost[i] = 0u;
    __syncthreads();
    {
      const uint32_t S = sh.S;
      uint32_t basecnt = 0;
      for (uint32_t b0 = 0; b0 < m; b0 += kThreads) {
        const uint32_t i = b0 + (uint32_t)tid;
        uint32_t c = 0;
        bool big = false;
        if (i < m) { c = G.ord[i]; big = G.sz[c] > S; }
        const unsigned long long mk = __ballot((int)big);
        if (lane == 0) sh.wcnt[wave] = (uint32_t)__builtin_popcountll(mk);
        __syncthreads();
        uint32_t off = basecnt, tot = 0;
        for (uint32_t w = 0; w < kNW; w++) { const uint32_t cw = sh.wcnt[w]; if (w < (uint32_t)wave) off += cw; tot += cw; }
        if (big) G.skl[off + (uint32_t)__builtin_popcountll(mk & ((1ull << lane) - 1ull))] = (uint16_t)c;
        basecnt += tot;
        __syncthreads();
      }
      if (tid == 0) sh.nskel = basecnt;
    }
    __syncthreads();
    MPF_GMARK(0);
    // ---- (2) skeleton: the root's child first (its U is the start tip's vector).  The other waves do not wait for the walk to end:
    // they take the parts it leaves as it goes (3).
    if (wave == 0) {
      const uint32_t c0 = sh.root;
      QT<KS, VW> U0, D0;
      g_ld<KS, VW>(G, U0, P.root_cid);
      g_ld<KS, VW>(G, D0, rfl((uint32_t)G.dcid[c0]));
      const uint32_t j0 = v_join<KS, VW>(U0, D0, Tv);
      if (G.cnt_lane) __hip_atomic_fetch_add(G.cost + G.zero, j0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      const uint32_t s0 = rfl((uint32_t)G.sz[c0]);
      if (s0 > 1u) {
        if (s0 > sh.S) {
          walk<KS, VW, true>(G, sh, c0, U0, Tv, park_s, ucp);
        } else {
          s_st<KS, VW>(ucp, 0u, U0, lane);
          if (lane == 0) { G.parts[0] = (uint16_t)c0; sh.nparts = 1u; }
        }
      }
      publish_parts(sh, rfl(*(volatile uint32_t *)&sh.nparts), lane);
      if (lane == 0) __hip_atomic_store(&sh.skel_done, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      MPF_GMARK(1);
    }
    // ---- (3) parts: a wave claims the next index and waits until the skeleton's walk has got that far (or is over)
    for (;;) {
      const uint32_t pi = wave_fetch_add(&sh.part_next, 1u, lane);
      bool have = false;
      for (;;) {
        const uint32_t done = rfl(__hip_atomic_load(&sh.skel_done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP));
        const uint32_t pub = rfl(__hip_atomic_load(&sh.npub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        if (pi < pub) { have = true; break; }
        if (done) break;                                     // (npub was read behind the flag: it is final)
        __builtin_amdgcn_s_sleep(1);
      }
      if (!have) break;
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      QT<KS, VW> Ur;
      s_ld<KS, VW>(ucp, pi, Ur, lane);
      walk<KS, VW, false>(G, sh, rfl((uint32_t)G.parts[pi]), Ur, Tv, park_w, ucp);
    }
    __syncthreads();
    MPF_GMARK(2);
    // ---- (4) exchange
    const uint32_t cw = (m + 1u) >> 1, pn = sh.path_n, fw = (pn + 31u) >> 5;
    {
      uint32_t *row = P.xrow + ((size_t)(step & 1u) * T + tile) * P.xstride;
      for (uint32_t w = (uint32_t)tid; w < cw; w += kThreads) {
        const uint32_t lo = G.cost[2u * w], hi = 2u * w + 1u < m ? G.cost[2u * w + 1u] : 0u;
        __hip_atomic_store(row + w, lo | (hi << 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      for (uint32_t w = (uint32_t)tid; w < fw; w += kThreads) {
        uint32_t bits = 0;
        for (uint32_t j = 0; j < 32u && 32u * w + j < pn; j++) bits |= (uint32_t)(G.pflag[32u * w + j] ? 1u : 0u) << j;
        __hip_atomic_store(row + cw + w, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    __syncthreads();
    if (tid == 0) {
      __hip_atomic_fetch_add(&P.hdr->xarrive, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      const uint32_t want = T * (step + 1u);
      unsigned long long wait0 = 0ull;
      uint32_t spins = 0;
      for (;;) {
        if (__hip_atomic_load(&P.hdr->xarrive, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= want) break;
        if ((++spins & 255u) == 0u) {
          // (a chip shared with other processes' persistent kernels can take workgroups of this launch off their CUs: nobody
          //  waits for ever -- after 100 ms whoever notices first tells everybody, the host's own loop builds the tree instead)
          if (__hip_atomic_load(&P.hdr->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { sh.err = 1u; break; }
          if (wait0 == 0ull) wait0 = __builtin_amdgcn_s_memrealtime();
          else if (__builtin_amdgcn_s_memrealtime() - wait0 > 100ull * 100000ull) {
            __hip_atomic_store(&P.hdr->abort, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sh.err = 1u;
            break;
          }
        }
        __builtin_amdgcn_s_sleep(2);
      }
    }
    __syncthreads();
    if (sh.err) break;
    {
      const uint32_t *rows = P.xrow + (size_t)(step & 1u) * T * P.xstride;
      for (uint32_t w = (uint32_t)tid; w < cw + fw; w += kThreads) {
        uint32_t lo = 0, hi = 0, bits = 0;
        for (uint32_t t = 0; t < T; t++) {
          const uint32_t v = __hip_atomic_load(rows + (size_t)t * P.xstride + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          lo += v & 0xFFFFu; hi += v >> 16; bits |= v;
        }
        if (w < cw) { G.cost[2u * w] = lo; if (2u * w + 1u < m) G.cost[2u * w + 1u] = hi; }
        else G.gflag[w - cw] = bits;
      }
    }
    __syncthreads();
    // the descent cut's flags (tr->parsimonyScore[q] > 0, :3014) of the nodes whose D has changed: nz(node) = nz(children) | "its own
    // join costs something somewhere".  A root path bottom-up is a running OR; the start tree's list goes node by node.
    if (wave == 0 && pn > 0u) {
      if (rfl((uint32_t)G.psib[0]) == kGrowNone && step == 0u) {
        if (lane == 0) {
          uint32_t nzero = 0;
          for (uint32_t i = 0; i < pn; i++) {
            const uint32_t c = G.path[i];
            const uint8_t v = (uint8_t)(G.nz[G.ch1[c]] | G.nz[G.ch2[c]] | ((G.gflag[i >> 5] >> (i & 31u)) & 1u));
            G.nz[c] = v;
            nzero += v ? 0u : 1u;
          }
          sh.nzero = nzero;
        }
      } else {
        uint32_t carry = 0;
        int dz = 0;
        for (uint32_t b0 = 0; b0 < pn; b0 += 64u) {
          const uint32_t i = b0 + (uint32_t)lane;
          const bool in = i < pn;
          uint32_t node = 0, bi = 0, old = 1;
          if (in) {
            node = G.path[i];
            bi = ((G.gflag[i >> 5] >> (i & 31u)) & 1u) | (uint32_t)G.nz[G.psib[i]];
            old = i == 0u ? 1u : (uint32_t)G.nz[node];               // (the new node had no flag: counted as "not zero" before)
          }
          const unsigned long long mk = __ballot((int)(bi != 0u));
          const uint32_t nzi = (carry | ((mk & ((2ull << lane) - 1ull)) != 0ull ? 1u : 0u));
          if (in) G.nz[node] = (uint8_t)nzi;
          dz += __builtin_popcountll(__ballot((int)(in && nzi == 0u))) - __builtin_popcountll(__ballot((int)(in && old == 0u)));
          carry |= mk != 0ull ? 1u : 0u;
        }
        if (lane == 0) sh.nzero = (uint32_t)((int)sh.nzero + dz);
      }
    }
    __syncthreads();
    // (a subtree without a single mutation below an inner node: its branches are not visited -- rare; marked here, read by (5))
    const bool cut_any = sh.nzero != 0u;
    if (cut_any) {
      for (uint32_t i = (uint32_t)tid; i < m; i += kThreads) G.pflag[i] = 0;
      __syncthreads();
      for (uint32_t i = (uint32_t)tid; i < m; i += kThreads) {
        const uint32_t c = G.ord[i], s = G.sz[c];
        if (s > 1u && !G.nz[c]) for (uint32_t j = i + 1u; j < i + s; j++) G.pflag[j] = 1;
      }
      __syncthreads();
    }
    MPF_GMARK(3);
    // ---- (5) decide
    if (wave == 0) {
      const bool rnd = P.tie_mode == (uint32_t)MPF_TIE_RANDOM;
      const uint32_t len = sh.len;
      uint32_t best = 0x7FFFFFFFu, sel = 0xFFFFFFFFu;
      unsigned long long rng = sh.rng, hits = 1ull, draws = sh.draws;
      for (uint32_t base = 0; base < m; base += 64u) {
        const uint32_t i = base + (uint32_t)lane;
        uint32_t v = 0xFFFFFFFFu;
        if (i < m && !(cut_any && G.pflag[i])) v = len + G.cost[i];
        unsigned long long mask = __ballot((int)(v <= best));
        while (mask) {
          const int l = __builtin_ctzll(mask);
          mask &= mask - 1ull;
          const uint32_t x = (uint32_t)__builtin_amdgcn_readlane((int)v, l);
          if (x > best) continue;                            // best has fallen meanwhile
          bool take;
          if (rnd) {
            if (x < best) hits = 1ull; else hits++;
            take = x < best;
            if (!take) { draws++; take = g_tie_draw(rng) <= 1.0 / (double)hits; }
          } else {
            take = x < best;
          }
          if (take) { best = x; sel = base + (uint32_t)l; }
        }
      }
      if (lane == 0) {
        sh.rng = rng; sh.draws = draws; sh.best = best; sh.ins_pos = sel;
        if (sel >= m) sh.err = 5u;
        else if (tile == 0) { P.out[2u * step] = (uint32_t)G.dcid[G.ord[sel]]; P.out[2u * step + 1u] = best; }
      }
    }
    __syncthreads();
    if (sh.err) break;
    MPF_GMARK(4);
    // ---- (6) insert: node x (down record = third record of the new inner node) takes the branch's place, its children are
    // the new tip (visited first: hookup(p, q), hookup(q->next, insert), hookup(q->next->next, r), :3158-3171) and the old node
    const uint32_t Pp = sh.ins_pos;
    const uint32_t cs = G.ord[Pp];
    const uint32_t szc = G.sz[cs], depc = G.dep[cs], pp = G.par[cs];
    const uint32_t xnode = (uint32_t)P.xnode[step], tnode = (uint32_t)P.tips[step];
    {
      constexpr int kPer = 8;                      // (m <= 8 * threads: grow_supported)
      uint32_t cc[kPer], fl[kPer];
#pragma unroll
      for (int k = 0; k < kPer; k++) {
        const uint32_t i = (uint32_t)tid + (uint32_t)k * kThreads;
        cc[k] = 0; fl[k] = 0;
        if (i < m) {
          const uint32_t c = G.ord[i];
          cc[k] = c;
          fl[k] = 1u | (i >= Pp ? 2u : 0u) | ((i >= Pp && i < Pp + szc) ? 4u : 0u) | ((i < Pp && i + (uint32_t)G.sz[c] > Pp) ? 8u : 0u);
        }
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < kPer; k++) {
        const uint32_t i = (uint32_t)tid + (uint32_t)k * kThreads;
        if (fl[k] & 1u) {
          const uint32_t c = cc[k];
          if (fl[k] & 2u) { G.ord[i + 2u] = (uint16_t)c; G.pos[c] = (uint16_t)(i + 2u); }
          if (fl[k] & 4u) G.dep[c] = (uint16_t)(G.dep[c] + 1u);
          if (fl[k] & 8u) G.sz[c] = (uint16_t)(G.sz[c] + 2u);
        }
      }
      __syncthreads();
      if (tid == 0) {
        G.ord[Pp] = (uint16_t)xnode; G.ord[Pp + 1u] = (uint16_t)tnode;
        G.pos[xnode] = (uint16_t)Pp; G.pos[tnode] = (uint16_t)(Pp + 1u);
        G.sz[xnode] = (uint16_t)(szc + 2u); G.sz[tnode] = 1;
        G.dep[xnode] = (uint16_t)depc; G.dep[tnode] = (uint16_t)(depc + 1u);
        G.par[xnode] = (uint16_t)pp; G.par[tnode] = (uint16_t)xnode; G.par[cs] = (uint16_t)xnode;
        G.ch1[xnode] = (uint16_t)tnode; G.ch2[xnode] = (uint16_t)cs;
        G.ch1[tnode] = (uint16_t)kGrowNone; G.ch2[tnode] = (uint16_t)kGrowNone;
        G.dcid[xnode] = P.xdcid[step]; G.dcid[tnode] = (uint16_t)tnode;
        G.nz[tnode] = 0; G.nz[xnode] = 0;
        if (pp == kGrowNone) sh.root = xnode;
        else if (G.ch1[pp] == cs) G.ch1[pp] = (uint16_t)xnode;
        else G.ch2[pp] = (uint16_t)xnode;
        sh.m = m + 2u; sh.len = sh.best; sh.step = step + 1u;
      }
      __syncthreads();
    }
    MPF_GMARK(5);
    const bool last = step + 1u >= P.steps;
    // ---- (7) path: D(x) = fitch(tip, D(old node)), then every ancestor from its path child (registers) and its other child
    if (wave == 0 && !last) {
      QT<KS, VW> cur, nw, sib[PF];
      {
        QT<KS, VW> dc;
        g_ld<KS, VW>(G, dc, rfl((uint32_t)G.dcid[cs]));
        const uint32_t cnt = v_fitch<KS, VW>(cur, Tv, dc);
        g_st<KS, VW>(G, cur, rfl((uint32_t)G.dcid[xnode]));
        const bool any = __ballot((int)(G.cnt_lane && cnt > 0u)) != 0ull;
        if (lane == 0) { G.path[0] = (uint16_t)xnode; G.psib[0] = (uint16_t)cs; G.pflag[0] = any ? 1 : 0; }
      }
      uint32_t k = 1, prev = xnode, a = pp;
      while (a != kGrowNone) {
        const uint32_t c1 = rfl((uint32_t)G.ch1[a]), c2 = rfl((uint32_t)G.ch2[a]);
        if (lane == 0) { G.path[k] = (uint16_t)a; G.psib[k] = (uint16_t)(c1 == prev ? c2 : c1); }
        prev = a;
        a = rfl((uint32_t)G.par[a]);
        k++;
      }
      for (uint32_t blk = 1; blk < k; blk += (uint32_t)PF) {
#pragma unroll
        for (int i = 0; i < PF; i++) {
          uint32_t q = blk + (uint32_t)i;
          q = q < k ? q : k - 1u;
          g_ld<KS, VW>(G, sib[i], rfl((uint32_t)G.dcid[rfl((uint32_t)G.psib[q])]));
        }
#pragma unroll
        for (int i = 0; i < PF; i++) {
          const uint32_t q = blk + (uint32_t)i;
          if (q < k) {
            const uint32_t cnt = v_fitch<KS, VW>(nw, cur, sib[i]);
            g_st<KS, VW>(G, nw, rfl((uint32_t)G.dcid[rfl((uint32_t)G.path[q])]));
            const bool any = __ballot((int)(G.cnt_lane && cnt > 0u)) != 0ull;
            if (lane == 0) G.pflag[q] = any ? 1 : 0;
            cur = nw;
          }
        }
      }
      if (lane == 0) sh.path_n = k;
    }
    __syncthreads();
    MPF_GMARK(6);
    if (last) { if (tid == 0) sh.exit_reason = GROW_DONE; __syncthreads(); break; }
  }
  // ---- hand the state back
  if (tile == 0 && tid == 0) {
    GrowHeader *h = P.hdr;
    h->rng = sh.rng; h->draws = sh.draws; h->steps_done = sh.step; h->err = sh.err; h->len = sh.len;
    h->reason = sh.err ? GROW_ERROR : sh.exit_reason;
    for (int i = 0; i < 8; i++) h->tph[i] = sh.tph[i];
  }
}

template <int KS, int VW>
__global__ __launch_bounds__((GCfg<KS, VW>::NT)) void k_grow(GrowParams P) { grow_body<KS, VW>(P); }
// DNA tiles: held to 128 registers, so that TWO workgroups (two trees) share a CU -- one tree's waves alone leave its SIMDs idle
// half of the time (one wave walks the skeleton, the parts wait for memory)
template <int KS, int VW>
__global__ __launch_bounds__((GCfg<KS, VW>::NT)) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_grow_tight(GrowParams P) { grow_body<KS, VW>(P); }

template <int KS, int VW>
size_t g_lds_bytes(uint32_t n)
{
  const size_t N2 = 2 * (size_t)n;
  size_t at = (sizeof(GSh) + 15) & ~(size_t)15;
  at += 12 * ((N2 * 2 + 15) & ~(size_t)15);
  at += N2 * 4;
  at += (((N2 / 32 + 2) * 4) + 15) & ~(size_t)15;
  at += (N2 + 15) & ~(size_t)15;
  at += (N2 + 15) & ~(size_t)15;
  at += (size_t)GCfg<KS, VW>::NW * 64 * sizeof(uint4);
  return (at + 15) & ~(size_t)15;
}

template <int KS, int VW>
constexpr bool g_tight() { return (KS == 1 && VW <= 4) || KS == 4; }

template <int KS, int VW>
const void *g_kernel()
{
  // (one of the two per shape: the other would be compiled for nothing -- and the tight form of the wide shapes spills)
  if constexpr (g_tight<KS, VW>()) return reinterpret_cast<const void *>(&k_grow_tight<KS, VW>);
  else return reinterpret_cast<const void *>(&k_grow<KS, VW>);
}

template <int KS, int VW>
hipError_t g_launch(hipStream_t st, const GrowParams &p)
{
  const size_t lds = g_lds_bytes<KS, VW>(p.n);
  static thread_local int attr_dev = -1;
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (lds > 64 * 1024 || attr_dev != dev) {
    hipError_t e = hipFuncSetAttribute(g_kernel<KS, VW>(), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_dev = dev;
  }
  if constexpr (g_tight<KS, VW>()) hipLaunchKernelGGL((k_grow_tight<KS, VW>), dim3(p.tiles), dim3(GCfg<KS, VW>::NT), lds, st, p);
  else hipLaunchKernelGGL((k_grow<KS, VW>), dim3(p.tiles), dim3(GCfg<KS, VW>::NT), lds, st, p);
  return hipGetLastError();
}

// workgroups of this launch shape that fit one CU together (registers, LDS): what the admission gate counts in
template <int KS, int VW>
int g_blocks_per_cu(uint32_t n)
{
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, g_kernel<KS, VW>(), GCfg<KS, VW>::NT, g_lds_bytes<KS, VW>(n)) != hipSuccess || nb < 1) nb = 1;
  return nb;
}

}  // namespace

// vw = 0: the word-major DNA layout (64 words per tile, a lane = one word with its four states)
int grow_tiles(const Geometry &g, int vw) { return vw == 0 ? (g.Wp + 63) / 64 : (g.Wp + 16 * vw - 1) / (16 * vw); }

int grow_waves(const Geometry &g, int vw)
{
  if (vw == 0) return 8;
  const int r = (g.S == 4 ? 1 : g.S == 32 ? 8 : 5) * vw;
  return r <= 2 ? 16 : 8;
}

size_t grow_vec_words(const Geometry &g, int vw) { return vw == 0 ? 256 : (size_t)(g.S == 4 ? 1 : g.S == 32 ? 8 : 5) * (size_t)vw * 64; }

size_t grow_lds_bytes(const Geometry &g, int n_taxa, int vw)
{
  (void)g; (void)vw;
  return g_lds_bytes<1, 1>((uint32_t)n_taxa);          // (the control arrays do not depend on the tile shape)
}

bool grow_supported(const Geometry &g, int n_taxa)
{
  if (g.sankoff || g.big) return false;
  if (g.S != 4 && g.S != 20 && g.S != 32) return false;
  if (n_taxa < 4 || 2 * (size_t)n_taxa > 8 * 512) return false;       // (the insertion moves at most 8 entries per thread)
  if ((uint32_t)n_taxa + 3u * (uint32_t)(n_taxa - 1) + 16u >= 0xFFFFu) return false;
  return grow_lds_bytes(g, n_taxa, 1) <= 150 * 1024;
}

int grow_blocks_per_cu(const Geometry &g, int n_taxa, int vw)
{
  if (g.S == 4 && vw == 0) return g_blocks_per_cu<4, 1>((uint32_t)n_taxa);
  if (g.S == 4) return vw == 1 ? g_blocks_per_cu<1, 1>((uint32_t)n_taxa) : vw == 2 ? g_blocks_per_cu<1, 2>((uint32_t)n_taxa)
                       : vw == 8 ? g_blocks_per_cu<1, 8>((uint32_t)n_taxa) : g_blocks_per_cu<1, 4>((uint32_t)n_taxa);
  if (g.S == 32) return g_blocks_per_cu<8, 1>((uint32_t)n_taxa);
  return g_blocks_per_cu<5, 1>((uint32_t)n_taxa);
}

hipError_t launch_grow(hipStream_t st, const Geometry &g, int vw, const GrowParams &p)
{
  if (g.S == 4) {
    if (vw == 0) return g_launch<4, 1>(st, p);        // (p.vec = the word-major copy)
    if (vw == 1) return g_launch<1, 1>(st, p);
    if (vw == 2) return g_launch<1, 2>(st, p);
    if (vw == 8) return g_launch<1, 8>(st, p);
    return g_launch<1, 4>(st, p);
  }
  if (g.S == 32) return g_launch<8, 1>(st, p);
  return g_launch<5, 1>(st, p);
}

}  // namespace mpf
