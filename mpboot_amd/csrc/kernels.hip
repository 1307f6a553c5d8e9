// kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the Fitch engine.
//
// Everything here is bitwise integer work bounded by memory traffic (no MFMA):
//   fitch(a,b):  t_k = a_k & b_k ; N = ~OR_k t_k ; c_k = t_k | (N & (a_k | b_k)) ; cost = popcount(N)
// the arithmetic of newviewParsimonyIterativeFast (reference sprparsimony.cpp:737-776, :841-869)
// and evaluateParsimonyIterativeFast (:1108-1124, :1178-1203).
//
// Work decomposition: sites are independent, so a wavefront owns a TILE of 64*VW
// consecutive words of every state row and never needs another wave's data; the second
// grid dimension is the batch (newview ops of one dependency level, or SPR scans).
// Per-candidate mutation counts are reduced across the 64 lanes with DPP row operations
// and leave the wave as ONE integer atomic (integer adds commute: results are exact and
// run-to-run identical).
#include "kernels.hpp"
#include <cstdlib>

namespace mpf {

// ---------------------------------------------------------------- wave-level helpers

// Sum over the 64 lanes of a wavefront, result valid in lane 63 (DPP) -- the classic GCN
// reduction: xor-1, xor-2 inside quads, half-row mirror, row mirror, then row broadcasts.
__device__ __forceinline__ uint32_t wave_sum_dpp(uint32_t v)
{
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, false);  // row_half_mirror
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, false);  // row_mirror
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);  // row_bcast:15 -> rows 1,3
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);  // row_bcast:31 -> rows 2,3
  return v;
}

// sum over aligned groups of TW lanes (4, 8, 16: inside a DPP row; 32: two rows), left in every lane of the group
template <int TW>
__device__ __forceinline__ uint32_t group_sum(uint32_t v)
{
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
  if constexpr (TW >= 8) v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, false);  // row_half_mirror
  if constexpr (TW >= 16) v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, false);  // row_mirror
  if constexpr (TW == 32) v += (uint32_t)__shfl_xor((int)v, 16, 32);
  return v;
}

template <int RED>
__device__ __forceinline__ uint32_t wave_total(uint32_t v)
{
  if constexpr (RED == 0) {
    v = wave_sum_dpp(v);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
  } else {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += (uint32_t)__shfl_xor((int)v, off, 64);
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
  }
}

__device__ __forceinline__ void atomic_add_u32(uint32_t *p, uint32_t v)
{
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------- vector tiles in registers

template <int S, int VW>
struct Tile {
  uint32_t v[S][VW];
};

template <int S, int VW>
__device__ __forceinline__ void load_tile(Tile<S, VW> &t, const uint32_t *__restrict__ vec, uint32_t slot, int Wp, int w0)
{
  const uint32_t *p = vec + (size_t)slot * (size_t)(S * Wp) + w0;
#pragma unroll
  for (int k = 0; k < S; k++) {
    if constexpr (VW == 1) {
      t.v[k][0] = p[(size_t)k * Wp];
    } else if constexpr (VW == 2) {
      uint2 x = *reinterpret_cast<const uint2 *>(p + (size_t)k * Wp);
      t.v[k][0] = x.x; t.v[k][1] = x.y;
    } else {
      uint4 x = *reinterpret_cast<const uint4 *>(p + (size_t)k * Wp);
      t.v[k][0] = x.x; t.v[k][1] = x.y; t.v[k][2] = x.z; t.v[k][3] = x.w;
    }
  }
}

template <int S, int VW>
__device__ __forceinline__ void store_tile(const Tile<S, VW> &t, uint32_t *__restrict__ vec, uint32_t slot, int Wp, int w0)
{
  uint32_t *p = vec + (size_t)slot * (size_t)(S * Wp) + w0;
#pragma unroll
  for (int k = 0; k < S; k++) {
    if constexpr (VW == 1) {
      p[(size_t)k * Wp] = t.v[k][0];
    } else if constexpr (VW == 2) {
      *reinterpret_cast<uint2 *>(p + (size_t)k * Wp) = make_uint2(t.v[k][0], t.v[k][1]);
    } else {
      *reinterpret_cast<uint4 *>(p + (size_t)k * Wp) = make_uint4(t.v[k][0], t.v[k][1], t.v[k][2], t.v[k][3]);
    }
  }
}

// gfx950 has a three-input bitwise op (v_bitop3_b32); truth table with src0 = 0xF0, src1 = 0xCC, src2 = 0xAA
#define MPF_B3_ANDOR 0xEA   // (a & b) | c
#define MPF_B3_FITCH 0xD4   // c ? (a & b) : (a | b)
__device__ __forceinline__ uint32_t b3_andor(uint32_t a, uint32_t b, uint32_t c)
{
  return (uint32_t)__builtin_amdgcn_bitop3_b32((int)a, (int)b, (int)c, MPF_B3_ANDOR);
}
__device__ __forceinline__ uint32_t b3_fitch(uint32_t a, uint32_t b, uint32_t any)
{
  return (uint32_t)__builtin_amdgcn_bitop3_b32((int)a, (int)b, (int)any, MPF_B3_FITCH);
}

// c = fitch(a, b); returns the number of sites of this lane's words whose intersection is empty.
// any = OR_k(a_k & b_k) as an and-or chain, then c_k = any ? a_k & b_k : a_k | b_k: 2 ops per state.
// SPLIT: the states of a word are spread over lanes l and l^32 (protein: 10 + 10), the two partial `any`
// words are exchanged across the wave halves.
template <int S, int VW, bool SPLIT = false>
__device__ __forceinline__ uint32_t fitch(Tile<S, VW> &c, const Tile<S, VW> &a, const Tile<S, VW> &b)
{
  uint32_t cost = 0;
#pragma unroll
  for (int j = 0; j < VW; j++) {
    uint32_t any = a.v[0][j] & b.v[0][j];
#pragma unroll
    for (int k = 1; k < S; k++) any = b3_andor(a.v[k][j], b.v[k][j], any);
    if constexpr (SPLIT) any |= (uint32_t)__shfl_xor((int)any, 32, 64);
#pragma unroll
    for (int k = 0; k < S; k++) c.v[k][j] = b3_fitch(a.v[k][j], b.v[k][j], any);
    cost += (uint32_t)__builtin_popcount(~any);
  }
  return cost;
}

// popcount(~OR_k(a_k & b_k))
template <int S, int VW>
__device__ __forceinline__ uint32_t empty_count(const Tile<S, VW> &a, const Tile<S, VW> &b)
{
  uint32_t cost = 0;
#pragma unroll
  for (int j = 0; j < VW; j++) {
    uint32_t any = a.v[0][j] & b.v[0][j];
#pragma unroll
    for (int k = 1; k < S; k++) any = b3_andor(a.v[k][j], b.v[k][j], any);
    cost += (uint32_t)__builtin_popcount(~any);
  }
  return cost;
}

// cost of joining subtree vector s onto the node x = fitch(u, d):  popcount(~OR_k(x_k & s_k)),
// x_k formed in registers and consumed at once (3 ops per state + 2)
template <int S, int VW, bool SPLIT = false>
__device__ __forceinline__ uint32_t join_cost(const Tile<S, VW> &u, const Tile<S, VW> &d, const Tile<S, VW> &s)
{
  uint32_t cost = 0;
#pragma unroll
  for (int j = 0; j < VW; j++) {
    uint32_t any = u.v[0][j] & d.v[0][j];
#pragma unroll
    for (int k = 1; k < S; k++) any = b3_andor(u.v[k][j], d.v[k][j], any);
    if constexpr (SPLIT) any |= (uint32_t)__shfl_xor((int)any, 32, 64);
    uint32_t hit = b3_fitch(u.v[0][j], d.v[0][j], any) & s.v[0][j];
#pragma unroll
    for (int k = 1; k < S; k++) hit = b3_andor(b3_fitch(u.v[k][j], d.v[k][j], any), s.v[k][j], hit);
    if constexpr (SPLIT) hit |= (uint32_t)__shfl_xor((int)hit, 32, 64);
    cost += (uint32_t)__builtin_popcount(~hit);
  }
  return cost;
}

// same as join_cost, also handing back the "no common state" words themselves (UFBoot masks)
template <int S, int VW, bool SPLIT = false>
__device__ __forceinline__ uint32_t join_mask(const Tile<S, VW> &u, const Tile<S, VW> &d, const Tile<S, VW> &s, uint32_t (&m)[VW])
{
  uint32_t cost = 0;
#pragma unroll
  for (int j = 0; j < VW; j++) {
    uint32_t any = u.v[0][j] & d.v[0][j];
#pragma unroll
    for (int k = 1; k < S; k++) any = b3_andor(u.v[k][j], d.v[k][j], any);
    if constexpr (SPLIT) any |= (uint32_t)__shfl_xor((int)any, 32, 64);
    uint32_t hit = b3_fitch(u.v[0][j], d.v[0][j], any) & s.v[0][j];
#pragma unroll
    for (int k = 1; k < S; k++) hit = b3_andor(b3_fitch(u.v[k][j], d.v[k][j], any), s.v[k][j], hit);
    if constexpr (SPLIT) hit |= (uint32_t)__shfl_xor((int)hit, 32, 64);
    m[j] = ~hit;
    cost += (uint32_t)__builtin_popcount(~hit);
  }
  return cost;
}

// tile index -> first word of this lane; lanes past the row end are clamped onto the last
// valid group (they load real data but contribute nothing), so EXEC stays full for the DPP ops
template <int VW>
__device__ __forceinline__ int lane_word(int tile, int lane, int Wp, bool &valid)
{
  int w0 = (tile * 64 + lane) * VW;
  valid = w0 < Wp;
  return valid ? w0 : Wp - VW;
}

// ---------------------------------------------------------------- K7: tip packing (compressDNA)

// state set of a PLL tip code: DNA bitVectorIdentity, protein bitVectorAA
// (reference pllrepo/src/globalVariables.h:60-78)
// (binary: bitVectorIdentity like DNA, codes 1..3; 32-state: bitVector32, code 32 = every state -- the engine carries the
//  first 20 rows of such data, globalVariables.h:98-102)
__device__ __forceinline__ uint32_t state_mask(int datatype, uint32_t code)
{
  if (datatype == 0 || datatype == 2) return code;
  if (datatype == 3) return code < 32u ? 1u << code : 0xFFFFFFFFu;      // (20-row packing: the rows that exist)
  if (code < 20u) return 1u << code;
  if (code == 20u) return 12u;
  if (code == 21u) return 96u;
  return 1048575u;
}

// One wavefront = (tip, 64 words = 2048 expanded sites).  64 lanes read 64 consecutive sites at a time (site -> pattern
// map coalesced, tip codes gathered within a few cache lines), and one ballot per state turns the 64 membership bits
// into two finished words, kept by lanes 2c and 2c + 1; the 64 words of each state row leave as one 256-byte store.
template <int S>
__global__ __launch_bounds__(256) void k_pack_tips(uint32_t *__restrict__ vec, const uint8_t *__restrict__ codes,
                                                   int n_taxa, int n_patterns, const int32_t *__restrict__ site2ptn,
                                                   int n_sites, int datatype, const uint32_t *__restrict__ tip_slots,
                                                   int Wp, uint32_t *__restrict__ shadow)
{
  const int lane = threadIdx.x & 63;
  const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int tip = blockIdx.y;
  if (tile * 64 >= Wp || tip >= n_taxa) return;          // wave-uniform
  uint32_t val[S];
#pragma unroll
  for (int k = 0; k < S; k++) val[k] = 0;
  const uint8_t *row = codes + (size_t)tip * n_patterns;
  const int site0 = tile * 64 * 32;
  // the two dependent gathers (site -> pattern -> code) of all 32 rounds are issued together: the kernel is bound by their
  // latency, not by the ballots
  int ptn[32];
  uint8_t code[32];
#pragma unroll
  for (int c = 0; c < 32; c++) {
    const int site = site0 + 64 * c + lane;
    ptn[c] = site < n_sites ? site2ptn[site] : -1;
  }
#pragma unroll
  for (int c = 0; c < 32; c++) code[c] = ptn[c] >= 0 ? row[ptn[c]] : (uint8_t)0;
#pragma unroll
  for (int c = 0; c < 32; c++) {
    // expanded sites beyond the alignment are all-ones in every state row so that they
    // never count (reference sprparsimony.cpp:2947-2960)
    uint32_t m = 0xFFFFFFFFu;
    if (ptn[c] >= 0) m = state_mask(datatype, code[c]);
#pragma unroll
    for (int k = 0; k < S; k++) {
      const unsigned long long bal = __ballot((int)((m >> k) & 1u));
      const uint32_t lo = (uint32_t)bal, hi = (uint32_t)(bal >> 32);
      val[k] = lane == 2 * c ? lo : (lane == 2 * c + 1 ? hi : val[k]);
    }
  }
  const int w = tile * 64 + lane;
  if (w < Wp) {
    uint32_t *dst = vec + (size_t)tip_slots[tip] * (size_t)(S * Wp) + w;
#pragma unroll
    for (int k = 0; k < S; k++) dst[(size_t)k * Wp] = val[k];
    if constexpr (S == 4)
      if (shadow) *reinterpret_cast<uint4 *>(shadow + ((size_t)tip_slots[tip] * (size_t)Wp + (size_t)w) * 4) = make_uint4(val[0], val[1], val[2], val[3]);
  }
}

// ---------------------------------------------------------------- K1: batched newview
//
// Per-(op, tile) mutation counts go to cntp[tile][dst] with plain stores (no zeroing, no atomics);
// k_cntsum folds the tiles afterwards.

template <int S, int VW, int RED>
__device__ __forceinline__ void newview_one(uint32_t *__restrict__ vec, const NvOp o, uint32_t *__restrict__ cntp,
                                            uint32_t nslots, int Wp, int tile, int lane)
{
  bool valid;
  const int w0 = lane_word<VW>(tile, lane, Wp, valid);
  Tile<S, VW> a, b, c;
  load_tile<S, VW>(a, vec, o.a, Wp, w0);
  load_tile<S, VW>(b, vec, o.b, Wp, w0);
  uint32_t cost = fitch<S, VW>(c, a, b);
  if (valid) store_tile<S, VW>(c, vec, o.dst, Wp, w0);
  cost = valid ? cost : 0u;
  const uint32_t tot = wave_total<RED>(cost);
  if (lane == 0) cntp[(size_t)tile * nslots + o.dst] = tot;
}

// one dependency level per launch: grid = ops x tiles waves
template <int S, int VW, int RED>
__global__ __launch_bounds__(256) void k_newview(uint32_t *__restrict__ vec, const NvOp *__restrict__ ops, int n_ops,
                                                 uint32_t *__restrict__ cntp, uint32_t nslots, int Wp, int tiles)
{
  const int lane = threadIdx.x & 63;
  int gw = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  gw = __builtin_amdgcn_readfirstlane(gw);
  if (gw >= n_ops * tiles) return;
  const int op = gw / tiles, tile = gw - op * tiles;
  newview_one<S, VW, RED>(vec, ops[op], cntp, nslots, Wp, tile, lane);
}

// cnt[dst] = sum over tiles of cntp[tile][dst] for the ops of a refresh: 32 lanes per op (one tile each, strided), two ops
// per group in flight.  The partial counts were written by other workgroups (other XCDs): the caller has acquired them.
__device__ __forceinline__ void fold_counts(const NvOp *__restrict__ ops, int n_ops, const uint32_t *__restrict__ cntp,
                                            uint32_t nslots, int tiles, uint32_t *__restrict__ cnt, int tid, int nthreads,
                                            uint32_t *__restrict__ cnt_host = nullptr)
{
  const int l32 = tid & 31, grp = tid >> 5, ngrp = nthreads >> 5;
  for (int i = grp; i < n_ops; i += 2 * ngrp) {
    const int j = i + ngrp < n_ops ? i + ngrp : i;
    const uint32_t d0 = ops[i].dst, d1 = ops[j].dst;
    uint32_t s0 = 0, s1 = 0;
    for (int t = l32; t < tiles; t += 32) {
      s0 += __builtin_nontemporal_load(cntp + (size_t)t * nslots + d0);
      s1 += __builtin_nontemporal_load(cntp + (size_t)t * nslots + d1);
    }
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) {
      s0 += (uint32_t)__shfl_xor((int)s0, m, 32);
      s1 += (uint32_t)__shfl_xor((int)s1, m, 32);
    }
    if (l32 == 0) {
      cnt[d0] = s0;
      if (j != i) cnt[d1] = s1;
      if (cnt_host) {                              // the host's copy directly (pinned memory): no copy-back dispatch
        cnt_host[d0] = s0;
        if (j != i) cnt_host[d1] = s1;
      }
    }
  }
}

// ALL levels in one launch: one 16-wave workgroup per tile walks the levels, the waves share a level's
// ops, and a workgroup barrier separates levels -- sites are independent, so no other workgroup's data
// is ever needed.  Replaces one launch per level (launch-latency-bound for the short levels of an
// incremental refresh).
// The per-tile mutation counts are folded by whichever workgroup finishes last (`done` counts finished workgroups and is
// left at zero again): no separate summation kernel behind every refresh.
template <int S, int VW, int RED>
__global__ __launch_bounds__(1024) void k_newview_wg(uint32_t *__restrict__ vec, const NvOp *__restrict__ ops,
                                                     const int32_t *__restrict__ lev_off, int n_lev,
                                                     uint32_t *__restrict__ cntp, uint32_t nslots, int Wp,
                                                     uint32_t *__restrict__ cnt, uint32_t *__restrict__ done,
                                                     RefreshExtra x)
{
  __shared__ int s_last;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int nw = (int)(blockDim.x >> 6);
  const int tile = blockIdx.x;
  bool valid;
  const int w0 = lane_word<VW>(tile, lane, Wp, valid);
  // clears the outputs of the scan launch that follows on the stream (no memset dispatch in front of it)
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < x.zero_words; i += gridDim.x * blockDim.x) x.zero_ptr[i] = 0u;
  for (int l = 0; l < n_lev; l++) {
    const int b = lev_off[l], e = lev_off[l + 1];
    int i = b + wave;
    // two independent ops in flight per wave: their four operand tiles are requested before either is combined
    for (; i + nw < e; i += 2 * nw) {
      const NvOp o0 = ops[i], o1 = ops[i + nw];
      Tile<S, VW> a0, b0, a1, b1, c0, c1;
      load_tile<S, VW>(a0, vec, o0.a, Wp, w0);
      load_tile<S, VW>(b0, vec, o0.b, Wp, w0);
      load_tile<S, VW>(a1, vec, o1.a, Wp, w0);
      load_tile<S, VW>(b1, vec, o1.b, Wp, w0);
      uint32_t k0 = fitch<S, VW>(c0, a0, b0);
      uint32_t k1 = fitch<S, VW>(c1, a1, b1);
      if (valid) { store_tile<S, VW>(c0, vec, o0.dst, Wp, w0); store_tile<S, VW>(c1, vec, o1.dst, Wp, w0); }
      const uint32_t packed = valid ? (k0 | (k1 << 16)) : 0u;       // <= 64*32*VW per wave each: fits 16 bits
      const uint32_t tot = wave_total<RED>(packed);
      if (lane == 0) {
        cntp[(size_t)tile * nslots + o0.dst] = tot & 0xFFFFu;
        cntp[(size_t)tile * nslots + o1.dst] = tot >> 16;
      }
    }
    if (i < e) newview_one<S, VW, RED>(vec, ops[i], cntp, nslots, Wp, tile, lane);
    __syncthreads();
  }
  // ---- fold the per-tile counts: every workgroup publishes its stores (agent-scope release), takes a ticket; the
  //      last one acquires and sums
  if (!done) return;                              // large refresh: launch_cntsum folds the counts with the whole chip
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    const uint32_t ticket = __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = ticket == gridDim.x - 1;
  }
  __syncthreads();
  if (s_last) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    fold_counts(ops, lev_off[n_lev], cntp, nslots, (int)gridDim.x, cnt, (int)threadIdx.x, (int)blockDim.x, x.cnt_host);
    if (threadIdx.x == 0) __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// The same with HALF a wave per op (one word per lane tiles only): a tile is 32 words, lanes 0-31 work on one op of the
// level and lanes 32-63 on another.  Twice as many workgroups -- a refresh of most of the tree keeps only Wp/64 (25 at
// C3) of the 256 CUs busy otherwise -- and half as many rounds per level.
template <int S, int RED>
__global__ __launch_bounds__(1024) void k_newview_wgh(uint32_t *__restrict__ vec, const NvOp *__restrict__ ops,
                                                      const int32_t *__restrict__ lev_off, int n_lev,
                                                      uint32_t *__restrict__ cntp, uint32_t nslots, int Wp,
                                                      uint32_t *__restrict__ cnt, uint32_t *__restrict__ done, RefreshExtra x)
{
  __shared__ int s_last;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int nw = (int)(blockDim.x >> 6);
  const int tile = blockIdx.x;
  const int half = lane >> 5;
  const int w0 = tile * 32 + (lane & 31);          // Wp is a multiple of 32: always inside the row
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < x.zero_words; i += gridDim.x * blockDim.x) x.zero_ptr[i] = 0u;
  for (int l = 0; l < n_lev; l++) {
    const int b = lev_off[l], e = lev_off[l + 1];
    for (int ib = b + 2 * wave; ib < e; ib += 2 * nw) {
      const bool active = ib + half < e;
      const NvOp o = ops[active ? ib + half : ib];
      Tile<S, 1> ta, tb, tc;
      load_tile<S, 1>(ta, vec, o.a, Wp, w0);
      load_tile<S, 1>(tb, vec, o.b, Wp, w0);
      uint32_t k = fitch<S, 1>(tc, ta, tb);
      if (active) store_tile<S, 1>(tc, vec, o.dst, Wp, w0);
#pragma unroll
      for (int m = 16; m >= 1; m >>= 1) k += (uint32_t)__shfl_xor((int)k, m, 32);
      if ((lane & 31) == 0 && active) cntp[(size_t)tile * nslots + o.dst] = k;
    }
    __syncthreads();
  }
  if (!done) return;
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    const uint32_t ticket = __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = ticket == gridDim.x - 1;
  }
  __syncthreads();
  if (s_last) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    fold_counts(ops, lev_off[n_lev], cntp, nslots, (int)gridDim.x, cnt, (int)threadIdx.x, (int)blockDim.x, x.cnt_host);
    if (threadIdx.x == 0) __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// The same with TW lanes per op (TW = 32, 16, 8: tiles of TW words, 64 / TW ops of the level per wave and round) and the
// next round's operands requested BEFORE this round's results are stored.  Finer tiles = more workgroups (a full refresh at
// C3: 49, 98, 196 of them on 256 CUs) with no barrier between them -- a tile of sites never needs another tile's vectors.
// The early request matters because gfx950 retires loads and stores through ONE in-order counter: a wave that asks for its
// next operands only after its stores cannot see them before the stores are acknowledged.
// (defined with k_walk_plan below) one wave = one (scan part, gap end) item of the walk plan
struct ProgEnt;
__device__ void walk_plan_item(const uint2 *__restrict__ kids, uint32_t n, const WalkDesc *__restrict__ desc, int n_scans,
                               ProgEnt *__restrict__ prog, uint32_t cid_mask, int item, int lane);

template <int S, int TW>
__global__ __launch_bounds__(1024) void k_newview_wgq(uint32_t *__restrict__ vec, const NvOp *__restrict__ ops,
                                                      const int32_t *__restrict__ lev_off, int n_lev,
                                                      uint32_t *__restrict__ cntp, uint32_t nslots, int Wp,
                                                      uint32_t *__restrict__ cnt, uint32_t *__restrict__ done, RefreshExtra x)
{
  constexpr int OPI = 64 / TW;                     // ops per wave and round
  __shared__ int s_last;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int nw = (int)(blockDim.x >> 6);
  const int nt = Wp / TW;                          // the refresh's own workgroups; any further ones plan the scan that follows
  if ((int)blockIdx.x >= nt) {
    const uint32_t n_parts = x.wp_hdr[0], n_out = x.wp_hdr[1];
    const uint32_t xb = blockIdx.x - (uint32_t)nt, nxb = gridDim.x - (uint32_t)nt;
    const uint32_t zw = ((n_out ? n_out : 1u) + 1u + 63u) & ~63u;      // (Engine::clear_words)
    for (uint32_t i = xb * blockDim.x + threadIdx.x; i < zw; i += nxb * blockDim.x) x.wp_out[i] = 0u;
    for (uint32_t item = xb * (uint32_t)nw + (uint32_t)wave; item < 2u * n_parts; item += nxb * (uint32_t)nw)
      walk_plan_item(x.wp_kids, x.wp_n, x.wp_desc, (int)n_parts, static_cast<ProgEnt *>(x.wp_prog), 0xFFFFFFFFu, (int)item, lane);
    return;
  }
  // workgroups go round the 8 XCDs: each XCD gets a contiguous run of tiles, so that the 64-byte segments of neighbouring
  // tiles -- two halves of one 128-byte line -- meet in ONE L2 instead of being fetched into two
  const int xcd = (int)(blockIdx.x & 7u), q8 = nt >> 3, r8 = nt & 7;
  const int tile = xcd * q8 + min(xcd, r8) + (int)(blockIdx.x >> 3);
  const int sub = lane / TW;
  const int w0 = tile * TW + (lane % TW);          // Wp is a multiple of 32: always inside the row
  const int step = OPI * nw;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < x.zero_words; i += (uint32_t)nt * blockDim.x) x.zero_ptr[i] = 0u;
  if (x.n_lev_ptr) n_lev = *x.n_lev_ptr;         // (schedule made by k_sched in front of this launch)
  for (int l = 0; l < n_lev; l++) {
    const int b = lev_off[l], e = lev_off[l + 1];
    int ib = b + OPI * wave;
    if (ib < e) {
      // descriptors run two rounds ahead, operands one.  Indices are clamped into the level: surplus lanes repeat its last op on
      // the same words as the lanes that own it and write the same values -- no lane is switched off, so every round issues
      // the same requests (1 descriptor, 2 S operand rows, S result rows, 1 count) and the wait for the next operands can
      // be counted past this round's stores
      NvOp o = ops[min(ib + sub, e - 1)], o1 = ops[min(ib + step + sub, e - 1)];
      Tile<S, 1> ta, tb;
      load_tile<S, 1>(ta, vec, o.a, Wp, w0);
      load_tile<S, 1>(tb, vec, o.b, Wp, w0);
      for (; ib < e; ib += step) {
        const NvOp o2 = ops[min(ib + 2 * step + sub, e - 1)];
        Tile<S, 1> na, nb, tc;
        load_tile<S, 1>(na, vec, o1.a, Wp, w0);
        load_tile<S, 1>(nb, vec, o1.b, Wp, w0);
        const uint32_t k = fitch<S, 1>(tc, ta, tb);
        store_tile<S, 1>(tc, vec, o.dst, Wp, w0);
        if constexpr (S == 4)
          if (x.shadow) *reinterpret_cast<uint4 *>(x.shadow + ((size_t)o.dst * (size_t)Wp + (size_t)w0) * 4) = make_uint4(tc.v[0][0], tc.v[1][0], tc.v[2][0], tc.v[3][0]);
        cntp[(size_t)tile * nslots + o.dst] = group_sum<TW>(k);
        ta = na; tb = nb; o = o1; o1 = o2;
      }
    }
    __syncthreads();
  }
  if (!done) return;
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    const uint32_t ticket = __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = ticket == (uint32_t)nt - 1u;
  }
  __syncthreads();
  if (s_last) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    fold_counts(ops, lev_off[n_lev], cntp, nslots, nt, cnt, (int)threadIdx.x, (int)blockDim.x, x.cnt_host);
    if (threadIdx.x == 0) __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// Chained refresh.  After a topology edit the stale vectors form out-trees fanning away from the edited nodes: nearly
// every stale vector has ONE stale input (the one from the edit's side) and one valid input.  The level-synchronous kernel
// above pays a store -> barrier -> load round trip for every link of such a path; here the host cuts the dependency graph
// into CHAINS (op k+1 takes op k's result and a vector that is already valid), a wave runs a chain with the running
// result in registers and the valid inputs prefetched two links ahead, and barriers separate only the (few) levels of
// the chain graph.  ops of (level l, wave w) = [wl_off[16 l + w], wl_off[16 l + w + 1]); o.a == kPrev = "previous result".
constexpr uint32_t kPrev = 0xFFFFFFFFu;
// Wide state sets (protein, 32-symbol data) keep D + 2 tiles of S registers each: at sixteen waves per workgroup (128 registers a
// lane) those spilled to scratch (round 5: 560-1300 bytes per lane).  They run EIGHT waves -- 256 registers --, each wave taking
// two of the sixteen chain slots of a level one after the other.
template <int S, int VW> constexpr int chain_waves() { return S * VW >= 16 ? 8 : 16; }
template <int S, int VW, int RED, int D>
__global__ __launch_bounds__((chain_waves<S, VW>() * 64)) void k_newview_chain(uint32_t *__restrict__ vec, const NvOp *__restrict__ ops,
                                                        const int32_t *__restrict__ wl_off, int n_lev,
                                                        uint32_t *__restrict__ cntp, uint32_t nslots, int Wp,
                                                        uint32_t *__restrict__ cnt, uint32_t *__restrict__ done, int n_ops,
                                                        RefreshExtra x)
{
  __shared__ int s_last;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int tile = blockIdx.x;
  bool valid;
  const int w0 = lane_word<VW>(tile, lane, Wp, valid);
  {
    // chores for the scan launch that follows on the stream, so that it needs neither a copy nor a memset dispatch in
    // front of it: topology updates for the device-walked scan (this kernel does not read kids) and the cleared outputs
    const int gt = (int)(blockIdx.x * blockDim.x + threadIdx.x), gn = (int)(gridDim.x * blockDim.x);
    for (int i = gt; i < x.n_kid_upd; i += gn) x.kids[x.kid_upd[3 * i]] = make_uint2(x.kid_upd[3 * i + 1], x.kid_upd[3 * i + 2]);
    for (uint32_t i = (uint32_t)gt; i < x.zero_words; i += (uint32_t)gn) x.zero_ptr[i] = 0u;
  }
  for (int l = 0; l < n_lev; l++) {
   for (int ws = wave; ws < 16; ws += chain_waves<S, VW>()) {
    const int b = wl_off[l * 16 + ws], e = wl_off[l * 16 + ws + 1];
    if (b < e) {
      // D register sets in rotation (no copies of in-flight registers): set d holds the operands of ops b + d, b + d + D, ...
      // requested D ops before they are combined; the op descriptors (scalar loads) run another D ahead
      // Wide state sets: only the operand nearly every op has -- the valid vector b -- rotates through D sets; the first operand of
      // a chain's HEAD (the one op per chain that does not continue the running result) is fetched when it is needed.  With a[D]
      // as well the 20-row kernel kept 52-76 bytes per lane in scratch even at 256 registers, the 32-row one 700.
      constexpr bool TA1 = S * VW >= 16;
      NvOp o[D], nx[D];
      Tile<S, VW> ta[TA1 ? 1 : D], tb[D], c;
#pragma unroll
      for (int d = 0; d < D; d++) o[d] = ops[b + d < e ? b + d : e - 1];
#pragma unroll
      for (int d = 0; d < D; d++) {                 // unconditional (indices are clamped): keeps the request counts static
        if constexpr (!TA1) { if (o[d].a != kPrev) load_tile<S, VW>(ta[d], vec, o[d].a, Wp, w0); }
        load_tile<S, VW>(tb[d], vec, o[d].b, Wp, w0);
      }
#pragma unroll
      for (int d = 0; d < D; d++) nx[d] = ops[b + D + d < e ? b + D + d : e - 1];
      // one op: combine set d, write the result, request the operands of the op D further on
#define MPF_CHAIN_STEP(d, RELOAD)                                                                        \
  {                                                                                                      \
    uint32_t cost;                                                                                       \
    if (o[d].a != kPrev) {                                                                               \
      if constexpr (TA1) load_tile<S, VW>(ta[0], vec, o[d].a, Wp, w0);                                   \
      cost = fitch<S, VW>(c, ta[TA1 ? 0 : d], tb[d]);                                                    \
    } else {                                                                                             \
      Tile<S, VW> p = c;                                                                                 \
      cost = fitch<S, VW>(c, p, tb[d]);                                                                  \
    }                                                                                                    \
    if (valid) store_tile<S, VW>(c, vec, o[d].dst, Wp, w0);                                              \
    if constexpr (S == 4 && VW == 1)                                                                     \
      if (valid && x.shadow)  /* the word-major copy the planned scan reads (Geometry::shoff) */         \
        *reinterpret_cast<uint4 *>(x.shadow + ((size_t)o[d].dst * (size_t)Wp + (size_t)w0) * 4) = make_uint4(c.v[0][0], c.v[1][0], c.v[2][0], c.v[3][0]); \
    const uint32_t tot = wave_total<RED>(valid ? cost : 0u);                                             \
    if (lane == 0) cntp[(size_t)tile * nslots + o[d].dst] = tot;                                         \
    o[d] = nx[d];                                                                                        \
    if (RELOAD) {                                                                                        \
      if constexpr (!TA1) { if (o[d].a != kPrev) load_tile<S, VW>(ta[d], vec, o[d].a, Wp, w0); }         \
      load_tile<S, VW>(tb[d], vec, o[d].b, Wp, w0);                                                      \
    }                                                                                                    \
  }
      int k = b;
      // steady state: every set is reloaded unconditionally, so the waits in front of a set's use can leave the younger
      // sets' requests in flight
      for (; k + 2 * D <= e; k += D) {
#pragma unroll
        for (int d = 0; d < D; d++) {
          MPF_CHAIN_STEP(d, true);
          nx[d] = ops[k + d + 2 * D < e ? k + d + 2 * D : e - 1];
        }
      }
      for (; k < e; k += D) {
#pragma unroll
        for (int d = 0; d < D; d++)
          if (k + d < e) MPF_CHAIN_STEP(d, k + d + D < e);
      }
#undef MPF_CHAIN_STEP
    }
   }
    __syncthreads();
  }
  if (!done) return;                              // large refresh: launch_cntsum folds the counts with the whole chip
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    const uint32_t ticket = __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = ticket == gridDim.x - 1;
  }
  __syncthreads();
  if (s_last) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    fold_counts(ops, n_ops, cntp, nslots, (int)gridDim.x, cnt, (int)threadIdx.x, (int)blockDim.x, x.cnt_host);
    if (threadIdx.x == 0) __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// Refresh schedule of a complete tree, made on the device.  The level of a directional vector is 1 + the larger level of its two
// inputs (tips: 0) -- a function of the topology array alone.  One workgroup: every thread owns the records tid, tid + 1024, ...
// (their kids in registers), levels settle by in-place relaxation in LDS (a record resolves in the round after its inputs, or
// in the same one if they were written earlier in it: the value is the same either way), one barrier per round; then a
// counting sort by level.  Rounds <= the tree's largest level (43 for the C3 start tree), ~0.2 us each.
static size_t sched_lds_bytes(uint32_t n, uint32_t ns, bool desc)
{
  const size_t a = (((size_t)ns * 2 + 15) & ~(size_t)15) + ((size_t)ns / 3 + 16) * 4;                      // levels (16 bit) + per-level counters
  const size_t b = (((size_t)ns * 2 + 15) & ~(size_t)15) + (((size_t)n * 2 + 15) & ~(size_t)15) + (size_t)ns * 4;   // N(., m) twice, tip links, kids
  return desc ? std::max(a, b) : a;
}

// second workgroup of k_sched: scan descriptors of a sweep (Engine::plan_walk on the device).  N(c, m) = insertion tests behind
// record c within m steps = 1 + N(kid1, m - 1) + N(kid2, m - 1), tips and m = 1: 1 -- at most 63 for radius 6, one byte each.
__device__ void sweep_desc_block(const uint2 *__restrict__ kids_g, uint32_t n, uint32_t n_ops, const SweepDescArgs &a, uint8_t *s_mem,
                                 uint32_t *s_w)
{
  const uint32_t tid = threadIdx.x, ns = n + n_ops;
  const size_t half = (((size_t)ns * 2 + 15) & ~(size_t)15) / 2;
  uint8_t *pa = s_mem, *pc = s_mem + half;
  uint16_t *s_tipback = reinterpret_cast<uint16_t *>(s_mem + 2 * half);
  uint32_t *s_k = reinterpret_cast<uint32_t *>(s_mem + 2 * half + (((size_t)n * 2 + 15) & ~(size_t)15));   // the topology array, 16 bits per kid
  for (uint32_t c = tid; c < ns; c += 1024u) {
    pa[c] = 1;
    pc[c] = 1;
    const uint2 k = c < n ? make_uint2(0u, 0u) : kids_g[c];
    s_k[c] = k.x | (k.y << 16);
  }
  __syncthreads();
  struct { const uint32_t *k; __device__ uint2 operator[](uint32_t c) const { const uint32_t v = k[c]; return make_uint2(v & 0xFFFFu, v >> 16); } } kids{s_k};
  // the record behind record c: back(c) = first kid of the record before c in its node's ring; tips: found by their neighbour
  auto backc = [&](uint32_t c) -> uint32_t {
    if (c < n) return s_tipback[c];
    const uint32_t k = c - n, b3 = n + 3u * (k / 3u), s = k % 3u;
    return kids[b3 + (s + 2u) % 3u].x;
  };
  for (uint32_t c = n + tid; c < ns; c += 1024u) {
    const uint32_t k = c - n, b3 = n + 3u * (k / 3u), s = k % 3u;
    const uint32_t bc = kids[b3 + (s + 2u) % 3u].x;
    if (bc < n) s_tipback[bc] = (uint16_t)c;
  }
  for (uint32_t m = 2; m <= a.maxtrav; m++) {
    for (uint32_t c = n + tid; c < ns; c += 1024u) { const uint2 k = kids[c]; pc[c] = (uint8_t)(1u + pa[k.x] + pa[k.y]); }
    __syncthreads();
    uint8_t *t = pa; pa = pc; pc = t;
  }
  __syncthreads();
  auto cv = [&](uint32_t c) -> uint32_t { return c < n ? 1u : (uint32_t)pa[c]; };
  // the parts of prune node i, in plan_walk's order: emit(s_cid, xa, xb, mintrav of the phase, side mask, child mask, candidates,
  // candidates behind the first gap end)
  auto node_parts = [&](uint32_t i, auto &&emit) {
    const uint32_t p = a.nodep[i], q = backc(p);
    auto phase = [&](uint32_t x, uint32_t s, uint32_t mt) {
      const uint2 xs = kids[x];
      const uint32_t xv[2] = {xs.x, xs.y}, skip = mt > 1u ? 1u : 0u;
      uint32_t cnt[2][2] = {{0u, 0u}, {0u, 0u}};
      for (int side = 0; side < 2; side++) {
        if (xv[side] < n) continue;
        const uint2 k2 = kids[xv[side]];
        cnt[side][0] = cv(k2.x) - skip;
        cnt[side][1] = cv(k2.y) - skip;
      }
      const uint32_t total = cnt[0][0] + cnt[0][1] + cnt[1][0] + cnt[1][1];
      if (total <= a.split_cands) { emit(s, xs.x, xs.y, mt, 3u, 3u, total, cnt[0][0] + cnt[0][1]); return; }
      for (int side = 0; side < 2; side++) {
        if (xv[side] < n) continue;
        emit(s, xs.x, xs.y, mt, 1u << side, 1u, cnt[side][0], 0u);
        emit(s, xs.x, xs.y, mt, 1u << side, 2u, cnt[side][1], 0u);
      }
    };
    if (p >= n) {
      const uint2 k = kids[p];
      if (k.x >= n || k.y >= n) phase(p, q, 1u);
    }
    if (q >= n) {
      const uint2 k = kids[q];
      bool ok = false;
      if (k.x >= n) { const uint2 k2 = kids[k.x]; ok = ok || k2.x >= n || k2.y >= n; }
      if (k.y >= n) { const uint2 k2 = kids[k.y]; ok = ok || k2.x >= n || k2.y >= n; }
      if (ok) phase(q, p, 2u);
    }
  };
  // consecutive prune nodes per thread; block-wide exclusive prefix of (parts, candidates)
  const uint32_t CH = (a.n_prune + 1023u) / 1024u;
  const uint32_t i0 = tid * CH, i1 = min(i0 + CH, a.n_prune);
  uint32_t np = 0, nc = 0;
  for (uint32_t i = i0; i < i1; i++) node_parts(i, [&](uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t c, uint32_t) { np++; nc += c; });
  uint32_t ip = np, ic = nc;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t op = (uint32_t)__shfl_up((int)ip, d, 64), oc = (uint32_t)__shfl_up((int)ic, d, 64);
    if ((int)(tid & 63u) >= d) { ip += op; ic += oc; }
  }
  if ((tid & 63u) == 63u) { s_w[tid >> 6] = ip; s_w[16 + (tid >> 6)] = ic; }
  __syncthreads();
  uint32_t bp = ip - np, bc = ic - nc, tp = 0, tc = 0;
  for (uint32_t w = 0; w < 16u; w++) {
    if (w < (tid >> 6)) { bp += s_w[w]; bc += s_w[16 + w]; }
    tp += s_w[w];
    tc += s_w[16 + w];
  }
  for (uint32_t i = i0; i < i1; i++)
    node_parts(i, [&](uint32_t s, uint32_t xa, uint32_t xb, uint32_t mt, uint32_t sm, uint32_t cm, uint32_t c, uint32_t first) {
      a.desc[bp] = WalkDesc{s, xa, xb, mt | (a.maxtrav << 8) | (sm << 16) | (cm << 18), bc, c, first, 0u};
      a.parts[bp] = make_uint2(bc, c);
      a.part_node[bp] = i;
      bp++;
      bc += c;
    });
  // the host polls the flag: everything this workgroup wrote to its memory lies in front of it
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
  __syncthreads();
  if (tid == 0u) {
    if (a.hdr_dev) { a.hdr_dev[0] = tp; a.hdr_dev[1] = tc; }
    a.hdr_host[0] = tp;
    a.hdr_host[1] = tc;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    __hip_atomic_store(a.hdr_host + 3, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// Every thread owns R records (cid n + tid + 1024 j) with their inputs in registers.  Relaxation is monotone -- a record that
// finds both inputs settled takes its final level whenever it looks -- so the waves need no barrier between passes, only the
// test whether anything is left does: four passes per test.  A pass costs one LDS round trip: all reads first (a settled
// record reads word 0, a broadcast), skipped per record slot when the whole wave is through with it.  39 levels at C3: 7 us
// (21 us when every thread re-read all its records every pass, 22 us with a compacted work list in LDS: five dependent LDS
// round trips per pass at ~200 cycles each).
template <int R>
__global__ __launch_bounds__(1024) void k_sched(const uint2 *__restrict__ kids, uint32_t n, uint32_t n_ops, NvOp *__restrict__ ops,
                                                int32_t *__restrict__ lev_off, int32_t *__restrict__ n_lev_out, SweepDescArgs sw,
                                                uint2 *__restrict__ kids_copy)
{
  extern __shared__ __attribute__((aligned(16))) uint8_t s_mem[];
  __shared__ uint32_t s_flag[3], s_max, s_wsum[32];
  const unsigned long long t_begin = __builtin_amdgcn_s_memrealtime();      // (100 MHz; the durations go out as diagnostics)
  if (blockIdx.x == 1u) {
    sweep_desc_block(kids, n, n_ops, sw, s_mem, s_wsum);
    if (threadIdx.x == 0u) n_lev_out[2] = (int32_t)(__builtin_amdgcn_s_memrealtime() - t_begin);
    return;
  }
  const uint32_t tid = threadIdx.x, lane = tid & 63u, ns = n + n_ops;
  uint16_t *s_lev = reinterpret_cast<uint16_t *>(s_mem);
  uint32_t *s_cnt = reinterpret_cast<uint32_t *>(s_mem + (((size_t)ns * 2 + 15) & ~(size_t)15));
  const uint32_t n_cnt = ns / 3u + 16u;
  for (uint32_t c = tid; c < n; c += 1024u) s_lev[c] = 0;
  for (uint32_t l = tid; l < n_cnt; l += 1024u) s_cnt[l] = 0u;
  if (tid < 3u) s_flag[tid] = 0u;
  if (tid == 0u) s_max = 0u;
  uint2 kk[R];
  uint32_t mylev[R], la[R], lb[R];
#pragma unroll
  for (int j = 0; j < R; j++) {
    const uint32_t c = n + tid + 1024u * (uint32_t)j;
    kk[j] = c < ns ? kids[c] : make_uint2(0u, 0u);
    if (kids_copy && c < ns) kids_copy[c] = kk[j];         // (kids came from pinned host memory: the device copy for the kernels behind)
  }
#pragma unroll
  for (int j = 0; j < R; j++) {
    const uint32_t c = n + tid + 1024u * (uint32_t)j;
    // records between two tips are level 1 at once
    mylev[j] = c >= ns ? 0u : (kk[j].x < n && kk[j].y < n) ? 1u : 0xFFFFu;
    if (c < ns) s_lev[c] = (uint16_t)mylev[j];
    la[j] = kk[j].x < n ? 0u : 0xFFFFu;          // (tips are level 0)
    lb[j] = kk[j].y < n ? 0u : 0xFFFFu;
  }
  __syncthreads();
  for (uint32_t round = 0; round <= ns; round++) { // (round > ns: a topology array that is not a tree -- leave, those ops stay unwritten)
    uint32_t left = 0;
    for (int pass = 0; pass < 4; pass++) {
      // an input seen settled is not read again (most records wait for ONE deep input): la / lb keep what was read
#pragma unroll
      for (int j = 0; j < R; j++) {
        const bool wa = mylev[j] == 0xFFFFu && la[j] == 0xFFFFu, wb = mylev[j] == 0xFFFFu && lb[j] == 0xFFFFu;
        if (__ballot(wa) != 0ull) { const uint32_t v = s_lev[wa ? kk[j].x : 0u]; la[j] = wa ? v : la[j]; }     // (wave-uniform branches;
        if (__ballot(wb) != 0ull) { const uint32_t v = s_lev[wb ? kk[j].y : 0u]; lb[j] = wb ? v : lb[j]; }     //  settled lanes read word 0: a broadcast)
      }
      left = 0;
#pragma unroll
      for (int j = 0; j < R; j++) {
        const bool open = mylev[j] == 0xFFFFu, ready = la[j] != 0xFFFFu && lb[j] != 0xFFFFu;
        if (open && ready) {
          mylev[j] = 1u + (la[j] > lb[j] ? la[j] : lb[j]);
          s_lev[n + tid + 1024u * (uint32_t)j] = (uint16_t)mylev[j];
        }
        left |= (open && !ready) ? 1u : 0u;
      }
      if (!__ballot(left != 0u)) break;             // this wave is through
    }
    // flags in rotation: round r raises flag r % 3; the one cleared after this barrier was read before it and is raised again
    // only after the next
    const uint32_t f = round % 3u;
    if (left) s_flag[f] = 1u;
    __syncthreads();
    const uint32_t any = s_flag[f];
    if (tid == 0u) s_flag[(f + 2u) % 3u] = 0u;
    if (!any) break;
  }
  // histogram over levels, largest level
  uint32_t mx = 0;
#pragma unroll
  for (int j = 0; j < R; j++) {
    const uint32_t c = n + tid + 1024u * (uint32_t)j;
    if (c < ns && mylev[j] != 0xFFFFu) {
      atomicAdd(&s_cnt[mylev[j]], 1u);
      mx = mylev[j] > mx ? mylev[j] : mx;
    }
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)mx, m, 64); mx = o > mx ? o : mx; }
  if (lane == 0u) atomicMax(&s_max, mx);
  __syncthreads();
  const uint32_t maxlev = s_max;
  // exclusive prefix over s_cnt[1 .. maxlev]: CH consecutive levels per thread, wave scan, wave sums through LDS
  const uint32_t CH = (maxlev + 1023u) / 1024u;
  const uint32_t l0 = 1u + tid * CH;
  uint32_t loc = 0;
  for (uint32_t l = l0; l < l0 + CH && l <= maxlev; l++) loc += s_cnt[l];
  uint32_t inc = loc;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)inc, d, 64); if ((int)lane >= d) inc += o; }
  if (lane == 63u) s_wsum[tid >> 6] = inc;
  __syncthreads();
  uint32_t base = inc - loc;
  for (uint32_t w = 0; w < (tid >> 6); w++) base += s_wsum[w];
  for (uint32_t l = l0; l < l0 + CH && l <= maxlev; l++) { const uint32_t c = s_cnt[l]; s_cnt[l] = base; base += c; }
  __syncthreads();
  // placement: s_cnt[l] runs from the start of level l to its end (= the start of level l + 1)
#pragma unroll
  for (int j = 0; j < R; j++) {
    const uint32_t c = n + tid + 1024u * (uint32_t)j;
    if (c < ns && mylev[j] != 0xFFFFu) {
      const uint32_t pos = atomicAdd(&s_cnt[mylev[j]], 1u);
      const uint32_t k = c - n, rec = 3u * (n + 1u + k / 3u) + k % 3u;
      ops[pos] = NvOp{c, kk[j].x, kk[j].y, rec};
    }
  }
  __syncthreads();
  for (uint32_t l = tid; l <= maxlev; l += 1024u) lev_off[l] = l ? (int32_t)s_cnt[l] : 0;
  if (tid == 0u) { n_lev_out[0] = (int32_t)maxlev; n_lev_out[1] = (int32_t)(__builtin_amdgcn_s_memrealtime() - t_begin); }
}

hipError_t launch_sched(hipStream_t st, const uint2 *kids, uint32_t n_taxa, uint32_t n_ops, NvOp *ops, int32_t *lev_off, int32_t *n_lev,
                        const SweepDescArgs &sw, uint2 *kids_copy)
{
  if (n_taxa + n_ops > kSchedMaxSlots) return hipErrorInvalidValue;
  if (sw.nodep && (sw.maxtrav < 1u || sw.maxtrav > 6u)) return hipErrorInvalidValue;
  const size_t lds = sched_lds_bytes(n_taxa, n_taxa + n_ops, sw.nodep != nullptr);
  const uint32_t per = (n_ops + 1023u) / 1024u;
#define MPF_SCHED(R_)                                                                                                                   \
  do {                                                                                                                                  \
    if (lds > 60 * 1024) {                                                                                                              \
      static thread_local int attr_dev = -1;                                                                                            \
      int dev = 0;                                                                                                                      \
      (void)hipGetDevice(&dev);                                                                                                         \
      if (attr_dev != dev) {                                                                                                            \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sched<R_>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        if (e != hipSuccess) return e;                                                                                                  \
        attr_dev = dev;                                                                                                                 \
      }                                                                                                                                 \
    }                                                                                                                                   \
    hipLaunchKernelGGL((k_sched<R_>), dim3(sw.nodep ? 2 : 1), dim3(1024), lds, st, kids, n_taxa, n_ops, ops, lev_off, n_lev, sw, kids_copy);       \
  } while (0)
  if (per <= 1u) MPF_SCHED(1);
  else if (per <= 2u) MPF_SCHED(2);
  else if (per <= 3u) MPF_SCHED(3);
  else if (per <= 4u) MPF_SCHED(4);
  else if (per <= 6u) MPF_SCHED(6);
  else if (per <= 8u) MPF_SCHED(8);
  else MPF_SCHED(16);
#undef MPF_SCHED
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void k_cntsum(const NvOp *__restrict__ ops, int n_ops, const uint32_t *__restrict__ cntp, uint32_t nslots,
                                                int tiles, uint32_t *__restrict__ cnt, uint32_t *__restrict__ cnt_host)
{
  const int l32 = threadIdx.x & 31;
  const int i = (int)(blockIdx.x * (blockDim.x >> 5) + (threadIdx.x >> 5));      // 32 lanes per op
  if (i >= n_ops) return;
  const uint32_t dst = ops[i].dst;
  uint32_t s = 0;
  for (int t = l32; t < tiles; t += 32) s += cntp[(size_t)t * nslots + dst];
#pragma unroll
  for (int m = 16; m >= 1; m >>= 1) s += (uint32_t)__shfl_xor((int)s, m, 32);
  if (l32 == 0) {
    cnt[dst] = s;
    if (cnt_host) cnt_host[dst] = s;
  }
}

// ---------------------------------------------------------------- K2: batched evaluate

template <int S, int VW, int RED>
__global__ __launch_bounds__(256) void k_evaluate(const uint32_t *__restrict__ vec, const EvOp *__restrict__ ops,
                                                  int n_ops, uint32_t *__restrict__ out, int Wp, int tiles)
{
  const int lane = threadIdx.x & 63;
  int gw = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  gw = __builtin_amdgcn_readfirstlane(gw);
  if (gw >= n_ops * tiles) return;
  const int op = gw / tiles, tile = gw - op * tiles;
  const EvOp o = ops[op];
  bool valid;
  const int w0 = lane_word<VW>(tile, lane, Wp, valid);
  Tile<S, VW> a, b;
  load_tile<S, VW>(a, vec, o.a, Wp, w0);
  load_tile<S, VW>(b, vec, o.b, Wp, w0);
  uint32_t cost = empty_count<S, VW>(a, b);
  cost = valid ? cost : 0u;
  const uint32_t tot = wave_total<RED>(cost);
  if (lane == 0 && tot) atomic_add_u32(out + o.out, tot);
}

// ---------------------------------------------------------------- K4: per-pattern scores
//
// Per-site Fitch length = number of (a, b) joins of the rooted traversal whose state sets do not
// intersect at that site (reference storePerSiteNodeScores / addPerSiteSubtreeScores,
// sprparsimony.cpp:294-376, which keeps a 32-bit counter per site per node: 401 MB at 1000 x 50k).
// Here a wave takes a chunk of <= 63 joins for its tile and adds their mutation masks into SIX
// bit-sliced counter planes (carry-save ripple), i.e. 32 sites are counted per instruction;
// k_pattern_sum then reads each pattern's first site out of the planes.
constexpr int kPlaneChunk = 63;
constexpr int kPlanes = 6;

template <int S, int VW>
__global__ __launch_bounds__(256) void k_site_planes(const uint32_t *__restrict__ vec, const EvOp *__restrict__ ops,
                                                     int n_ops, uint32_t *__restrict__ planes, int Wp, int tiles)
{
  const int lane = threadIdx.x & 63;
  int gw = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  gw = __builtin_amdgcn_readfirstlane(gw);
  const int n_chunks = (n_ops + kPlaneChunk - 1) / kPlaneChunk;
  if (gw >= n_chunks * tiles) return;
  const int chunk = gw / tiles, tile = gw - chunk * tiles;
  bool valid;
  const int w0 = lane_word<VW>(tile, lane, Wp, valid);
  uint32_t c[kPlanes][VW];
#pragma unroll
  for (int j = 0; j < kPlanes; j++)
#pragma unroll
    for (int v = 0; v < VW; v++) c[j][v] = 0;
  const int b = chunk * kPlaneChunk, e = min(n_ops, b + kPlaneChunk);
  for (int i = b; i < e; i++) {
    const EvOp o = ops[i];
    Tile<S, VW> x, y;
    load_tile<S, VW>(x, vec, o.a, Wp, w0);
    load_tile<S, VW>(y, vec, o.b, Wp, w0);
#pragma unroll
    for (int v = 0; v < VW; v++) {
      uint32_t any = 0;
#pragma unroll
      for (int k = 0; k < S; k++) any |= x.v[k][v] & y.v[k][v];
      uint32_t carry = ~any;
#pragma unroll
      for (int j = 0; j < kPlanes; j++) {
        const uint32_t t = c[j][v] & carry;
        c[j][v] ^= carry;
        carry = t;
      }
    }
  }
  if (valid) {
#pragma unroll
    for (int j = 0; j < kPlanes; j++)
#pragma unroll
      for (int v = 0; v < VW; v++) planes[((size_t)chunk * kPlanes + j) * Wp + w0 + v] = c[j][v];
  }
}

__global__ void k_pattern_sum(const uint32_t *__restrict__ planes, int n_chunks, int Wp,
                              const int32_t *__restrict__ first_site, int n_patterns, uint16_t *__restrict__ ptn)
{
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_patterns) return;
  const int site = first_site[p];
  if (site < 0) { ptn[p] = 0; return; }
  const int w = site >> 5, bit = site & 31;
  uint32_t total = 0;
  for (int ch = 0; ch < n_chunks; ch++)
#pragma unroll
    for (int j = 0; j < kPlanes; j++) total += ((planes[((size_t)ch * kPlanes + j) * Wp + w] >> bit) & 1u) << j;
  ptn[p] = (uint16_t)total;
}

// ---------------------------------------------------------------- SPR scan (K1+K2 fused over a DFS program)
//
// One wavefront = one (scan, tile).  A scan is the radius-limited neighbourhood of one prune
// record (reference rearrangeParsimony, sprparsimony.cpp:2259-2376).  With the pruned
// subtree's vector s fixed in registers, the candidate on branch (own, parent) costs
//     popcount(~OR_k( fitch(U_d, vec[own])_k & s_k )),    U_d = fitch(U_{d-1}, vec[sibling])
// where U_d ("up" vector of the remaining tree at depth d of the DFS) lives in registers:
// U is indexed by the wave-uniform depth through a switch so that every access is a
// compile-time register name.  Per candidate the wave reads two vectors and writes none.

template <int S, int VW, int MAXD, int RED>
__device__ __forceinline__ void scan_body(const uint32_t *__restrict__ vec, const ScanHdr *__restrict__ hdr,
                                          int n_scans, const ScanOp *__restrict__ ops, uint32_t *__restrict__ out,
                                          int Wp, int tiles, int map)
{
  const int lane = threadIdx.x & 63;
  int gw = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  gw = __builtin_amdgcn_readfirstlane(gw);
  int scan, tile;
  if (map == 0) {
    if (gw >= n_scans * tiles) return;
    scan = gw / tiles;
    tile = gw - scan * tiles;
  } else {
    // tiles pinned to XCD classes: workgroups are dealt round-robin over the 8 XCDs, so all
    // workgroups with equal blockIdx%8 share one L2; class c owns tiles {c, c+8, ...} and walks
    // the scans in order, which keeps a tile's slice of the directional vectors L2-resident
    const int wpb = blockDim.x >> 6;
    const int cls = blockIdx.x & 7;
    const int idx = (blockIdx.x >> 3) * wpb + (threadIdx.x >> 6);
    const int ntc = (tiles - cls + 7) >> 3;            // tiles in this class
    if (ntc <= 0 || idx >= n_scans * ntc) return;
    scan = idx / ntc;
    tile = cls + 8 * (idx - scan * ntc);
  }
  const ScanHdr h = hdr[scan];
  bool valid;
  const int w0 = lane_word<VW>(tile, lane, Wp, valid);

  Tile<S, VW> sv, U[MAXD + 1], dsib, down;
  load_tile<S, VW>(sv, vec, h.s_slot, Wp, w0);

  for (uint32_t i = h.op_begin; i < h.op_end; i++) {
    const ScanOp o = ops[i];
    const int d = (int)(o.meta & 0xFFu);
    const bool test = (o.meta >> 8) & 1u;
    const int kind = (int)((o.meta >> 16) & 0xFFu);
    if (kind == SCAN_ROOT) {
      load_tile<S, VW>(U[0], vec, o.own, Wp, w0);
      continue;
    }
    load_tile<S, VW>(dsib, vec, o.sib, Wp, w0);
    uint32_t cost = 0;
    if (kind == SCAN_JOIN) {
      load_tile<S, VW>(down, vec, o.own, Wp, w0);
      cost = join_cost<S, VW>(dsib, down, sv);
    } else {
      if (test) load_tile<S, VW>(down, vec, o.own, Wp, w0);
#define MPF_LEVEL(c)                                           \
  case c:                                                      \
    if constexpr (c <= MAXD) {                                 \
      fitch<S, VW>(U[c], U[c - 1], dsib);                      \
      if (test) cost = join_cost<S, VW>(U[c], down, sv);       \
    }                                                          \
    break;
      switch (d) {
        MPF_LEVEL(1) MPF_LEVEL(2) MPF_LEVEL(3) MPF_LEVEL(4) MPF_LEVEL(5) MPF_LEVEL(6)
        MPF_LEVEL(7) MPF_LEVEL(8) MPF_LEVEL(9) MPF_LEVEL(10) MPF_LEVEL(11) MPF_LEVEL(12)
        default: break;
      }
#undef MPF_LEVEL
    }
    if (test || kind == SCAN_JOIN) {
      cost = valid ? cost : 0u;
      const uint32_t tot = wave_total<RED>(cost);
      if (lane == 0 && tot) atomic_add_u32(out + o.out, tot);
    }
  }
}

// The same programs at ANY radius (rearrangeParsimony takes whatever -spr_rad gives it, sprparsimony.cpp:2259-2376): the up-vectors
// of the levels do not fit registers beyond MAXD = 12, so here they live in a scratch area of the wave in HBM (one tile per
// level).  The vector of the level just computed stays in registers -- a DFS mostly goes on one level down --, a step back up
// re-reads its parent level.  Rarely used (mpboot's default radius is 6): one store per step, sometimes a load; the launches are
// cut so that the scratch stays bounded (launch_scan).
template <int S, int VW, int RED>
__global__ __launch_bounds__(256) void k_scan_deep(const uint32_t *__restrict__ vec, const ScanHdr *__restrict__ hdr,
                                                   int n_scans, const ScanOp *__restrict__ ops, uint32_t *__restrict__ out,
                                                   int Wp, int tiles, uint32_t *__restrict__ scratch, int levels)
{
  const int lane = threadIdx.x & 63;
  int gw = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  gw = __builtin_amdgcn_readfirstlane(gw);
  if (gw >= n_scans * tiles) return;
  const int scan = gw / tiles, tile = gw - scan * tiles;
  const ScanHdr h = hdr[scan];
  bool valid;
  const int w0 = lane_word<VW>(tile, lane, Wp, valid);
  uint32_t *mine = scratch + (size_t)gw * (size_t)levels * (size_t)(S * VW * 64) + lane;
  auto put = [&](int lev, const Tile<S, VW> &t) {
    uint32_t *q = mine + (size_t)lev * (size_t)(S * VW * 64);
#pragma unroll
    for (int k = 0; k < S * VW; k++) q[k * 64] = t.v[k / VW][k % VW];
  };
  auto get = [&](int lev, Tile<S, VW> &t) {
    const uint32_t *q = mine + (size_t)lev * (size_t)(S * VW * 64);
#pragma unroll
    for (int k = 0; k < S * VW; k++) t.v[k / VW][k % VW] = q[k * 64];
  };
  Tile<S, VW> sv, last, par, dsib, down;
  int ld = -1;                                       // level of `last`
  load_tile<S, VW>(sv, vec, h.s_slot, Wp, w0);
  for (uint32_t i = h.op_begin; i < h.op_end; i++) {
    const ScanOp o = ops[i];
    const int d = (int)(o.meta & 0xFFu);
    const bool test = (o.meta >> 8) & 1u;
    const int kind = (int)((o.meta >> 16) & 0xFFu);
    if (kind == SCAN_ROOT) {
      load_tile<S, VW>(last, vec, o.own, Wp, w0);
      put(0, last);
      ld = 0;
      continue;
    }
    load_tile<S, VW>(dsib, vec, o.sib, Wp, w0);
    uint32_t cost = 0;
    if (kind == SCAN_JOIN) {
      load_tile<S, VW>(down, vec, o.own, Wp, w0);
      cost = join_cost<S, VW>(dsib, down, sv);
    } else {
      if (test) load_tile<S, VW>(down, vec, o.own, Wp, w0);
      if (d - 1 == ld) par = last; else get(d - 1, par);
      fitch<S, VW>(last, par, dsib);
      ld = d;
      if (d + 1 < levels) put(d, last);              // (the deepest level has no children)
      if (test) cost = join_cost<S, VW>(last, down, sv);
    }
    if (test || kind == SCAN_JOIN) {
      cost = valid ? cost : 0u;
      const uint32_t tot = wave_total<RED>(cost);
      if (lane == 0 && tot) atomic_add_u32(out + o.out, tot);
    }
  }
}

// Tail of a launch whose last workgroup has just copied its n_out results into the host's pinned buffer: the completion
// counter goes back to zero and host_out[n_out] = 1 tells a polling host thread that results (and the mutation counts an
// earlier launch wrote to the host) are there -- without the wake-up latency of a stream synchronisation.
__device__ __forceinline__ void host_results_ready(uint32_t *__restrict__ host_out, uint32_t n_out, uint32_t *__restrict__ done)
{
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");   // every wave: its own words have arrived before anyone raises the flag
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    __hip_atomic_store(host_out + n_out, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// host_out != nullptr (stepwise addition): the last workgroup copies the n_out costs to the host's pinned buffer, as in
// k_scan_walk
template <int S, int VW, int MAXD, int RED>
__global__ __launch_bounds__(256) void k_scan(const uint32_t *__restrict__ vec, const ScanHdr *__restrict__ hdr,
                                              int n_scans, const ScanOp *__restrict__ ops, uint32_t *__restrict__ out,
                                              int Wp, int tiles, int map, uint32_t *__restrict__ host_out, uint32_t n_out,
                                              uint32_t *__restrict__ done)
{
  scan_body<S, VW, MAXD, RED>(vec, hdr, n_scans, ops, out, Wp, tiles, map);
  if (!host_out) return;
  __shared__ int s_last;
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    const uint32_t ticket = __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = ticket == gridDim.x - 1;
  }
  __syncthreads();
  if (s_last) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    for (uint32_t i = threadIdx.x; i < n_out; i += blockDim.x) host_out[i] = __builtin_nontemporal_load(out + i);
    host_results_ready(host_out, n_out, done);
  }
}


// ---------------------------------------------------------------- SPR scan, device-walked
//
// Same arithmetic as k_scan, but the wave enumerates the neighbourhood itself: the DFS of
// addTraverseParsimony (reference sprparsimony.cpp:2208-2218) runs on the scalar unit over the
// `back` links, its frame stack lives in LDS (one small stack per wave), and the two children of
// a node are expanded TOGETHER: their vectors d1, d2 are loaded once and give both
//     U(c1) = fitch(U(parent), d2),  U(c2) = fitch(U(parent), d1)
// and both candidates' costs, i.e. one vector read per candidate.  U lives in registers, two slots per
// depth, selected by a wave-uniform switch.  Candidate costs are emitted in the reference's order
// (the second child's cost waits in its stack frame until its turn).

// uniform-base tile load: `base` = vec + cid * S * Wp is wave-uniform (SGPR pair), the lane supplies a
// 32-bit element offset -> global_load with scalar base + vector offset, no 64-bit vector address math
template <int S, int VW>
__device__ __forceinline__ void load_tile_u(Tile<S, VW> &t, const uint32_t *__restrict__ base, uint32_t w0, uint32_t Wp)
{
#pragma unroll
  for (int k = 0; k < S; k++) {
    const uint32_t off = w0 + (uint32_t)k * Wp;
    if constexpr (VW == 1) {
      t.v[k][0] = base[off];
    } else if constexpr (VW == 2) {
      uint2 x = *reinterpret_cast<const uint2 *>(base + off);
      t.v[k][0] = x.x; t.v[k][1] = x.y;
    } else {
      uint4 x = *reinterpret_cast<const uint4 *>(base + off);
      t.v[k][0] = x.x; t.v[k][1] = x.y; t.v[k][2] = x.z; t.v[k][3] = x.w;
    }
  }
}

// buffer-addressed tile load: scalar byte offset of the vector + per-lane byte offsets of the rows held in
// registers -> one buffer_load per row with no per-load address arithmetic
template <int S, int VW>
__device__ __forceinline__ void load_tile_b(Tile<S, VW> &t, __amdgpu_buffer_rsrc_t rsrc, const uint32_t (&voff)[S],
                                            uint32_t soff)
{
#pragma unroll
  for (int k = 0; k < S; k++) {
    if constexpr (VW == 1) {
      t.v[k][0] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff[k], soff, 0);
    } else if constexpr (VW == 2) {
      auto x = __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff[k], soff, 0);
      t.v[k][0] = x[0]; t.v[k][1] = x[1];
    } else {
      auto x = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[k], soff, 0);
      t.v[k][0] = x[0]; t.v[k][1] = x[1]; t.v[k][2] = x[2]; t.v[k][3] = x[3];
    }
  }
}

// the same through a 64-bit base: for vector stores of 2 GiB and more (a raw buffer addresses 32 bits)
template <int S, int VW>
__device__ __forceinline__ void load_tile_g(Tile<S, VW> &t, const uint32_t *__restrict__ base, const uint32_t (&voff)[S])
{
  const char *b = reinterpret_cast<const char *>(base);
#pragma unroll
  for (int k = 0; k < S; k++) {
    if constexpr (VW == 1) {
      t.v[k][0] = *reinterpret_cast<const uint32_t *>(b + voff[k]);
    } else if constexpr (VW == 2) {
      const uint2 x = *reinterpret_cast<const uint2 *>(b + voff[k]);
      t.v[k][0] = x.x; t.v[k][1] = x.y;
    } else {
      const uint4 x = *reinterpret_cast<const uint4 *>(b + voff[k]);
      t.v[k][0] = x.x; t.v[k][1] = x.y; t.v[k][2] = x.z; t.v[k][3] = x.w;
    }
  }
}

// S = states held per lane.  SPLIT (protein): lanes l and l^32 share a word and hold states 0..9 / 10..19, a
// wave covers 32 words; this keeps the protein kernel at DNA-like register counts instead of one wave per SIMD.
// MASKS (online UFBoot, ufboot.hip): every candidate's "no common state" words go to masks[out_base + m][Wp], m
// counting candidates in the order they are COMPUTED; info[out_base + k] = (out_base + m, scan) names the row of the
// k-th EMITTED candidate.  The part's last slot (index out_base + count) receives the join of the pruned subtree
// onto its home edge, fitch(vec[xa], vec[xb]).
// BIG: the vector store does not fit a raw buffer's 32-bit range: plain global loads from a 64-bit base per vector.
// DEEP (any radius above kWalkMaxDepth, MAXD = 255): the parked up-vectors live in a scratch area of the wave in HBM (pend_g:
// [wave of the launch][level][state, word][lane]) instead of the LDS; the launches are cut so that it stays bounded
// (launch_scan_walk), scan_base = index of the launch's first scan among the batch's (what info[] names).
template <int S, int VW, int MAXD, int RED, bool SPLIT, bool MASKS, bool BIG, bool WM = false, bool DEEP = false>
__device__ __forceinline__ void scan_walk_body(const uint32_t *__restrict__ vec, const uint2 *__restrict__ kids,
                                               uint32_t n, const WalkDesc *__restrict__ desc, int n_scans,
                                               uint32_t *__restrict__ out, uint32_t *__restrict__ ncand, int Wp,
                                               int tiles, int map, uint32_t *__restrict__ masks, uint2 *__restrict__ info,
                                               uint32_t *__restrict__ pend_g = nullptr, int levels = 0, uint32_t scan_base = 0)
{
  constexpr int STK = MAXD + 2;
  constexpr bool LANEACC = MAXD <= 6;      // short walks: candidate costs gathered in a lane register, 64 per flush
  __shared__ uint2 s_frame[4][STK];        // x = cid | depth<<24, y = cost
  // U of the not-yet-expanded second child, one slot per depth: [wave][depth][state, word][lane] (a lane reads back
  // what it wrote: no synchronisation, no bank conflicts).  Indexed by the wave-uniform depth -- in registers the same
  // indexing costs a cascade of compares and register copies per expansion.
  // (depth 1 stays in registers: it is touched once per side, and without its slot 8 workgroups fit a CU)
  // (protein tiles -- 10 states per wave half -- keep depth 2 in registers as well: with one LDS slot fewer a fourth
  //  workgroup fits a CU)
  constexpr int REGP = (S * VW >= 10) ? 2 : 1;
  __shared__ uint32_t s_pend[4][DEEP ? 1 : MAXD - 1 - REGP][DEEP ? 1 : S * VW][DEEP ? 1 : 64];
  uint32_t *my_pend = nullptr;                 // DEEP: this wave's levels in HBM

  const int lane = threadIdx.x & 63;
  const int wib = threadIdx.x >> 6;
  int scan, tile;
  if (map == 0 || DEEP) {
    int gw = blockIdx.x * (blockDim.x >> 6) + wib;
    gw = __builtin_amdgcn_readfirstlane(gw);
    if (gw >= n_scans * tiles) return;
    scan = gw / tiles;
    tile = gw - scan * tiles;
    if constexpr (DEEP) my_pend = pend_g + (size_t)gw * (size_t)levels * (size_t)(S * VW * 64) + lane;
  } else {
    // XCD-aware: workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 labels the groups that
    // share an L2).  The (tile, scan) items, tile-major, are cut into 8 equal contiguous chunks, one per
    // XCD class, and each class walks its chunk in order.  Affects speed only.
    const long total = (long)n_scans * tiles;
    const long chunk = (total + 7) / 8;
    const int cls = blockIdx.x & 7;
    long idx = (long)(blockIdx.x >> 3) * (blockDim.x >> 6) + wib;
    idx = __builtin_amdgcn_readfirstlane((int)idx);
    if (idx >= chunk) return;
    const long item = (long)cls * chunk + idx;
    if (item >= total) return;
    tile = (int)(item / n_scans);
    scan = (int)(item - (long)tile * n_scans);
  }
  const WalkDesc de = desc[scan];
  const uint32_t mintrav = de.trav & 0xFFu, maxtrav = (de.trav >> 8) & 0xFFu;
  // a scan may be cut into parts for latency (small batches): side_mask selects the gap ends to walk,
  // child_mask the first-level children whose candidate and subtree this part owns
  const uint32_t side_mask = (de.trav >> 16) & 3u, child_mask = (de.trav >> 18) & 3u;
  bool valid;
  uint32_t w0, row0 = 0;
  if constexpr (SPLIT) {
    const int w = tile * 32 + (lane & 31);
    valid = w < Wp && lane < 32;                         // the low half counts a word's mutations once
    w0 = (uint32_t)(w < Wp ? w : Wp - 1);
    row0 = (uint32_t)(lane >> 5) * (uint32_t)S;
  } else {
    w0 = (uint32_t)lane_word<VW>(tile, lane, Wp, valid);
  }
  constexpr uint32_t ST = SPLIT ? 2u * (uint32_t)S : (uint32_t)S;    // states per vector
  const uint32_t SW = ST * (uint32_t)Wp;
  // the whole vector array as one raw buffer (< 4 GiB, checked by the host): loads take a scalar byte
  // offset (vector) plus a per-lane byte offset (row, word)
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)vec, 0, 0x7FFFFFFF, 0x00020000);
  uint32_t voff[S];
#pragma unroll
  for (int k = 0; k < S; k++) voff[k] = (w0 + (row0 + (uint32_t)k) * (uint32_t)Wp) * 4u;
  // WM: `vec` is the word-major copy (Geometry::shoff): the four state words of a site word in one 16-byte load
  static_assert(!WM || (S == 4 && VW == 1 && !SPLIT && !BIG), "word-major copy: DNA, one word per lane, below 2 GiB");
#define MPF_LOAD(T, cid)                                                                       \
  do {                                                                                         \
    if constexpr (WM) {                                                                        \
      const auto x4 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, w0 * 16u, (uint32_t)(cid) * SW * 4u, 0); \
      T.v[0][0] = x4[0]; T.v[1][0] = x4[1]; T.v[2][0] = x4[2]; T.v[3][0] = x4[3];              \
    } else if constexpr (BIG) load_tile_g<S, VW>(T, vec + (size_t)(cid) * (size_t)SW, voff);   \
    else load_tile_b<S, VW>(T, rsrc, voff, (uint32_t)(cid) * SW * 4u);                          \
  } while (0)

  // sv: pruned subtree; par: U of the node being expanded; pend[d]: U of the not-yet-expanded second
  // child at depth d (one per depth suffices: the first child is expanded immediately)
  Tile<S, VW> sv, par, u1, u2, d1, d2, pend1, pend2;
  MPF_LOAD(sv, de.s_cid);
  uint32_t k = 0;                              // candidates emitted so far (scan-local index)
  uint32_t acc0 = 0;
  uint2 *stk = s_frame[wib];

  uint32_t mrow = 0;                           // MASKS: candidates computed so far
  auto put_mask = [&](const uint32_t (&m)[VW], uint32_t row) {
    if constexpr (MASKS) {
      const bool wr = SPLIT ? (valid && lane < 32) : valid;
      if (wr) {
        uint32_t *p = masks + (size_t)(de.out_base + row) * (size_t)Wp + w0;
#pragma unroll
        for (int j = 0; j < VW; j++) p[j] = m[j];
      }
    }
  };
  if constexpr (MASKS) {
    // home edge: the pruned subtree joined onto fitch(vec[xa], vec[xb])
    Tile<S, VW> ha, hb;
    uint32_t m0[VW];
    MPF_LOAD(ha, de.xa_cid);
    MPF_LOAD(hb, de.xb_cid);
    uint32_t c0 = join_mask<S, VW, SPLIT>(ha, hb, sv, m0);
    put_mask(m0, de.pad0);
    c0 = valid ? c0 : 0u;
    if constexpr (SPLIT) c0 = lane < 32 ? c0 : 0u;
    const uint32_t t0 = wave_total<RED>(c0);
    if (lane == 0 && t0) atomic_add_u32(out + de.out_base + de.pad0, t0);
    if (tile == 0 && lane == 0) info[de.out_base + de.pad0] = make_uint2(de.out_base + de.pad0, 0xFFFFFFFFu);
  }
  auto emit = [&](uint32_t c, uint32_t row) {
    if constexpr (MASKS) {
      if (tile == 0 && lane == 0) info[de.out_base + k] = make_uint2(de.out_base + row, scan_base + (uint32_t)scan);
    }
    if constexpr (LANEACC) {
      // candidate k is kept by lane k & 63 (a lane select instead of a memory atomic per candidate); every 64
      // candidates the wave adds its partial costs to the output with one atomic per lane
      acc0 = (uint32_t)lane == (k & 63u) ? c : acc0;
      if ((k & 63u) == 63u) {
        if (acc0) atomic_add_u32(out + de.out_base + (k - 63u) + (uint32_t)lane, acc0);
        acc0 = 0;
      }
    } else {
      if (lane == 0 && c) atomic_add_u32(out + de.out_base + k, c);
    }
    k++;
  };

  for (int side = 0; side < 2; side++) {
    const uint32_t a = side ? de.xb_cid : de.xa_cid, other = side ? de.xa_cid : de.xb_cid;
    if (a < n || !((side_mask >> side) & 1u)) continue;   // a tip has nothing behind it
    MPF_LOAD(par, other);
    int sp = 0;
    uint32_t node = a, d = 0;
    while (true) {
      // ---- expand `node` (inner, depth d < maxtrav): both children at once
      const uint2 kc = kids[node];
      const uint32_t c1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)kc.x);
      const uint32_t c2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)kc.y);
      MPF_LOAD(d1, c1);
      MPF_LOAD(d2, c2);
      const uint32_t dd = d + 1;
      const bool test = dd >= mintrav;
      const bool deeper = dd < maxtrav;
      fitch<S, VW, SPLIT>(u1, par, d2);
      fitch<S, VW, SPLIT>(u2, par, d1);
      uint32_t tot = 0;
      const bool own1 = dd > 1u || (child_mask & 1u), own2 = dd > 1u || (child_mask & 2u);
      uint32_t r1 = 0, r2 = 0;
      if (test) {
        uint32_t cost;
        if constexpr (MASKS) {
          uint32_t m1[VW], m2[VW];
          cost = join_mask<S, VW, SPLIT>(u1, d1, sv, m1) | (join_mask<S, VW, SPLIT>(u2, d2, sv, m2) << 16);
          if (own1) { r1 = mrow++; put_mask(m1, r1); }
          if (own2) { r2 = mrow++; put_mask(m2, r2); }
        } else {
          cost = join_cost<S, VW, SPLIT>(u1, d1, sv) | (join_cost<S, VW, SPLIT>(u2, d2, sv) << 16);
        }
        cost = valid ? cost : 0u;
        tot = wave_total<RED>(cost);
      }
      if (deeper && c2 >= n) {             // dd in [1, MAXD - 1]
        if (dd == 1u) pend1 = u2;
        else if (REGP == 2 && dd == 2u) pend2 = u2;
        else if constexpr (DEEP) {
          uint32_t *q = my_pend + (size_t)(dd - 1 - REGP) * (size_t)(S * VW * 64);
#pragma unroll
          for (int kk = 0; kk < S; kk++)
#pragma unroll
            for (int j = 0; j < VW; j++) q[(kk * VW + j) * 64] = u2.v[kk][j];
        } else {
#pragma unroll
          for (int kk = 0; kk < S; kk++)
#pragma unroll
            for (int j = 0; j < VW; j++) s_pend[wib][dd - 1 - REGP][kk * VW + j][lane] = u2.v[kk][j];
        }
      }
      if (own2) { stk[sp] = make_uint2(c2 | (dd << 24), (tot >> 16) | (r2 << 16)); sp++; }
      if (test && own1) emit(tot & 0xFFFFu, r1);
      if (own1 && deeper && c1 >= n) { par = u1; node = c1; d = dd; continue; }
      // ---- unwind: emit pending second children until one of them has to be expanded
      bool more = false;
      while (sp > 0) {
        sp--;
        const uint2 fr = stk[sp];
        const uint32_t fx = (uint32_t)__builtin_amdgcn_readfirstlane((int)fr.x);
        const uint32_t q = fx & 0xFFFFFFu, dq = fx >> 24;
        if (dq >= mintrav) {
          const uint32_t fy = (uint32_t)__builtin_amdgcn_readfirstlane((int)fr.y);
          emit(fy & 0xFFFFu, fy >> 16);
        }
        if (dq < maxtrav && q >= n) {
          if (dq == 1u) par = pend1;
          else if (REGP == 2 && dq == 2u) par = pend2;
          else if constexpr (DEEP) {
            const uint32_t *q = my_pend + (size_t)(dq - 1 - REGP) * (size_t)(S * VW * 64);
#pragma unroll
            for (int kk = 0; kk < S; kk++)
#pragma unroll
              for (int j = 0; j < VW; j++) par.v[kk][j] = q[(kk * VW + j) * 64];
          } else {
#pragma unroll
            for (int kk = 0; kk < S; kk++)
#pragma unroll
              for (int j = 0; j < VW; j++) par.v[kk][j] = s_pend[wib][dq - 1 - REGP][kk * VW + j][lane];
          }
          node = q; d = dq; more = true;
          break;
        }
      }
      if (!more) break;
    }
  }
#undef MPF_LOAD
  if constexpr (LANEACC) {
    if ((uint32_t)lane < (k & 63u) && acc0) atomic_add_u32(out + de.out_base + (k & ~63u) + (uint32_t)lane, acc0);
  }
  if (tile == 0 && lane == 0) ncand[scan] = k;
}

// host_out != nullptr (small batches): the workgroup that finishes last copies the n_out candidate costs to the host's
// pinned buffer itself, so that the batch needs no copy-back dispatch behind the kernel (`done` = a zeroed device word,
// left zeroed)
template <int S, int VW, int MAXD, int RED, bool SPLIT = false, bool MASKS = false, bool BIG = false, bool WM = false>
__global__ __launch_bounds__(256, (S == 4 && VW == 1 && MAXD <= 6 && !MASKS) ? 8 : 1) void k_scan_walk(const uint32_t *__restrict__ vec, const uint2 *__restrict__ kids,
                                                   uint32_t n, const WalkDesc *__restrict__ desc, int n_scans,
                                                   uint32_t *__restrict__ out, uint32_t *__restrict__ ncand, int Wp,
                                                   int tiles, int map, uint32_t *__restrict__ masks, uint2 *__restrict__ info,
                                                   uint32_t *__restrict__ host_out, uint32_t n_out, uint32_t *__restrict__ done)
{
  scan_walk_body<S, VW, MAXD, RED, SPLIT, MASKS, BIG, WM>(vec, kids, n, desc, n_scans, out, ncand, Wp, tiles, map, masks, info);
  if (!host_out) return;
  __shared__ int s_last;
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    const uint32_t ticket = __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = ticket == gridDim.x - 1;
  }
  __syncthreads();
  if (s_last) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    for (uint32_t i = threadIdx.x; i < n_out; i += blockDim.x) host_out[i] = __builtin_nontemporal_load(out + i);
    host_results_ready(host_out, n_out, done);
  }
}

// the device-walked scan at any radius (DEEP, above): with or without the candidates' masks
template <int S, int VW, int RED, bool SPLIT, bool MASKS, bool BIG>
__global__ __launch_bounds__(256) void k_scan_walk_deep(const uint32_t *__restrict__ vec, const uint2 *__restrict__ kids, uint32_t n,
                                                        const WalkDesc *__restrict__ desc, int n_scans, uint32_t *__restrict__ out,
                                                        uint32_t *__restrict__ ncand, int Wp, int tiles, uint32_t *__restrict__ masks,
                                                        uint2 *__restrict__ info, uint32_t *__restrict__ pend_g, int levels, uint32_t scan_base)
{
  scan_walk_body<S, VW, 255, RED, SPLIT, MASKS, BIG, false, true>(vec, kids, n, desc, n_scans, out, ncand, Wp, tiles, 0, masks, info, pend_g, levels, scan_base);
}

// ---------------------------------------------------------------- SPR scan, planned program (k_walk_plan + k_scan_prog)
//
// k_scan_walk pays, per expansion of the DFS, a DEPENDENT chain  kids[node] -> vectors of the two children -> arithmetic
// with nothing in flight behind it: eight waves per SIMD cannot hide two memory round trips per ~220 cycles of VALU work.
// The DFS itself is pure topology, so it is split off:
//   k_walk_plan : one wave per (scan part, gap end).  Lane h = heap index of an expansion (root 1, children 2h / 2h+1,
//                 depth <= 5 for radius 6): the lanes of one depth fetch kids[] together (6 dependent rounds for a whole
//                 neighbourhood instead of one per node), subtree sizes go bottom-up and pre-order positions top-down
//                 through ds_bpermute, and every expansion writes one 16-byte entry at its DFS position:
//                 (children, depth, which of the two candidates exist, their indices in the REFERENCE's order, whether
//                 the second child's up-vector must be kept, where this node's up-vector comes from).
//   k_scan_prog : one wave per (scan part, tile) runs the entries in order.  Entries arrive through the scalar cache two
//                 ahead, the two child vectors of entry e+1 are requested before entry e is combined (two register sets in
//                 rotation, unconditional requests so that the waits leave the younger ones in flight), candidate costs
//                 are dropped into lane (k & 63) of accumulator k >> 6 with v_writelane -- k = position in the reference's
//                 order, so no frame stack is needed to delay the second child's cost -- and leave as one atomic per lane
//                 at the end.  Counts are taken as popcount(hit) with s = 0 on lanes past the row end: no per-candidate
//                 v_not / v_cndmask; cost = 32 * VW * (valid lanes) - sum.
struct ProgEnt { uint32_t c1, c2, meta, k; };   // root entry (index 0): c1 = cid of the start up-vector, c2 = entries of this side
constexpr int kProgStride = 64;                  // entries per (scan part, gap end): root + at most 63 expansions (radius <= 6)
enum { PE_T1 = 16, PE_T2 = 32, PE_SAVE = 64, PE_PEND = 128 };

__device__ void walk_plan_item(const uint2 *__restrict__ kids, uint32_t n, const WalkDesc *__restrict__ desc, int n_scans,
                               ProgEnt *__restrict__ prog, uint32_t cid_mask, int item, int lane)
{
  item = __builtin_amdgcn_readfirstlane(item);
  if (item >= 2 * n_scans) return;
  const int scan = item >> 1, side = item & 1;
  const WalkDesc de = desc[scan];
  const int mintrav = (int)(de.trav & 0xFFu), maxtrav = (int)((de.trav >> 8) & 0xFFu);
  const uint32_t side_mask = (de.trav >> 16) & 3u, child_mask = (de.trav >> 18) & 3u;
  const uint32_t a = side ? de.xb_cid : de.xa_cid, other = side ? de.xa_cid : de.xb_cid;
  ProgEnt *P = prog + (size_t)item * kProgStride;
  if (a < n || !((side_mask >> side) & 1u) || maxtrav < 1) {          // wave-uniform: nothing behind this gap end
    if (lane == 0) P[0] = ProgEnt{other, 0u, 0u, 0u};
    return;
  }
  const int h = lane;                                   // heap index; lane 0 idles
  const int d = h ? 31 - __builtin_clz((unsigned)h) : -1;   // depth of the node this lane expands; its children sit at d + 1
  const int par = h >> 1, lc = (2 * h) & 63, rc = (2 * h + 1) & 63;
  const bool has_kids = h >= 1 && h < 32;
  uint32_t node = a;
  int ex = h == 1;
  uint2 kc = make_uint2(0u, 0u);
  for (int lev = 0; lev < maxtrav; lev++) {
    if (ex && d == lev) kc = kids[node];
    const uint32_t pc1 = (uint32_t)__shfl((int)kc.x, par, 64), pc2 = (uint32_t)__shfl((int)kc.y, par, 64);
    const int pex = __shfl(ex, par, 64);
    if (d == lev + 1) {
      node = (h & 1) ? pc2 : pc1;
      const bool own = lev + 1 > 1 || ((child_mask >> (h & 1)) & 1u);
      ex = pex && own && node >= n && lev + 1 < maxtrav;
    }
  }
  const int dd = d + 1;
  const bool tested = dd >= mintrav;
  const int t1 = (ex && tested && (dd > 1 || (child_mask & 1u))) ? 1 : 0;
  const int t2 = (ex && tested && (dd > 1 || (child_mask & 2u))) ? 1 : 0;
  // (every shuffle below is executed by ALL lanes: ds_bpermute reads nothing from a lane that is switched off)
  const int sx1 = __shfl(ex, lc, 64), sx2 = __shfl(ex, rc, 64);
  const int ex1 = has_kids ? sx1 : 0, ex2 = has_kids ? sx2 : 0;
  // candidates (T) and expansions (E) of the subtree hanging off this expansion, itself included; left child's share kept
  int T = t1 + t2, E = ex ? 1 : 0, TL = 0, EL = 0;
  for (int lev = maxtrav - 2; lev >= 0; lev--) {
    const int tl = __shfl(T, lc, 64), tr = __shfl(T, rc, 64), el = __shfl(E, lc, 64), er = __shfl(E, rc, 64);
    if (d == lev && ex && has_kids) {
      TL = ex1 ? tl : 0;
      EL = ex1 ? el : 0;
      T = t1 + t2 + TL + (ex2 ? tr : 0);
      E = 1 + EL + (ex2 ? er : 0);
    }
  }
  // pre-order position of the expansion and reference index K of its first child's candidate
  int pos = 1, K = side ? (int)de.pad1 : 0;
  for (int lev = 0; lev + 1 < maxtrav; lev++) {
    const int ppos = __shfl(pos, par, 64), pK = __shfl(K, par, 64), pt1 = __shfl(t1, par, 64), pt2 = __shfl(t2, par, 64);
    const int pTL = __shfl(TL, par, 64), pEL = __shfl(EL, par, 64);
    if (d == lev + 1 && ex) {
      if (!(h & 1)) { pos = ppos + 1; K = pK + pt1; }
      else { pos = ppos + 1 + pEL; K = pK + pt1 + pTL + pt2; }
    }
  }
  if (ex) {
    const uint32_t meta = (uint32_t)dd | (t1 ? PE_T1 : 0u) | (t2 ? PE_T2 : 0u) | (ex2 ? PE_SAVE : 0u) | ((h > 1 && (h & 1)) ? PE_PEND : 0u);
    P[pos] = ProgEnt{kc.x & cid_mask, kc.y & cid_mask, meta, (uint32_t)K | ((uint32_t)(K + t1 + TL) << 16)};
    if (h == 1) P[0] = ProgEnt{other, (uint32_t)(1 + E), 0u, (uint32_t)T};
  }
}

__global__ __launch_bounds__(256) void k_walk_plan(const uint2 *__restrict__ kids, uint32_t n, const WalkDesc *__restrict__ desc,
                                                   int n_scans, ProgEnt *__restrict__ prog, uint32_t cid_mask,
                                                   uint32_t *__restrict__ zero_ptr, uint32_t zero_words)
{
  // (the outputs of the scan that follows on the stream, cleared here: one memset dispatch less in front of it)
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < zero_words; i += gridDim.x * blockDim.x) zero_ptr[i] = 0u;
  walk_plan_item(kids, n, desc, n_scans, prog, cid_mask, (int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)), (int)(threadIdx.x & 63));
}

// acc[lane] = val for ONE lane (v_writelane_b32: value and lane select are wave-uniform scalars).  This clang has no builtin
// for it, so the LLVM intrinsic is declared directly (the way hip/amd_detail declares ds_bpermute); the compiler then moves
// the lane select through M0 itself (gfx9 allows one scalar register per VALU instruction) and tracks the hazards.
extern "C" __device__ int mpf_llvm_writelane(int, int, int) __asm("llvm.amdgcn.writelane.i32");
__device__ __forceinline__ uint32_t write_lane(uint32_t acc, uint32_t val, uint32_t lane)
{
  return (uint32_t)mpf_llvm_writelane((int)val, (int)lane, (int)acc);
}

// popcount of the sites where joining s onto fitch(u, d) finds a common state (the complement of join_cost)
template <int S, int VW>
__device__ __forceinline__ uint32_t join_hits(const Tile<S, VW> &u, const Tile<S, VW> &d, const Tile<S, VW> &s)
{
  uint32_t cnt = 0;
#pragma unroll
  for (int j = 0; j < VW; j++) {
    uint32_t any = u.v[0][j] & d.v[0][j];
#pragma unroll
    for (int k = 1; k < S; k++) any = b3_andor(u.v[k][j], d.v[k][j], any);
    uint32_t hit = b3_fitch(u.v[0][j], d.v[0][j], any) & s.v[0][j];
#pragma unroll
    for (int k = 1; k < S; k++) hit = b3_andor(b3_fitch(u.v[k][j], d.v[k][j], any), s.v[k][j], hit);
    cnt += (uint32_t)__builtin_popcount(hit);
  }
  return cnt;
}

struct ProgEnt4 { ProgEnt e[4]; };               // four entries = one 64-byte scalar load

// EXPR (experiments, MPF_PROG_EXPERIMENT): 0 = the kernel; 1 = no vector loads in the loop (children = register garbage:
// arithmetic + control only); 2 = loads only (one AND per loaded register instead of the Fitch arithmetic)
template <int S, int VW, bool BIG, int EXPR = 0, bool WM = false>
__global__ __launch_bounds__(64, (S * VW <= 4) ? 8 : (S * VW <= 8) ? 4 : 2) void k_scan_prog(const uint32_t *__restrict__ vec, const WalkDesc *__restrict__ desc, int n_scans,
                                                         const ProgEnt *__restrict__ prog, uint32_t *__restrict__ out,
                                                         uint32_t *__restrict__ ncand, int Wp, int tiles, int map,
                                                         uint32_t *__restrict__ host_out, uint32_t n_out, uint32_t *__restrict__ done,
                                                         unsigned long long *__restrict__ trace)
{
  const unsigned bid = blockIdx.x;
  // up-vectors of second children that wait for their turn, depths 2..5 (depth 1 stays in registers): a lane reads back
  // what it wrote, no synchronisation.
  // ONE wave per workgroup: scans differ in length by two orders of magnitude (2 .. 250 insertion tests), and a
  // multi-wave workgroup holds its LDS and its place on the CU until its longest wave is done.
  __shared__ uint32_t s_pend[4][S * VW][64];
  const int lane = threadIdx.x;
  int scan = -1, tile = 0;
  if (map == 0) {
    const int gw = (int)bid;
    if (gw < n_scans * tiles) { scan = gw / tiles; tile = gw - scan * tiles; }
  } else {
    // XCD-aware, as k_scan_walk: the (tile, scan) items, tile-major, in 8 contiguous chunks, one per blockIdx % 8 class
    const long total = (long)n_scans * tiles;
    const long chunk = (total + 7) / 8;
    const int cls = bid & 7;
    const long idx = (long)(bid >> 3);
    const long item = (long)cls * chunk + idx;
    if (idx < chunk && item < total) { tile = (int)(item / n_scans); scan = (int)(item - (long)tile * n_scans); }
  }
  if (scan >= 0) {
    // diagnostic timeline (engine option "scan_trace"): start / end of this wave on the 100 MHz clock + where it ran
    unsigned long long t_begin = 0;
    if (trace) t_begin = __builtin_amdgcn_s_memrealtime();
    const WalkDesc de = desc[scan];
    bool valid;
    const uint32_t w0 = (uint32_t)lane_word<VW>(tile, lane, Wp, valid);
    const uint32_t SW = (uint32_t)S * (uint32_t)Wp;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)vec, 0, 0x7FFFFFFF, 0x00020000);
    uint32_t voff[S];
#pragma unroll
    for (int k = 0; k < S; k++) voff[k] = (w0 + (uint32_t)k * (uint32_t)Wp) * 4u;
    // WM: `vec` is the word-major copy (the four state words of a site word side by side): one 16-byte load per lane and vector
#define MPF_LOAD(T, cid)                                                                       \
  do {                                                                                         \
    if constexpr (WM) {                                                                        \
      const auto x4 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, w0 * 16u, (uint32_t)(cid) * SW * 4u, 0); \
      T.v[0][0] = x4[0]; T.v[1][0] = x4[1]; T.v[2][0] = x4[2]; T.v[3][0] = x4[3];              \
    } else if constexpr (BIG) load_tile_g<S, VW>(T, vec + (size_t)(cid) * (size_t)SW, voff);   \
    else load_tile_b<S, VW>(T, rsrc, voff, (uint32_t)(cid) * SW * 4u);                          \
  } while (0)
    static_assert(!WM || (S == 4 && VW == 1 && !BIG), "word-major copy: DNA, one word per lane, below 2 GiB");
    Tile<S, VW> sv, par, pend1, ta1, ta2, tb1, tb2;
    MPF_LOAD(sv, de.s_cid);
    if (!valid) {                                  // lanes past the row end hold s = 0: they never hit ...
#pragma unroll
      for (int kk = 0; kk < S; kk++)
#pragma unroll
        for (int j = 0; j < VW; j++) sv.v[kk][j] = 0u;
    }
    const uint32_t full = 32u * (uint32_t)VW * (uint32_t)__builtin_popcountll(__ballot(valid));   // ... and are left out here
    uint32_t *const pend_lane = &s_pend[0][0][lane];
    uint32_t accp = 0;                             // lane e: packed hit counts of the two candidates of expansion e
    uint32_t total = 0;

    // one expansion: entry `en` (wave-uniform), children's vectors d1, d2 already requested
    auto step = [&](const ProgEnt &en, uint32_t e, const Tile<S, VW> &d1, const Tile<S, VW> &d2) {
      const uint32_t dd = en.meta & 15u;
      if (en.meta & PE_PEND) {                     // this node was a second child: its up-vector was parked at its depth
        if (dd == 2u) par = pend1;
        else {
          const uint32_t *p = pend_lane + ((dd - 1u) & 3u) * (uint32_t)(S * VW * 64);   // slot = depth & 3 (depths 2..5)
#pragma unroll
          for (int kk = 0; kk < S; kk++)
#pragma unroll
            for (int j = 0; j < VW; j++) par.v[kk][j] = p[(kk * VW + j) * 64];
        }
      }
      Tile<S, VW> u1, u2;
      if constexpr (EXPR == 2) {
#pragma unroll
        for (int kk = 0; kk < S; kk++)
#pragma unroll
          for (int j = 0; j < VW; j++) { u1.v[kk][j] = par.v[kk][j] & d1.v[kk][j]; u2.v[kk][j] = par.v[kk][j] & d2.v[kk][j]; }
        if (en.meta & (PE_T1 | PE_T2)) accp ^= u1.v[0][0] ^ u2.v[1][0];
      } else {
      fitch<S, VW>(u2, par, d1);
      fitch<S, VW>(u1, par, d2);
      if (en.meta & (PE_T1 | PE_T2)) {
        const uint32_t c = join_hits<S, VW>(u1, d1, sv) | (join_hits<S, VW>(u2, d2, sv) << 16);
        accp = write_lane(accp, wave_total<0>(c), e);
      }
      }
      if (en.meta & PE_SAVE) {
        if (dd == 1u) pend1 = u2;
        else {
          uint32_t *p = pend_lane + (dd & 3u) * (uint32_t)(S * VW * 64);
#pragma unroll
          for (int kk = 0; kk < S; kk++)
#pragma unroll
            for (int j = 0; j < VW; j++) p[(kk * VW + j) * 64] = u2.v[kk][j];
        }
      }
      par = u1;
    };

    for (int side = 0; side < 2; side++) {
      const ProgEnt *P = prog + ((size_t)scan * 2 + (size_t)side) * kProgStride;
      const ProgEnt4 *P4 = reinterpret_cast<const ProgEnt4 *>(P);
      // entries arrive four at a time through the scalar cache, one group ahead of their use, so that the wait in front
      // of a group's first use (which also covers the LDS traffic of the parked up-vectors) finds them long there
      ProgEnt4 cur = P4[0];
      const uint32_t ne = cur.e[0].c2;             // root + expansions
      total += cur.e[0].k;
      if (ne < 2u) continue;
      const uint32_t safe = cur.e[0].c1;           // any valid vector: what the clamped requests past the end fetch
      MPF_LOAD(par, safe);
      MPF_LOAD(ta1, cur.e[1].c1);
      MPF_LOAD(ta2, cur.e[1].c2);
      // STAGE(i, X, Y): entry g + i sits in set X; request entry g + i + 1 into set Y (unconditionally -- clamped past the
      // end --, so that the wait in front of X leaves those requests in flight), then combine X
#define MPF_STAGE(EN, NX, E, X1, X2, Y1, Y2)                                         \
  {                                                                                  \
    const bool more = (E) + 1u < ne;                                                 \
    const uint32_t n1 = more ? (NX).c1 : safe, n2 = more ? (NX).c2 : safe;           \
    if constexpr (EXPR == 1) { Y1 = X2; Y2 = X1; Y1.v[0][0] ^= n1; Y2.v[0][0] ^= n2; } \
    else { MPF_LOAD(Y1, n1); MPF_LOAD(Y2, n2); }                                     \
    step(EN, E, X1, X2);                                                             \
    if (!more) break;                                                                \
  }
      uint32_t g = 0;
      while (true) {
        const ProgEnt4 nxt = P4[g + 4u < (uint32_t)kProgStride ? (g >> 2) + 1u : (g >> 2)];
        if (g) MPF_STAGE(cur.e[0], cur.e[1], g, tb1, tb2, ta1, ta2)
        MPF_STAGE(cur.e[1], cur.e[2], g + 1u, ta1, ta2, tb1, tb2)
        MPF_STAGE(cur.e[2], cur.e[3], g + 2u, tb1, tb2, ta1, ta2)
        MPF_STAGE(cur.e[3], nxt.e[0], g + 3u, ta1, ta2, tb1, tb2)
        cur = nxt;
        g += 4u;
      }
#undef MPF_STAGE
      // lane e owns expansion e: its two candidates go to their places in the reference's order
      if ((uint32_t)lane >= 1u && (uint32_t)lane < ne) {
        const uint2 mk = *reinterpret_cast<const uint2 *>(&P[lane].meta);
        uint32_t *o = out + de.out_base;
        if (mk.x & PE_T1) { const uint32_t c = full - (accp & 0xFFFFu); if (c) atomic_add_u32(o + (mk.y & 0xFFFFu), c); }
        if (mk.x & PE_T2) { const uint32_t c = full - (accp >> 16); if (c) atomic_add_u32(o + (mk.y >> 16), c); }
      }
    }
#undef MPF_LOAD
    if (tile == 0 && lane == 0) ncand[scan] = total;
    if (trace) {
      const unsigned long long t_end = __builtin_amdgcn_s_memrealtime();
      if (lane == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg((4 /* HW_ID */) | (0 << 6) | (31 << 11));
        const unsigned xcc = __builtin_amdgcn_s_getreg((20 /* XCC_ID */) | (0 << 6) | (3 << 11));
        unsigned long long *t = trace + (size_t)bid * 4;
        t[0] = t_begin;
        t[1] = t_end;
        t[2] = ((unsigned long long)xcc << 32) | hw;
        t[3] = ((unsigned long long)(unsigned)scan << 32) | (total & 0xFFFFu) | ((unsigned)tile << 16);
      }
    }
  }
  if (!host_out) return;
  __shared__ int s_last;
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    const uint32_t ticket = __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = ticket == gridDim.x - 1;
  }
  __syncthreads();
  if (s_last) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    for (uint32_t i = threadIdx.x; i < n_out; i += blockDim.x) host_out[i] = __builtin_nontemporal_load(out + i);
    host_results_ready(host_out, n_out, done);
  }
}

// ================================================================ Sankoff (weighted parsimony) kernels
//
// reference: newviewSankoffParsimonyIterativeFastSIMD / evaluateSankoffParsimonyIterativeFastSIMD
// (sprparsimony.cpp:477-551, :880-961).  A vector is S rows of Wp 32-bit costs (one per pattern); a lane owns
// a pattern.  The min-plus transform  m[z] = min_x(v[x] + cost[z][x])  is the whole arithmetic: packed in
// registers per lane, the cost matrix read through the scalar cache (wave-uniform).  VALU-bound by
// nature (2*S*S add/min per transform), not MFMA-shaped (min-plus, not multiply-add).
// Exact 32-bit arithmetic (the reference's -short_off mode).  Cost matrices must be symmetric: only then
// is the length independent of root placement, which the directional-view formulation relies on.
// Every vector v is stored together with m(v) (at `moff` words behind it): a parent view is m(a) + m(b), a branch
// length min_x(a[x] + m(b)[x]), so newview needs ONE transform (of its result) instead of two, evaluate none, and an
// SPR candidate one (of the running up-vector) instead of four.

// Element type: PK = false: one 32-bit cost per lane (any cost matrix); PK = true: two 16-bit costs per lane
// (v_pk_add_u16 / v_pk_min_u16: the reference's default "short" arithmetic, usable while 2 n (max cost + 1) < 65536),
// half the instructions and half the bytes per pattern.  `cost` is [S*S] words; in PK mode each word holds the entry
// twice (c | c << 16) so that a uniform load IS the packed operand.
typedef unsigned short us2 __attribute__((ext_vector_type(2)));

template <bool PK> struct SnkT;
template <> struct SnkT<false> {
  typedef uint32_t E;
  static __device__ __forceinline__ E add(E a, E b) { return a + b; }
  static __device__ __forceinline__ E mn(E a, E b) { return min(a, b); }
  static __device__ __forceinline__ E cst(uint32_t c) { return c; }
  static __device__ __forceinline__ E inf() { return 0xFFFFFFFFu; }
  static __device__ __forceinline__ uint32_t sum(E a) { return a; }
  static __device__ __forceinline__ uint32_t wsum(E a, const uint32_t *__restrict__ pwgt, int e) { return a * pwgt[e]; }
};
template <> struct SnkT<true> {
  typedef us2 E;
  static __device__ __forceinline__ E add(E a, E b) { return a + b; }
  static __device__ __forceinline__ E mn(E a, E b) { return __builtin_elementwise_min(a, b); }
  static __device__ __forceinline__ E cst(uint32_t c) { return __builtin_bit_cast(us2, c); }
  static __device__ __forceinline__ E inf() { return (us2){0xFFFF, 0xFFFF}; }
  static __device__ __forceinline__ uint32_t sum(E a) { return (uint32_t)a.x + (uint32_t)a.y; }
  static __device__ __forceinline__ uint32_t wsum(E a, const uint32_t *__restrict__ pwgt, int e)
  {
    return (uint32_t)a.x * pwgt[2 * e] + (uint32_t)a.y * pwgt[2 * e + 1];
  }
};

template <int S, bool PK>
struct Costs { typename SnkT<PK>::E v[S]; };

// We = elements per state row (Wp patterns, or Wp / 2 pairs); a vector is S rows of We elements
template <int S, bool PK>
__device__ __forceinline__ void load_costs(Costs<S, PK> &t, const uint32_t *__restrict__ vec, uint32_t slot, int We, int e0)
{
  const typename SnkT<PK>::E *p = reinterpret_cast<const typename SnkT<PK>::E *>(vec) + (size_t)slot * (size_t)(S * We) + e0;
#pragma unroll
  for (int k = 0; k < S; k++) t.v[k] = p[(size_t)k * We];
}

// The same through a raw buffer: the slot's base and the row stride are uniform, so every row's address is the lane's own byte
// offset (voff, constant for the whole kernel) plus a SCALAR offset -- no 64-bit vector address arithmetic per row (the scan
// kernel spent a quarter of its vector instructions on that: profiles/r4/snk_scan_isa.txt).  The store may exceed 2 GiB: the scalar
// offset is 32-bit unsigned (a weighted C5 store with its transform copy is 1.6 + 1.6 GB, each half behind its own descriptor).
template <int S, bool PK>
__device__ __forceinline__ void load_costs_b(Costs<S, PK> &t, __amdgpu_buffer_rsrc_t rsrc, uint32_t slot, uint32_t row_bytes, uint32_t voff)
{
  const uint32_t base = slot * (uint32_t)S * row_bytes;
#pragma unroll
  for (int k = 0; k < S; k++)
    t.v[k] = __builtin_bit_cast(typename SnkT<PK>::E, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, base + (uint32_t)k * row_bytes, 0));
}

// m[z] = min_x(v[x] + cost[z][x]), four z at a time: four independent add -> min chains side by side (a dependent
// packed-math pair costs a wait state), each reading its row of the matrix through the scalar cache.
template <int S, bool PK>
__device__ __forceinline__ void mplus(Costs<S, PK> &m, const Costs<S, PK> &v, const uint32_t *__restrict__ cost)
{
  typedef SnkT<PK> T;
  static_assert(S % 4 == 0, "state count must be a multiple of 4");
#pragma unroll
  for (int z0 = 0; z0 < S; z0 += 4) {
    typename T::E a0 = T::add(v.v[0], T::cst(cost[(z0 + 0) * S])), a1 = T::add(v.v[0], T::cst(cost[(z0 + 1) * S]));
    typename T::E a2 = T::add(v.v[0], T::cst(cost[(z0 + 2) * S])), a3 = T::add(v.v[0], T::cst(cost[(z0 + 3) * S]));
#pragma unroll
    for (int x = 1; x < S; x++) {
      const typename T::E t0 = T::add(v.v[x], T::cst(cost[(z0 + 0) * S + x])), t1 = T::add(v.v[x], T::cst(cost[(z0 + 1) * S + x]));
      const typename T::E t2 = T::add(v.v[x], T::cst(cost[(z0 + 2) * S + x])), t3 = T::add(v.v[x], T::cst(cost[(z0 + 3) * S + x]));
      a0 = T::mn(a0, t0); a1 = T::mn(a1, t1); a2 = T::mn(a2, t2); a3 = T::mn(a3, t3);
    }
    m.v[z0] = a0; m.v[z0 + 1] = a1; m.v[z0 + 2] = a2; m.v[z0 + 3] = a3;
  }
}

// The 16-bit form of the transform on a PAIR-PACKED matrix (round 6): the device copy of the cost matrix carries, behind its S x S
// replicated entries (c | c << 16), S x S / 2 words holding two neighbouring entries of a row each (c[z][2j] | c[z][2j + 1] << 16);
// VOP3P's op_sel picks which half of the scalar operand feeds BOTH halves of the packed add, so a group of four rows needs 40
// scalar registers instead of 80 -- the matrix operand of the next group can be on its way while this one is combined (with 80 of
// the ~100 scalar registers tied up per group the loads could not run ahead: the transform waited on the scalar cache for about
// as long as its 780 packed operations took, profiles/r3/valu_rate.txt).
// (written as shuffles of the scalar word: the instruction selector folds them into the packed add's op_sel / op_sel_hi bits --
//  inline asm for the same instruction kept the compiler from scheduling around it and cost 1.2 KB of scratch per lane)
__device__ __forceinline__ us2 pk_add_lo(us2 v, uint32_t c2) { const us2 p = __builtin_bit_cast(us2, c2); return v + __builtin_shufflevector(p, p, 0, 0); }
__device__ __forceinline__ us2 pk_add_hi(us2 v, uint32_t c2) { const us2 p = __builtin_bit_cast(us2, c2); return v + __builtin_shufflevector(p, p, 1, 1); }
template <int S>
__device__ __forceinline__ void mplus(Costs<S, true> &m, const Costs<S, true> &v, const uint32_t *__restrict__ cost, int /* pair-packed */)
{
  typedef SnkT<true> T;
  static_assert(S % 4 == 0, "state count must be a multiple of 4");
  const uint32_t *__restrict__ c2 = cost + S * S;
  constexpr int H = S / 2;
#pragma unroll
  for (int z0 = 0; z0 < S; z0 += 4) {
    us2 a0, a1, a2, a3;
#pragma unroll
    for (int j = 0; j < H; j++) {
      const uint32_t w0 = c2[(z0 + 0) * H + j], w1 = c2[(z0 + 1) * H + j], w2 = c2[(z0 + 2) * H + j], w3 = c2[(z0 + 3) * H + j];
      const us2 t0 = pk_add_lo(v.v[2 * j], w0), t1 = pk_add_lo(v.v[2 * j], w1), t2 = pk_add_lo(v.v[2 * j], w2), t3 = pk_add_lo(v.v[2 * j], w3);
      if (j == 0) { a0 = t0; a1 = t1; a2 = t2; a3 = t3; }
      else { a0 = T::mn(a0, t0); a1 = T::mn(a1, t1); a2 = T::mn(a2, t2); a3 = T::mn(a3, t3); }
      const us2 u0 = pk_add_hi(v.v[2 * j + 1], w0), u1 = pk_add_hi(v.v[2 * j + 1], w1), u2 = pk_add_hi(v.v[2 * j + 1], w2), u3 = pk_add_hi(v.v[2 * j + 1], w3);
      a0 = T::mn(a0, u0); a1 = T::mn(a1, u1); a2 = T::mn(a2, u2); a3 = T::mn(a3, u3);
    }
    m.v[z0] = a0; m.v[z0 + 1] = a1; m.v[z0 + 2] = a2; m.v[z0 + 3] = a3;
  }
}
// (the scan kernels' call: pair-packed for the 16-bit form, the plain matrix otherwise)
template <int S, bool PK>
__device__ __forceinline__ void mplus_fast(Costs<S, PK> &m, const Costs<S, PK> &v, const uint32_t *__restrict__ cost)
{
  if constexpr (PK) mplus<S>(m, v, cost, 0);
  else mplus<S, PK>(m, v, cost);
}

template <int S, bool PK>
__device__ __forceinline__ void newview_one_snk(uint32_t *__restrict__ vec, size_t moff, const NvOp o, const uint32_t *__restrict__ cost,
                                                uint32_t *__restrict__ cntp, uint32_t nslots, int We, int tile, int lane)
{
  typedef SnkT<PK> T;
  bool valid;
  const int e0 = lane_word<1>(tile, lane, We, valid);
  Costs<S, PK> ma, mb, c, mc;
  load_costs<S, PK>(ma, vec + moff, o.a, We, e0);
  load_costs<S, PK>(mb, vec + moff, o.b, We, e0);
  typename T::E *dst = reinterpret_cast<typename T::E *>(vec) + (size_t)o.dst * (size_t)(S * We) + e0;
  typename T::E *mdst = reinterpret_cast<typename T::E *>(vec + moff) + (size_t)o.dst * (size_t)(S * We) + e0;
  typename T::E cur = T::inf();
#pragma unroll
  for (int z = 0; z < S; z++) {
    c.v[z] = T::add(ma.v[z], mb.v[z]);
    cur = T::mn(cur, c.v[z]);
  }
  mplus<S, PK>(mc, c, cost);
  if (valid) {
#pragma unroll
    for (int z = 0; z < S; z++) { dst[(size_t)z * We] = c.v[z]; mdst[(size_t)z * We] = mc.v[z]; }
  }
  const uint32_t cs = valid ? T::sum(cur) : 0u;
  const uint32_t tot = wave_total<0>(cs);
  if (lane == 0) cntp[(size_t)tile * nslots + o.dst] = tot;
}

template <int S, bool PK>
__global__ __launch_bounds__(256) void k_snk_newview(uint32_t *__restrict__ vec, size_t moff, const NvOp *__restrict__ ops, int n_ops,
                                                     const uint32_t *__restrict__ cost, uint32_t *__restrict__ cntp,
                                                     uint32_t nslots, int We, int tiles)
{
  const int lane = threadIdx.x & 63;
  int gw = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  gw = __builtin_amdgcn_readfirstlane(gw);
  if (gw >= n_ops * tiles) return;
  const int op = gw / tiles, tile = gw - op * tiles;
  newview_one_snk<S, PK>(vec, moff, ops[op], cost, cntp, nslots, We, tile, lane);
}

template <int S, bool PK>
__global__ __launch_bounds__(1024) void k_snk_newview_wg(uint32_t *__restrict__ vec, size_t moff, const NvOp *__restrict__ ops,
                                                         const int32_t *__restrict__ lev_off, int n_lev,
                                                         const uint32_t *__restrict__ cost, uint32_t *__restrict__ cntp,
                                                         uint32_t nslots, int We)
{
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int nw = (int)(blockDim.x >> 6);
  const int tile = blockIdx.x;
  for (int l = 0; l < n_lev; l++) {
    const int b = lev_off[l], e = lev_off[l + 1];
    for (int i = b + wave; i < e; i += nw) newview_one_snk<S, PK>(vec, moff, ops[i], cost, cntp, nslots, We, tile, lane);
    __syncthreads();
  }
}

// weighted length across branch (a, b): sum_ptn w * min_x(A[x] + m(B)[x])
template <int S, bool PK>
__global__ __launch_bounds__(256) void k_snk_evaluate(const uint32_t *__restrict__ vec, size_t moff, const EvOp *__restrict__ ops,
                                                      int n_ops, const uint32_t *__restrict__ cost,
                                                      const uint32_t *__restrict__ pwgt, uint32_t *__restrict__ out,
                                                      int We, int tiles)
{
  typedef SnkT<PK> T;
  const int lane = threadIdx.x & 63;
  int gw = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  gw = __builtin_amdgcn_readfirstlane(gw);
  if (gw >= n_ops * tiles) return;
  const int op = gw / tiles, tile = gw - op * tiles;
  const EvOp o = ops[op];
  bool valid;
  const int e0 = lane_word<1>(tile, lane, We, valid);
  Costs<S, PK> a, mb;
  load_costs<S, PK>(a, vec, o.a, We, e0);
  load_costs<S, PK>(mb, vec + moff, o.b, We, e0);
  typename T::E best = T::inf();
#pragma unroll
  for (int x = 0; x < S; x++) best = T::mn(best, T::add(a.v[x], mb.v[x]));
  const uint32_t c = valid ? T::wsum(best, pwgt, e0) : 0u;
  const uint32_t tot = wave_total<0>(c);
  if (lane == 0 && tot) atomic_add_u32(out + o.out, tot);
}

// SPR / stepwise scan over a host-planned program (same ops as k_scan).  With m() the min-plus transform:
//   CHAIN : U[d] = m(U[d-1]) + m(vec[sib]);  test: out += sum_ptn w * min_s(m(U[d])[s] + m(vec[own])[s] + m(S)[s])
//   JOIN  : out += sum_ptn w * min_s(m(vec[own])[s] + m(vec[sib])[s] + m(S)[s])
// and out is the FULL length of the rearranged tree (there is no additive base in the weighted case).
// ASYM (a cost matrix that is not symmetric): the reference scores a rearranged tree rooted at the edge it evaluates, parent
// state = row of the matrix (newview :477-551, evaluate :880-961: min_x(left[x] + min_y(cost[x][y] + right[y])), left = the
// far end of the edge handed to evaluateParsimony).  The stored transforms m(v)[z] = min_x(v[x] + cost[z][x]) have that
// orientation already (the viewer is the parent); what differs is the ROOT side of each test, which enters through the
// transposed matrix, mT(v)[y] = min_x(v[x] + cost[x][y]):
//   SPR test (testInsertParsimony evaluates p->next->next, whose back is r = q->back, the NEAR side):
//       min_y( m(S)[y] + m(vec[own])[y] + mT(U)[y] )        U = the near side's vector, computed along the chain
//   stepwise addition (evaluates the new inner node against p->back, the NEW TIP):
//       min_y( m(vec[own])[y] + m(vec[sib])[y] + mT(S)[y] )
// With a symmetric matrix mT = m and both collapse to the form below.  costT == nullptr selects ASYM = false.
template <int S, int MAXD, bool PK, bool ASYM = false, bool BUF = true>
__global__ __launch_bounds__(256, (S == 20 && MAXD <= 6) ? 2 : 1) void k_snk_scan(const uint32_t *__restrict__ vec, size_t moff, const ScanHdr *__restrict__ hdr,
                                                  int n_scans, const ScanOp *__restrict__ ops,
                                                  const uint32_t *__restrict__ cost, const uint32_t *__restrict__ pwgt,
                                                  uint32_t *__restrict__ out, int We, int tiles,
                                                  uint16_t *__restrict__ vals, uint32_t npat, uint32_t *__restrict__ vmax,
                                                  const uint32_t *__restrict__ costT)
{
  typedef SnkT<PK> T;
  const int lane = threadIdx.x & 63;
  // Work items = (scan, tile of 64 elements), scan-major: the 157 tile-waves of a C5 scan read WHOLE rows of its vectors between
  // them (a row is 40 KB of consecutive bytes).  Round 6 tried the Fitch kernels' XCD classes with a tile-major order inside a class
  // (an XCD's waves = neighbouring scans of one tile, for L2 reuse): 31.0 ms against 24.8 -- 256-byte pieces of 20 rows of many
  // vectors are a worse stream for the memory system than whole rows, and 4 MB of L2 hold 40 of these vectors' pieces anyway.
  int gw = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  gw = __builtin_amdgcn_readfirstlane(gw);
  if (gw >= n_scans * tiles) return;
  const int scan = gw / tiles, tile = gw - scan * tiles;
  const ScanHdr h = hdr[scan];
  uint32_t lane_max = 0;                           // (vals != nullptr: largest per-pattern length written by this lane)
  bool valid;
  const int e0 = lane_word<1>(tile, lane, We, valid);
  const uint32_t *mvec = vec + moff;
  // (descriptors over the two halves of the store: the vectors and their min-plus transforms; a half stays below 4 GiB or the
  //  launcher takes the 64-bit-pointer variant)
  const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc((void *)vec, 0, -1, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_m = __builtin_amdgcn_make_buffer_rsrc((void *)mvec, 0, -1, 0x00020000);
  const uint32_t row_bytes = (uint32_t)We * 4u, voff = (uint32_t)e0 * 4u;
  // (BUF = false: a half of the store beyond 4 GiB -- 64-bit pointers per row as before)
  auto ldv = [&](Costs<S, PK> &t, uint32_t slot) { if constexpr (BUF) load_costs_b<S, PK>(t, rs_v, slot, row_bytes, voff); else load_costs<S, PK>(t, vec, slot, We, e0); };
  auto ldm = [&](Costs<S, PK> &t, uint32_t slot) { if constexpr (BUF) load_costs_b<S, PK>(t, rs_m, slot, row_bytes, voff); else load_costs<S, PK>(t, mvec, slot, We, e0); };

  // MU[d] = m(U[d]): what the children of depth d and the test at depth d both need; U itself is never kept
  Costs<S, PK> ms, MU[MAXD + 1], t1, t2;
  if (ASYM && (h.pad & 1u)) {                       // stepwise addition: the new tip is the root side
    ldv(t2, h.s_slot);
    mplus_fast<S, PK>(ms, t2, costT);
  } else ldm(ms, h.s_slot);

  for (uint32_t i = h.op_begin; i < h.op_end; i++) {
    const ScanOp o = ops[i];
    const int d = (int)(o.meta & 0xFFu);
    const bool test = (o.meta >> 8) & 1u;
    const int kind = (int)((o.meta >> 16) & 0xFFu);
    if (kind == SCAN_ROOT) {
      ldm(MU[0], o.own);
      continue;
    }
    typename T::E best = T::inf();
    bool eval_op = false;           // (SCAN_EVAL exists in programs under an asymmetric matrix only: no code for it in the other variant)
    if constexpr (ASYM) eval_op = kind == SCAN_EVAL;
    if (eval_op) {                  // the current tree at the edge of a TIP q: min_x(vec[q][x] + m(S)[x]) (evaluate, :880-961)
      ldv(t1, o.sib);
#pragma unroll
      for (int s = 0; s < S; s++) best = T::mn(best, T::add(t1.v[s], ms.v[s]));
    } else
    ldm(t1, o.sib);       // m(vec[sib])
    if (eval_op) {
    } else if (kind == SCAN_JOIN) {
      ldm(t2, o.own);
#pragma unroll
      for (int s = 0; s < S; s++) best = T::mn(best, T::add(T::add(t1.v[s], t2.v[s]), ms.v[s]));
    } else {
#define MPF_SLEVEL(c)                                                                  \
  case c:                                                                              \
    if constexpr (c <= MAXD) {                                                         \
      _Pragma("unroll") for (int s = 0; s < S; s++) t2.v[s] = T::add(t1.v[s], MU[c - 1].v[s]); /* U[c] */ \
      if (ASYM && test) {   /* the near side is the root side of the test: mT(U) first, in MU[c]'s registers */ \
        mplus_fast<S, PK>(MU[c], t2, costT);                                                \
        ldm(t1, o.own);                                    \
        _Pragma("unroll") for (int s = 0; s < S; s++) best = T::mn(best, T::add(T::add(t1.v[s], MU[c].v[s]), ms.v[s])); \
      }                                                                                \
      mplus_fast<S, PK>(MU[c], t2, cost);                                                   \
      if (!ASYM && test) {                                                             \
        ldm(t1, o.own);                                    \
        _Pragma("unroll") for (int s = 0; s < S; s++) best = T::mn(best, T::add(T::add(t1.v[s], MU[c].v[s]), ms.v[s])); \
      }                                                                                \
    }                                                                                  \
    break;
      switch (d) {
        MPF_SLEVEL(1) MPF_SLEVEL(2) MPF_SLEVEL(3) MPF_SLEVEL(4) MPF_SLEVEL(5) MPF_SLEVEL(6)
        MPF_SLEVEL(7) MPF_SLEVEL(8) MPF_SLEVEL(9) MPF_SLEVEL(10) MPF_SLEVEL(11) MPF_SLEVEL(12)
        default: break;
      }
#undef MPF_SLEVEL
    }
    if (test || kind == SCAN_JOIN) {       // (a SCAN_EVAL op carries the test bit)
      const uint32_t c = valid ? T::wsum(best, pwgt, e0) : 0u;
      const uint32_t tot = wave_total<0>(c);
      if (lane == 0 && tot) atomic_add_u32(out + o.out, tot);
      // online UFBoot on the weighted engine: the tentative tree's per-pattern lengths (what pllComputeSankoffPatternParsimony
      // reads after the evaluate, sprparsimony.cpp:3341-3355), row o.out of vals[][npat]
      if (vals && valid) {
        if constexpr (PK) {
          *reinterpret_cast<us2 *>(vals + (size_t)o.out * npat + 2 * e0) = best;
          lane_max = max(lane_max, max((uint32_t)best.x, (uint32_t)best.y));
        } else {
          vals[(size_t)o.out * npat + e0] = (uint16_t)best;
          lane_max = max(lane_max, (uint32_t)best);
        }
      }
    }
  }
  if (vals) {
    uint32_t m = lane_max;
#pragma unroll
    for (int sh = 32; sh >= 1; sh >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, sh, 64));
    if (lane == 0 && m) atomicMax(vmax, m);
  }
}

// The weighted scan at ANY radius (k_snk_scan keeps m(U) of every level in registers: 12 levels for DNA, 6 otherwise): here the
// levels' transforms live in a scratch area of the wave in HBM, the one just computed stays in registers (a DFS mostly goes on one
// level down), a step back up re-reads its parent level -- the layout of k_scan_deep.  Launches are cut to the scratch (launch_scan).
template <int S, bool PK, bool ASYM, bool BUF>
__global__ __launch_bounds__(256) void k_snk_scan_deep(const uint32_t *__restrict__ vec, size_t moff, const ScanHdr *__restrict__ hdr,
                                                       int n_scans, const ScanOp *__restrict__ ops,
                                                       const uint32_t *__restrict__ cost, const uint32_t *__restrict__ pwgt,
                                                       uint32_t *__restrict__ out, int We, int tiles,
                                                       uint16_t *__restrict__ vals, uint32_t npat, uint32_t *__restrict__ vmax,
                                                       const uint32_t *__restrict__ costT, uint32_t *__restrict__ scratch, int levels)
{
  typedef SnkT<PK> T;
  typedef typename T::E E;
  const int lane = threadIdx.x & 63;
  int gw = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  gw = __builtin_amdgcn_readfirstlane(gw);
  if (gw >= n_scans * tiles) return;
  const int scan = gw / tiles, tile = gw - scan * tiles;
  const ScanHdr h = hdr[scan];
  uint32_t lane_max = 0;
  bool valid;
  const int e0 = lane_word<1>(tile, lane, We, valid);
  const uint32_t *mvec = vec + moff;
  const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc((void *)vec, 0, -1, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_m = __builtin_amdgcn_make_buffer_rsrc((void *)mvec, 0, -1, 0x00020000);
  const uint32_t row_bytes = (uint32_t)We * 4u, voff = (uint32_t)e0 * 4u;
  auto ldv = [&](Costs<S, PK> &t, uint32_t slot) { if constexpr (BUF) load_costs_b<S, PK>(t, rs_v, slot, row_bytes, voff); else load_costs<S, PK>(t, vec, slot, We, e0); };
  auto ldm = [&](Costs<S, PK> &t, uint32_t slot) { if constexpr (BUF) load_costs_b<S, PK>(t, rs_m, slot, row_bytes, voff); else load_costs<S, PK>(t, mvec, slot, We, e0); };
  uint32_t *mine = scratch + (size_t)gw * (size_t)levels * (size_t)(S * 64) + lane;
  auto put = [&](int lev, const Costs<S, PK> &t) {
    uint32_t *q = mine + (size_t)lev * (size_t)(S * 64);
#pragma unroll
    for (int k = 0; k < S; k++) q[k * 64] = __builtin_bit_cast(uint32_t, t.v[k]);
  };
  auto get = [&](int lev, Costs<S, PK> &t) {
    const uint32_t *q = mine + (size_t)lev * (size_t)(S * 64);
#pragma unroll
    for (int k = 0; k < S; k++) t.v[k] = __builtin_bit_cast(E, q[k * 64]);
  };
  Costs<S, PK> ms, last, t1, t2;                   // last = m(U[ld])
  int ld = -1;
  if (ASYM && (h.pad & 1u)) {
    ldv(t2, h.s_slot);
    mplus<S, PK>(ms, t2, costT);
  } else ldm(ms, h.s_slot);
  for (uint32_t i = h.op_begin; i < h.op_end; i++) {
    const ScanOp o = ops[i];
    const int d = (int)(o.meta & 0xFFu);
    const bool test = (o.meta >> 8) & 1u;
    const int kind = (int)((o.meta >> 16) & 0xFFu);
    if (kind == SCAN_ROOT) {
      ldm(last, o.own);
      put(0, last);
      ld = 0;
      continue;
    }
    E best = T::inf();
    bool eval_op = false;
    if constexpr (ASYM) eval_op = kind == SCAN_EVAL;
    if (eval_op) {
      ldv(t1, o.sib);
#pragma unroll
      for (int s = 0; s < S; s++) best = T::mn(best, T::add(t1.v[s], ms.v[s]));
    } else
    ldm(t1, o.sib);
    if (eval_op) {
    } else if (kind == SCAN_JOIN) {
      ldm(t2, o.own);
#pragma unroll
      for (int s = 0; s < S; s++) best = T::mn(best, T::add(T::add(t1.v[s], t2.v[s]), ms.v[s]));
    } else {
      if (d - 1 != ld) get(d - 1, last);
#pragma unroll
      for (int s = 0; s < S; s++) t2.v[s] = T::add(t1.v[s], last.v[s]);      // U[d]
      if (ASYM && test) {
        mplus<S, PK>(last, t2, costT);
        ldm(t1, o.own);
#pragma unroll
        for (int s = 0; s < S; s++) best = T::mn(best, T::add(T::add(t1.v[s], last.v[s]), ms.v[s]));
      }
      mplus<S, PK>(last, t2, cost);
      ld = d;
      if (d + 1 < levels) put(d, last);
      if (!ASYM && test) {
        ldm(t1, o.own);
#pragma unroll
        for (int s = 0; s < S; s++) best = T::mn(best, T::add(T::add(t1.v[s], last.v[s]), ms.v[s]));
      }
    }
    if (test || kind == SCAN_JOIN) {
      const uint32_t c = valid ? T::wsum(best, pwgt, e0) : 0u;
      const uint32_t tot = wave_total<0>(c);
      if (lane == 0 && tot) atomic_add_u32(out + o.out, tot);
      if (vals && valid) {
        if constexpr (PK) {
          *reinterpret_cast<us2 *>(vals + (size_t)o.out * npat + 2 * e0) = best;
          lane_max = max(lane_max, max((uint32_t)best.x, (uint32_t)best.y));
        } else {
          vals[(size_t)o.out * npat + e0] = (uint16_t)best;
          lane_max = max(lane_max, (uint32_t)best);
        }
      }
    }
  }
  if (vals) {
    uint32_t m = lane_max;
#pragma unroll
    for (int sh = 32; sh >= 1; sh >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, sh, 64));
    if (lane == 0 && m) atomicMax(vmax, m);
  }
}

template <int S, bool PK>
__global__ __launch_bounds__(256) void k_snk_pattern(const uint32_t *__restrict__ vec, size_t moff, uint32_t a_slot, uint32_t b_slot,
                                                     const uint32_t *__restrict__ cost, uint16_t *__restrict__ ptn, int We,
                                                     uint32_t *__restrict__ vmax)
{
  typedef SnkT<PK> T;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= We) return;
  Costs<S, PK> a, mb;
  load_costs<S, PK>(a, vec, a_slot, We, j);
  load_costs<S, PK>(mb, vec + moff, b_slot, We, j);
  typename T::E best = T::inf();
#pragma unroll
  for (int x = 0; x < S; x++) best = T::mn(best, T::add(a.v[x], mb.v[x]));
  if constexpr (PK) { ptn[2 * j] = best.x; ptn[2 * j + 1] = best.y; if (vmax) atomicMax(vmax, max((uint32_t)best.x, (uint32_t)best.y)); }
  else { ptn[j] = (uint16_t)best; if (vmax) atomicMax(vmax, (uint32_t)best); }
}

// compressSankoffDNA (reference sprparsimony.cpp:2636-2825): cost 0 for states in the tip's set, highest_cost otherwise
template <int S, bool PK>
__global__ __launch_bounds__(256) void k_snk_pack(uint32_t *__restrict__ vec, size_t moff, const uint8_t *__restrict__ codes, int n_taxa,
                                                  int n_patterns, const int32_t *__restrict__ inf_index, int n_inf,
                                                  int datatype, uint32_t highest, const uint32_t *__restrict__ cost, int We)
{
  typedef SnkT<PK> T;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int tip = blockIdx.y;
  if (j >= We || tip >= n_taxa) return;
  Costs<S, PK> v, mv;
  auto tip_cost = [&](int pattern, int k) -> uint32_t {
    if (pattern >= n_inf) return 0u;                               // padded patterns, :2766-2775
    const uint32_t m = state_mask(datatype, codes[(size_t)tip * n_patterns + inf_index[pattern]]);
    return ((m >> k) & 1u) ? 0u : highest;
  };
#pragma unroll
  for (int k = 0; k < S; k++) {
    if constexpr (PK) v.v[k] = (us2){(unsigned short)tip_cost(2 * j, k), (unsigned short)tip_cost(2 * j + 1, k)};
    else v.v[k] = tip_cost(j, k);
  }
  mplus<S, PK>(mv, v, cost);
  typename T::E *dst = reinterpret_cast<typename T::E *>(vec) + (size_t)tip * (size_t)(S * We) + j;
  typename T::E *mdst = reinterpret_cast<typename T::E *>(vec + moff) + (size_t)tip * (size_t)(S * We) + j;
#pragma unroll
  for (int k = 0; k < S; k++) { dst[(size_t)k * We] = v.v[k]; mdst[(size_t)k * We] = mv.v[k]; }
}

// ---------------------------------------------------------------- launch wrappers

static inline int tiles_of(const Geometry &g) { return (g.Wp + 64 * g.vw - 1) / (64 * g.vw); }
// weighted mode: elements per state row (patterns, or pattern pairs in the 16-bit packing)
static inline int snk_elems(const Geometry &g) { return g.snk16 ? g.Wp / 2 : g.Wp; }
#define MPF_DISPATCH_SNK(FN)                                                          \
  do {                                                                                \
    if (g.S == 4) { if (g.snk16) FN(4, true); else FN(4, false); }                    \
    else if (g.S == 20) { if (g.snk16) FN(20, true); else FN(20, false); }            \
    else { if (g.snk16) FN(32, true); else FN(32, false); }                           \
  } while (0)

hipError_t launch_pack_tips(hipStream_t st, const Geometry &g, uint32_t *vec, const uint8_t *codes, int n_taxa,
                            int n_patterns, const int32_t *site2ptn, int n_sites, int datatype,
                            const uint32_t *tip_slots)
{
  dim3 grid((g.Wp + 255) / 256, n_taxa), block(256);          // 4 waves x 64 words per block
  if (g.S == 4)
    hipLaunchKernelGGL(k_pack_tips<4>, grid, block, 0, st, vec, codes, n_taxa, n_patterns, site2ptn, n_sites, datatype,
                       tip_slots, g.Wp, g.shoff ? vec + g.shoff : nullptr);
  else if (g.S == 20)
    hipLaunchKernelGGL(k_pack_tips<20>, grid, block, 0, st, vec, codes, n_taxa, n_patterns, site2ptn, n_sites, datatype,
                       tip_slots, g.Wp, nullptr);
  else
    hipLaunchKernelGGL(k_pack_tips<32>, grid, block, 0, st, vec, codes, n_taxa, n_patterns, site2ptn, n_sites, datatype,
                       tip_slots, g.Wp, nullptr);
  return hipGetLastError();
}

#define MPF_DISPATCH_SV(FN, ...)                                                   \
  do {                                                                             \
    if (g.S == 4) {                                                                \
      if (g.vw == 1) { FN(4, 1, __VA_ARGS__); }                                    \
      else if (g.vw == 2) { FN(4, 2, __VA_ARGS__); }                               \
      else { FN(4, 4, __VA_ARGS__); }                                              \
    } else if (g.S == 20) {                                                        \
      FN(20, 1, __VA_ARGS__);      /* 20 states: one word per lane (engine.cpp, "words_per_lane") */ \
    } else {                                                                       \
      FN(32, 1, __VA_ARGS__);      /* 32 states: one word per lane */              \
    }                                                                              \
  } while (0)

hipError_t launch_newview(hipStream_t st, const Geometry &g, uint32_t *vec, const NvOp *ops, int n_ops, uint32_t *cntp,
                          uint32_t nslots)
{
  if (n_ops <= 0) return hipSuccess;
  const int tiles = tiles_of(g);
  const long waves = (long)n_ops * tiles;
  dim3 grid((unsigned)((waves + 3) / 4)), block(256);
  if (g.sankoff) {
    const int We = snk_elems(g), stiles = (We + 63) / 64;
    dim3 sgrid((unsigned)(((long)n_ops * stiles + 3) / 4));
#define SNK(S_, PK_) hipLaunchKernelGGL((k_snk_newview<S_, PK_>), sgrid, block, 0, st, vec, g.moff, ops, n_ops, g.cost, cntp, nslots, We, stiles)
    MPF_DISPATCH_SNK(SNK);
#undef SNK
    return hipGetLastError();
  }
#define NV(S_, VW_, RED_) hipLaunchKernelGGL((k_newview<S_, VW_, RED_>), grid, block, 0, st, vec, ops, n_ops, cntp, nslots, g.Wp, tiles)
#define NV2(S_, VW_, dummy) do { if (g.reduce == 0) NV(S_, VW_, 0); else NV(S_, VW_, 1); } while (0)
  MPF_DISPATCH_SV(NV2, 0);
#undef NV2
#undef NV
  return hipGetLastError();
}

int newview_tile(const Geometry &g);

hipError_t launch_newview_levels(hipStream_t st, const Geometry &g, uint32_t *vec, const NvOp *ops, const int32_t *lev_off,
                                 int n_lev, uint32_t *cntp, uint32_t nslots, uint32_t *cnt, uint32_t *done, const RefreshExtra &x)
{
  if (n_lev <= 0) return hipSuccess;
  // waves per workgroup: sixteen, or fewer where the levels are narrow (x.waves_hint: a partial tree in the addition phase has a
  // few ops per level -- sixteen waves per tile then only hold wave slots that other engines' launches could use)
  const unsigned nthreads = (x.waves_hint >= 1 && x.waves_hint < 16) ? 64u * (unsigned)x.waves_hint : 1024u;
  dim3 grid((unsigned)tiles_of(g)), block(nthreads);
  if (g.sankoff) {
    const int We = snk_elems(g);
    dim3 sgrid((unsigned)((We + 63) / 64));
#define SNK(S_, PK_) hipLaunchKernelGGL((k_snk_newview_wg<S_, PK_>), sgrid, block, 0, st, vec, g.moff, ops, lev_off, n_lev, g.cost, cntp, nslots, We)
    MPF_DISPATCH_SNK(SNK);
#undef SNK
    return hipGetLastError();
  }
  if (g.vw == 1 && g.nv_pipe) {                    // TW lanes per op on TW-word tiles, operands requested a round ahead
    RefreshExtra xs = x;
    if (g.shoff && g.S == 4) xs.shadow = vec + g.shoff;
    const int tw = newview_tile(g);
    // (a device-planned sweep: one extra 16-wave workgroup per 64 possible walk-plan items, at most as many as fit beside the refresh)
    const unsigned extra = x.wp_desc ? std::min(512u, (2u * x.wp_max_parts + 63u) / 64u) : 0u;
    dim3 qgrid((unsigned)(g.Wp / tw) + extra);
#define NQ(S_, TW_) hipLaunchKernelGGL((k_newview_wgq<S_, TW_>), qgrid, block, 0, st, vec, ops, lev_off, n_lev, cntp, nslots, g.Wp, cnt, done, xs)
    if (g.S == 4) { if (tw == 32) NQ(4, 32); else if (tw == 16) NQ(4, 16); else if (tw == 8) NQ(4, 8); else NQ(4, 4); }
    else if (g.S == 20) { if (tw == 32) NQ(20, 32); else if (tw == 16) NQ(20, 16); else if (tw == 8) NQ(20, 8); else NQ(20, 4); }
    else { if (tw == 32) NQ(32, 32); else if (tw == 16) NQ(32, 16); else if (tw == 8) NQ(32, 8); else NQ(32, 4); }
#undef NQ
    return hipGetLastError();
  }
  if (g.vw == 1) {                                 // half a wave per op on 32-word tiles
    dim3 hgrid((unsigned)(g.Wp / 32));
    if (g.S == 4) hipLaunchKernelGGL((k_newview_wgh<4, 0>), hgrid, block, 0, st, vec, ops, lev_off, n_lev, cntp, nslots, g.Wp, cnt, done, x);
    else if (g.S == 20) hipLaunchKernelGGL((k_newview_wgh<20, 0>), hgrid, block, 0, st, vec, ops, lev_off, n_lev, cntp, nslots, g.Wp, cnt, done, x);
    else hipLaunchKernelGGL((k_newview_wgh<32, 0>), hgrid, block, 0, st, vec, ops, lev_off, n_lev, cntp, nslots, g.Wp, cnt, done, x);
    return hipGetLastError();
  }
#define NW(S_, VW_, RED_) hipLaunchKernelGGL((k_newview_wg<S_, VW_, RED_>), grid, block, 0, st, vec, ops, lev_off, n_lev, cntp, nslots, g.Wp, cnt, done, x)
#define NW2(S_, VW_, dummy) do { if (g.reduce == 0) NW(S_, VW_, 0); else NW(S_, VW_, 1); } while (0)
  // (several words per lane: DNA only -- wider alphabets run one word per lane and took the branches above)
  if (g.S != 4) return hipErrorInvalidValue;
  if (g.vw == 2) NW2(4, 2, 0); else NW2(4, 4, 0);
#undef NW2
#undef NW
  return hipGetLastError();
}

hipError_t launch_newview_chains(hipStream_t st, const Geometry &g, uint32_t *vec, const NvOp *ops, const int32_t *wl_off,
                                 int n_lev, int n_ops, uint32_t *cntp, uint32_t nslots, uint32_t *cnt, uint32_t *done,
                                 const RefreshExtra &x)
{
  if (n_lev <= 0) return hipSuccess;
  dim3 grid((unsigned)tiles_of(g));
  RefreshExtra xs = x;
  if (g.shoff && g.S == 4 && g.vw == 1) xs.shadow = vec + g.shoff;
#define NC(S_, VW_, RED_) hipLaunchKernelGGL((k_newview_chain<S_, VW_, RED_, (S_ * VW_ <= 4 ? 4 : 2)>), grid, dim3(chain_waves<S_, VW_>() * 64), 0, st, vec, ops, wl_off, n_lev, cntp, nslots, g.Wp, cnt, done, n_ops, xs)
#define NC2(S_, VW_, dummy) do { if (g.reduce == 0) NC(S_, VW_, 0); else NC(S_, VW_, 1); } while (0)
  MPF_DISPATCH_SV(NC2, 0);
#undef NC2
#undef NC
  return hipGetLastError();
}

// tile of the "views_pipe" refresh: as set, or the smallest one that keeps the workgroups within one round of the chip's 256
// CUs (C3: 1568 words -> 8-word tiles, 196 workgroups; C5: 640 -> 4, 160; measured in profiles/r2/refresh_tile_experiments.txt)
int newview_tile(const Geometry &g)
{
  if (g.nv_tile) return g.nv_tile;
  for (int t = 4; t < 32; t *= 2)
    if (g.Wp / t <= 256) return t;
  return 32;
}
int tiles_for_levels(const Geometry &g) { return (!g.sankoff && g.vw == 1) ? g.Wp / (g.nv_pipe ? newview_tile(g) : 32) : tiles_for(g); }

hipError_t launch_cntsum(hipStream_t st, const Geometry &g, const NvOp *ops, int n_ops, const uint32_t *cntp,
                         uint32_t nslots, uint32_t *cnt, int tiles, uint32_t *cnt_host)
{
  if (n_ops <= 0) return hipSuccess;
  if (tiles <= 0) tiles = tiles_for(g);
  hipLaunchKernelGGL(k_cntsum, dim3((n_ops + 7) / 8), dim3(256), 0, st, ops, n_ops, cntp, nslots, tiles, cnt, cnt_host);
  return hipGetLastError();
}

int tiles_for(const Geometry &g) { return g.sankoff ? (snk_elems(g) + 63) / 64 : tiles_of(g); }

hipError_t launch_evaluate(hipStream_t st, const Geometry &g, const uint32_t *vec, const EvOp *ops, int n_ops,
                           uint32_t *out)
{
  if (n_ops <= 0) return hipSuccess;
  const int tiles = tiles_of(g);
  const long waves = (long)n_ops * tiles;
  dim3 grid((unsigned)((waves + 3) / 4)), block(256);
  if (g.sankoff) {
    const int We = snk_elems(g), stiles = (We + 63) / 64;
    dim3 sgrid((unsigned)(((long)n_ops * stiles + 3) / 4));
#define SNK(S_, PK_) hipLaunchKernelGGL((k_snk_evaluate<S_, PK_>), sgrid, block, 0, st, vec, g.moff, ops, n_ops, g.cost, g.pwgt, out, We, stiles)
    MPF_DISPATCH_SNK(SNK);
#undef SNK
    return hipGetLastError();
  }
#define EV(S_, VW_, RED_) hipLaunchKernelGGL((k_evaluate<S_, VW_, RED_>), grid, block, 0, st, vec, ops, n_ops, out, g.Wp, tiles)
#define EV2(S_, VW_, dummy) do { if (g.reduce == 0) EV(S_, VW_, 0); else EV(S_, VW_, 1); } while (0)
  MPF_DISPATCH_SV(EV2, 0);
#undef EV2
#undef EV
  return hipGetLastError();
}

hipError_t launch_scan(hipStream_t st, const Geometry &g, const uint32_t *vec, const ScanHdr *hdr, int n_scans,
                       const ScanOp *ops, uint32_t *out, int max_depth, uint32_t *host_out, uint32_t n_out, uint32_t *done,
                       uint16_t *vals, uint32_t npat, uint32_t *vmax)
{
  if (n_scans <= 0) return hipSuccess;
  const int tiles = tiles_of(g);
  dim3 block(256);
  unsigned nblocks;
  if (g.sankoff) {
    const int We = snk_elems(g), stiles = (We + 63) / 64;
    const long waves = (long)n_scans * stiles;
    dim3 sgrid((unsigned)((waves + 3) / 4));
    const bool buf_ok = (unsigned long long)g.moff * 4ull < (1ull << 32);      // each half of the store behind one 32-bit-offset descriptor
#define SNKSCAN3(S_, D_, PK_, A_) do { if (buf_ok) hipLaunchKernelGGL((k_snk_scan<S_, D_, PK_, A_, true>), sgrid, block, 0, st, vec, g.moff, hdr, n_scans, ops, g.cost, g.pwgt, out, We, stiles, vals, npat, vmax, g.costT); \
                                       else hipLaunchKernelGGL((k_snk_scan<S_, D_, PK_, A_, false>), sgrid, block, 0, st, vec, g.moff, hdr, n_scans, ops, g.cost, g.pwgt, out, We, stiles, vals, npat, vmax, g.costT); } while (0)
#define SNKSCAN2(S_, D_, PK_) do { if (g.costT) SNKSCAN3(S_, D_, PK_, true); else SNKSCAN3(S_, D_, PK_, false); } while (0)
#define SNKSCAN(S_, D_) do { if (g.snk16) SNKSCAN2(S_, D_, true); else SNKSCAN2(S_, D_, false); } while (0)
    if (max_depth > (g.S == 4 ? kMaxDepth : 6)) {
      // beyond the levels that fit registers: k_snk_scan_deep, cut into launches whose waves' scratch stays within the area
      if (!g.deep_scratch) return hipErrorInvalidValue;
      const int levels = max_depth + 1;
      const size_t per_wave = (size_t)levels * (size_t)(g.S * 64);
      long per_launch = (long)(g.deep_scratch_words / per_wave) / stiles;
      if (per_launch < 1) return hipErrorOutOfMemory;
      for (long s0 = 0; s0 < n_scans; s0 += per_launch) {
        const int ns = (int)std::min<long>(per_launch, n_scans - s0);
        dim3 dgrid((unsigned)(((long)ns * stiles + 3) / 4));
#define SNKD3(S_, PK_, A_) do { if (buf_ok) hipLaunchKernelGGL((k_snk_scan_deep<S_, PK_, A_, true>), dgrid, block, 0, st, vec, g.moff, hdr + s0, ns, ops, g.cost, g.pwgt, out, We, stiles, vals, npat, vmax, g.costT, g.deep_scratch, levels); \
                                else hipLaunchKernelGGL((k_snk_scan_deep<S_, PK_, A_, false>), dgrid, block, 0, st, vec, g.moff, hdr + s0, ns, ops, g.cost, g.pwgt, out, We, stiles, vals, npat, vmax, g.costT, g.deep_scratch, levels); } while (0)
#define SNKD2(S_, PK_) do { if (g.costT) SNKD3(S_, PK_, true); else SNKD3(S_, PK_, false); } while (0)
#define SNKD(S_) do { if (g.snk16) SNKD2(S_, true); else SNKD2(S_, false); } while (0)
        if (g.S == 4) SNKD(4); else if (g.S == 20) SNKD(20); else SNKD(32);
#undef SNKD
#undef SNKD2
#undef SNKD3
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
      }
      return hipSuccess;
    }
    if (g.S == 4) {
      if (max_depth <= 6) SNKSCAN(4, 6); else SNKSCAN(4, 12);
    } else {
      if (g.S == 20) SNKSCAN(20, 6); else SNKSCAN(32, 6);
    }
#undef SNKSCAN
#undef SNKSCAN2
#undef SNKSCAN3
    return hipGetLastError();
  }
  if (max_depth > scan_reg_depth(g.S, false)) {
    // beyond the levels that fit registers: k_scan_deep, cut into launches whose waves' scratch (levels x one tile each) stays
    // within scan_deep_scratch_words
    if (host_out || !g.deep_scratch) return hipErrorInvalidValue;
    const int levels = max_depth + 1;
    const size_t per_wave = (size_t)levels * (size_t)(g.S * g.vw * 64);
    long per_launch = (long)(g.deep_scratch_words / per_wave) / tiles;
    if (per_launch < 1) return hipErrorOutOfMemory;
    for (long s0 = 0; s0 < n_scans; s0 += per_launch) {
      const int ns = (int)std::min<long>(per_launch, n_scans - s0);
      dim3 dgrid((unsigned)(((long)ns * tiles + 3) / 4));
#define SD(S_, VW_, dummy)                                                                                                     \
      do {                                                                                                                     \
        if (g.reduce == 0) hipLaunchKernelGGL((k_scan_deep<S_, VW_, 0>), dgrid, block, 0, st, vec, hdr + s0, ns, ops, out, g.Wp, tiles, g.deep_scratch, levels); \
        else hipLaunchKernelGGL((k_scan_deep<S_, VW_, 1>), dgrid, block, 0, st, vec, hdr + s0, ns, ops, out, g.Wp, tiles, g.deep_scratch, levels);              \
      } while (0)
      MPF_DISPATCH_SV(SD, 0);
#undef SD
      const hipError_t e = hipGetLastError();
      if (e != hipSuccess) return e;
    }
    return hipSuccess;
  }
  if (g.map == 0) {
    const long waves = (long)n_scans * tiles;
    nblocks = (unsigned)((waves + 3) / 4);
  } else {
    const int ntc0 = (tiles + 7) / 8;                    // largest class
    const long per_class = ((long)n_scans * ntc0 + 3) / 4;
    nblocks = (unsigned)(per_class * 8);
  }
  dim3 grid(nblocks);
#define SC(S_, VW_, MAXD_, RED_) \
  hipLaunchKernelGGL((k_scan<S_, VW_, MAXD_, RED_>), grid, block, 0, st, vec, hdr, n_scans, ops, out, g.Wp, tiles, g.map, host_out, n_out, done)
#define SC2(S_, VW_, dummy)                                                          \
  do {                                                                               \
    if (max_depth <= 6) { if (g.reduce == 0) SC(S_, VW_, 6, 0); else SC(S_, VW_, 6, 1); } \
    else if constexpr (scan_reg_depth(S_, false) > 6) { if (g.reduce == 0) SC(S_, VW_, 12, 0); else SC(S_, VW_, 12, 1); } \
  } while (0)
  MPF_DISPATCH_SV(SC2, 0);
#undef SC2
#undef SC
  return hipGetLastError();
}

hipError_t launch_scan_walk(hipStream_t st, const Geometry &g, const uint32_t *vec, const uint2 *kids, int n_taxa,
                            const WalkDesc *desc, int n_scans, uint32_t *out, uint32_t *ncand, int max_depth,
                            uint32_t *masks, uint2 *info, uint32_t *host_out, uint32_t n_out, uint32_t *done, bool word_major)
{
  if (n_scans <= 0) return hipSuccess;
  const bool split = g.S >= 20;                                    // protein / 32-state data: states split over the wave halves
  const int tiles = split ? (g.Wp + 31) / 32 : (g.big ? (g.Wp + 63) / 64 : tiles_of(g));    // the 64-bit path is one word per lane
  if (max_depth > kWalkMaxDepth) {
    // 8: the per-depth LDS slots of the walk are sized for it.  Beyond: k_scan_walk_deep, the parked up-vectors in HBM scratch, cut
    // into launches whose waves' levels fit it (the batch's scans are laid out independently: out_base per scan)
    if (host_out || !g.deep_scratch) return hipErrorInvalidValue;
    const int levels = max_depth;                                  // (slots for depths 1 + REGP .. max_depth - 1)
    const int rows = split ? (g.S / 2) : g.S * (g.big ? 1 : g.vw);
    const size_t per_wave = (size_t)levels * (size_t)(rows * 64);
    long per_launch = (long)(g.deep_scratch_words / per_wave) / tiles;
    if (per_launch < 1) return hipErrorOutOfMemory;
    for (long s0 = 0; s0 < n_scans; s0 += per_launch) {
      const int ns = (int)std::min<long>(per_launch, n_scans - s0);
      dim3 dgrid((unsigned)(((long)ns * tiles + 3) / 4)), dblock(256);
#define SWD(S_, VW_, SPLIT_, BIG_)                                                                                                          \
      do {                                                                                                                                  \
        if (masks) {                                                                                                                        \
          if (g.reduce == 0) hipLaunchKernelGGL((k_scan_walk_deep<S_, VW_, 0, SPLIT_, true, BIG_>), dgrid, dblock, 0, st, vec, kids, (uint32_t)n_taxa, desc + s0, ns, out, ncand + s0, g.Wp, tiles, masks, info, g.deep_scratch, levels, (uint32_t)s0); \
          else hipLaunchKernelGGL((k_scan_walk_deep<S_, VW_, 1, SPLIT_, true, BIG_>), dgrid, dblock, 0, st, vec, kids, (uint32_t)n_taxa, desc + s0, ns, out, ncand + s0, g.Wp, tiles, masks, info, g.deep_scratch, levels, (uint32_t)s0); \
        } else {                                                                                                                            \
          if (g.reduce == 0) hipLaunchKernelGGL((k_scan_walk_deep<S_, VW_, 0, SPLIT_, false, BIG_>), dgrid, dblock, 0, st, vec, kids, (uint32_t)n_taxa, desc + s0, ns, out, ncand + s0, g.Wp, tiles, masks, info, g.deep_scratch, levels, (uint32_t)s0); \
          else hipLaunchKernelGGL((k_scan_walk_deep<S_, VW_, 1, SPLIT_, false, BIG_>), dgrid, dblock, 0, st, vec, kids, (uint32_t)n_taxa, desc + s0, ns, out, ncand + s0, g.Wp, tiles, masks, info, g.deep_scratch, levels, (uint32_t)s0); \
        }                                                                                                                                   \
      } while (0)
      if (g.big) {
        if (g.S == 4) SWD(4, 1, false, true); else if (g.S == 20) SWD(10, 1, true, true); else SWD(16, 1, true, true);
      } else if (g.S == 4) {
        if (g.vw == 1) SWD(4, 1, false, false); else SWD(4, 2, false, false);
      } else if (g.S == 20) SWD(10, 1, true, false);
      else if (g.S == 32) SWD(16, 1, true, false);
      else return hipErrorInvalidValue;
#undef SWD
      const hipError_t e = hipGetLastError();
      if (e != hipSuccess) return e;
    }
    return hipSuccess;
  }
  const long waves = (long)n_scans * tiles;
  dim3 block(256);
  unsigned nblocks;
  if (g.map == 0) nblocks = (unsigned)((waves + 3) / 4);
  else { const long chunk = (waves + 7) / 8; nblocks = (unsigned)(((chunk + 3) / 4) * 8); }
  dim3 grid(nblocks);
#define SWB(S_, VW_, MAXD_, RED_, SPLIT_, BIG_)                                                                                       \
  do {                                                                                                                                \
    if (masks)                                                                                                                        \
      hipLaunchKernelGGL((k_scan_walk<S_, VW_, MAXD_, RED_, SPLIT_, true, BIG_>), grid, block, 0, st, vec, kids, (uint32_t)n_taxa,    \
                         desc, n_scans, out, ncand, g.Wp, tiles, g.map, masks, info, host_out, n_out, done);                           \
    else                                                                                                                              \
      hipLaunchKernelGGL((k_scan_walk<S_, VW_, MAXD_, RED_, SPLIT_, false, BIG_>), grid, block, 0, st, vec, kids, (uint32_t)n_taxa,   \
                         desc, n_scans, out, ncand, g.Wp, tiles, g.map, masks, info, host_out, n_out, done);                           \
  } while (0)
#define SW(S_, VW_, MAXD_, RED_, SPLIT_) SWB(S_, VW_, MAXD_, RED_, SPLIT_, false)
#define SW2(S_, VW_, SPLIT_)                                                                                 \
  do {                                                                                                       \
    if (max_depth <= 6) { if (g.reduce == 0) SW(S_, VW_, 6, 0, SPLIT_); else SW(S_, VW_, 6, 1, SPLIT_); }    \
    else { if (g.reduce == 0) SW(S_, VW_, 8, 0, SPLIT_); else SW(S_, VW_, 8, 1, SPLIT_); }                   \
  } while (0)
  if (g.big) {
    // >= 2 GiB of vectors: one code path (one word per lane, DPP reduction), 64-bit addressing
    if (g.S == 4) { if (max_depth <= 6) SWB(4, 1, 6, 0, false, true); else SWB(4, 1, 8, 0, false, true); }
    else if (g.S == 20) { if (max_depth <= 6) SWB(10, 1, 6, 0, true, true); else SWB(10, 1, 8, 0, true, true); }
    else { if (max_depth <= 6) SWB(16, 1, 6, 0, true, true); else SWB(16, 1, 8, 0, true, true); }
  } else if (g.S == 4 && g.vw == 1 && word_major && g.shoff) {
    // the vectors from the word-major copy (the caller knows it is current): one 16-byte load per lane and vector
#define SWM(MAXD_, RED_)                                                                                                              \
  do {                                                                                                                                \
    if (masks)                                                                                                                        \
      hipLaunchKernelGGL((k_scan_walk<4, 1, MAXD_, RED_, false, true, false, true>), grid, block, 0, st, vec + g.shoff, kids,         \
                         (uint32_t)n_taxa, desc, n_scans, out, ncand, g.Wp, tiles, g.map, masks, info, host_out, n_out, done);         \
    else                                                                                                                              \
      hipLaunchKernelGGL((k_scan_walk<4, 1, MAXD_, RED_, false, false, false, true>), grid, block, 0, st, vec + g.shoff, kids,        \
                         (uint32_t)n_taxa, desc, n_scans, out, ncand, g.Wp, tiles, g.map, masks, info, host_out, n_out, done);         \
  } while (0)
    if (max_depth <= 6) { if (g.reduce == 0) SWM(6, 0); else SWM(6, 1); }
    else { if (g.reduce == 0) SWM(8, 0); else SWM(8, 1); }
#undef SWM
  } else if (g.S == 4) {
    if (g.vw == 1) SW2(4, 1, false); else SW2(4, 2, false);
  } else if (g.S == 20) {
    SW2(10, 1, true);
  } else {
    SW2(16, 1, true);
  }
#undef SW2
#undef SW
#undef SWB
  return hipGetLastError();
}

hipError_t launch_walk_plan(hipStream_t st, const uint2 *kids, int n_taxa, const WalkDesc *desc, int n_scans, void *prog,
                            uint32_t *zero_ptr, uint32_t zero_words)
{
  if (n_scans <= 0) return hipSuccess;
  const long waves = 2L * n_scans;
  // MPF_PROG_CID_MASK (experiments only): wrong results on purpose -- every child vector is taken from a small set, to time
  // the scan kernel with its memory traffic confined to the nearest cache
#ifdef MPF_EXPERIMENTS                     // (timing builds of tools/scan_bounds.sh only: `make EXPERIMENTS=1`)
  static const uint32_t cid_mask = getenv("MPF_PROG_CID_MASK") ? (uint32_t)strtoul(getenv("MPF_PROG_CID_MASK"), nullptr, 0) : 0xFFFFFFFFu;
#else
  const uint32_t cid_mask = 0xFFFFFFFFu;
#endif
  hipLaunchKernelGGL(k_walk_plan, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, kids, (uint32_t)n_taxa, desc, n_scans,
                     static_cast<ProgEnt *>(prog), cid_mask, zero_ptr, zero_words);
  return hipGetLastError();
}

size_t scan_prog_bytes(int n_scans) { return (size_t)n_scans * 2u * kProgStride * sizeof(ProgEnt); }

// pmin[i] = min of out[parts[i].x .. + parts[i].y): the cheapest candidate of every scan part, for callers that only want a
// sweep's best move per prune node -- 4 bytes per part cross the bus instead of every candidate's cost.  Sixteen lanes (one
// DPP row) per part read its costs side by side; a workgroup's sixteen minima leave through consecutive lanes (pmin may be
// pinned host memory).  The refresh's mutation counts ride along (cnt -> cnt_host, n_cnt words; nullptr: not wanted), and
// with `done` (a zeroed device word, left zeroed) the last workgroup raises pmin[n_parts] = 1 behind everything: a polling
// host thread then needs neither a copy dispatch nor the wake-up of a stream synchronisation.
__global__ __launch_bounds__(256) void k_part_min(const uint32_t *__restrict__ out, const uint2 *__restrict__ parts, int n_parts,
                                                  uint32_t *__restrict__ pmin, const uint32_t *__restrict__ cnt,
                                                  uint32_t *__restrict__ cnt_host, uint32_t n_cnt, uint32_t *__restrict__ done)
{
  __shared__ uint32_t s_min[16];
  __shared__ int s_last;
  const int grp = (int)(threadIdx.x >> 4), l = (int)(threadIdx.x & 15);
  const int i = (int)blockIdx.x * 16 + grp;
  uint32_t m = 0xFFFFFFFFu;
  if (i < n_parts) {
    const uint2 pr = parts[i];
    for (uint32_t k = (uint32_t)l; k < pr.y; k += 16u) m = min(m, out[pr.x + k]);
  }
  m = min(m, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)m, 0xB1, 0xF, 0xF, false));    // quad_perm [1,0,3,2]
  m = min(m, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)m, 0x4E, 0xF, 0xF, false));    // quad_perm [2,3,0,1]
  m = min(m, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)m, 0x141, 0xF, 0xF, false));   // row_half_mirror
  m = min(m, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)m, 0x140, 0xF, 0xF, false));   // row_mirror
  if (l == 0) s_min[grp] = m;
  __syncthreads();
  if (threadIdx.x < 16 && (int)blockIdx.x * 16 + (int)threadIdx.x < n_parts) pmin[blockIdx.x * 16 + threadIdx.x] = s_min[threadIdx.x];
  if (cnt_host)
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n_cnt; j += gridDim.x * blockDim.x) cnt_host[j] = __builtin_nontemporal_load(cnt + j);
  if (!done) return;
  // every wave that wrote to the host (wave 0: the minima; the first n_cnt threads of the grid: the counts): its words are in
  // host memory before the workgroup's ticket counts
  const uint32_t first = blockIdx.x * blockDim.x + (threadIdx.x & ~63u);
  if (threadIdx.x < 64u || (cnt_host && first < n_cnt)) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t ticket = __hip_atomic_fetch_add(done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    s_last = ticket == gridDim.x - 1;
    if (s_last) {
      __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
      __hip_atomic_store(pmin + n_parts, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

hipError_t launch_part_min(hipStream_t st, const uint32_t *out, const uint2 *parts, int n_parts, uint32_t *pmin, const uint32_t *cnt,
                           uint32_t *cnt_host, uint32_t n_cnt, uint32_t *done)
{
  if (n_parts <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_part_min, dim3((n_parts + 15) / 16), dim3(256), 0, st, out, parts, n_parts, pmin, cnt, cnt_host, n_cnt, done);
  return hipGetLastError();
}

// DNA (one or two words per lane) and, since round 6, the 20-row alphabets (one word per lane, row-major loads: C5's Fitch sweep ran on the
// device-walked kernel before, every wave chasing kids[] itself)
bool scan_prog_supported(const Geometry &g, int max_depth)
{
  return !g.sankoff && max_depth <= 6 && ((g.S == 4 && (g.vw == 1 || g.vw == 2)) || (g.S == 20 && g.vw == 1 && !g.big));
}

size_t scan_prog_blocks(const Geometry &g, int n_scans)
{
  const int vw = g.big ? 1 : g.vw;
  const long waves = (long)n_scans * ((g.Wp + 64 * vw - 1) / (64 * vw));
  return g.map == 0 ? (size_t)waves : (size_t)(((waves + 7) / 8) * 8);
}

hipError_t launch_scan_prog(hipStream_t st, const Geometry &g, const uint32_t *vec, const WalkDesc *desc, int n_scans,
                            const void *prog, uint32_t *out, uint32_t *ncand, uint32_t *host_out, uint32_t n_out, uint32_t *done,
                            unsigned long long *trace, bool word_major)
{
  if (n_scans <= 0) return hipSuccess;
  const int vw = g.big ? 1 : g.vw;
  const int tiles = (g.Wp + 64 * vw - 1) / (64 * vw);
  const long waves = (long)n_scans * tiles;
  dim3 block(64);                                  // one wave per workgroup (see the kernel)
  unsigned nblocks;
  if (g.map == 0) nblocks = (unsigned)waves;
  else { const long chunk = (waves + 7) / 8; nblocks = (unsigned)(chunk * 8); }
  dim3 grid(nblocks);
  const ProgEnt *pg = static_cast<const ProgEnt *>(prog);
#define SP(VW_, BIG_) hipLaunchKernelGGL((k_scan_prog<4, VW_, BIG_>), grid, block, 0, st, vec, desc, n_scans, pg, out, ncand, g.Wp, tiles, g.map, host_out, n_out, done, trace)
#ifdef MPF_EXPERIMENTS                     // (wrong results on purpose: never in the production library)
  static const int expr = getenv("MPF_PROG_EXPERIMENT") ? atoi(getenv("MPF_PROG_EXPERIMENT")) : 0;
  const bool wm = word_major && g.shoff && !g.big && vw == 1;
  if (expr == 1) hipLaunchKernelGGL((k_scan_prog<4, 1, false, 1>), grid, block, 0, st, vec, desc, n_scans, pg, out, ncand, g.Wp, tiles, g.map, host_out, n_out, done, trace);
  else if (expr == 2 && wm) hipLaunchKernelGGL((k_scan_prog<4, 1, false, 2, true>), grid, block, 0, st, vec + g.shoff, desc, n_scans, pg, out, ncand, g.Wp, tiles, g.map, host_out, n_out, done, trace);
  else if (expr == 2) hipLaunchKernelGGL((k_scan_prog<4, 1, false, 2>), grid, block, 0, st, vec, desc, n_scans, pg, out, ncand, g.Wp, tiles, g.map, host_out, n_out, done, trace);
  else
#endif
  if (g.S == 20) hipLaunchKernelGGL((k_scan_prog<20, 1, false>), grid, block, 0, st, vec, desc, n_scans, pg, out, ncand, g.Wp, tiles, g.map, host_out, n_out, done, trace);
  else if (g.big) SP(1, true);
  else if (vw == 1 && word_major && g.shoff)
    hipLaunchKernelGGL((k_scan_prog<4, 1, false, 0, true>), grid, block, 0, st, vec + g.shoff, desc, n_scans, pg, out, ncand, g.Wp, tiles, g.map, host_out, n_out, done, trace);
  else if (vw == 1) SP(1, false);
  else SP(2, false);
#undef SP
  return hipGetLastError();
}

hipError_t launch_site_counts(hipStream_t st, const Geometry &g, const uint32_t *vec, const EvOp *ops, int n_ops,
                              uint32_t *planes, const int32_t *ptn_first_site, int n_ptn, uint16_t *ptn_out)
{
  if (n_ops <= 0) return hipSuccess;
  const int tiles = tiles_of(g);
  const int n_chunks = (n_ops + kPlaneChunk - 1) / kPlaneChunk;
  const long waves = (long)n_chunks * tiles;
  dim3 grid((unsigned)((waves + 3) / 4)), block(256);
#define SP(S_, VW_, dummy) hipLaunchKernelGGL((k_site_planes<S_, VW_>), grid, block, 0, st, vec, ops, n_ops, planes, g.Wp, tiles)
  MPF_DISPATCH_SV(SP, 0);
#undef SP
  hipLaunchKernelGGL(k_pattern_sum, dim3((n_ptn + 255) / 256), dim3(256), 0, st, planes, n_chunks, g.Wp, ptn_first_site,
                     n_ptn, ptn_out);
  return hipGetLastError();
}

size_t site_planes_words(const Geometry &g, int n_ops)
{
  const int n_chunks = (n_ops + kPlaneChunk - 1) / kPlaneChunk;
  return (size_t)n_chunks * kPlanes * (size_t)g.Wp;
}

hipError_t launch_sankoff_pattern(hipStream_t st, const Geometry &g, const uint32_t *vec, uint32_t a, uint32_t b,
                                  uint16_t *ptn_out, uint32_t *vmax)
{
  const int We = snk_elems(g);
  dim3 grid((We + 255) / 256), block(256);
#define SNK(S_, PK_) hipLaunchKernelGGL((k_snk_pattern<S_, PK_>), grid, block, 0, st, vec, g.moff, a, b, g.cost, ptn_out, We, vmax)
  MPF_DISPATCH_SNK(SNK);
#undef SNK
  return hipGetLastError();
}

hipError_t launch_pack_tips_sankoff(hipStream_t st, const Geometry &g, uint32_t *vec, const uint8_t *codes, int n_taxa,
                                    int n_patterns, const int32_t *inf_index, int n_inf, int datatype)
{
  const int We = snk_elems(g);
  dim3 grid((We + 255) / 256, n_taxa), block(256);
#define SNK(S_, PK_) hipLaunchKernelGGL((k_snk_pack<S_, PK_>), grid, block, 0, st, vec, g.moff, codes, n_taxa, n_patterns, inf_index, n_inf, datatype, g.highest_cost, g.cost, We)
  MPF_DISPATCH_SNK(SNK);
#undef SNK
  return hipGetLastError();
}

}  // namespace mpf
