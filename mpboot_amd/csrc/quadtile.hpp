// quadtile.hpp -- device helpers shared by the persistent kernels (climb.hip, grow.hip): "quad" tiles of the vector store.
//
// A wavefront covers 16 * VW words of every state row; lane = 4 * w + g holds, for word group w, the KS states
// [g * KS, (g + 1) * KS) -- DNA: one state per lane, the four lanes of a DPP quad are the four states of a word; protein: five
// states per lane.  The only cross-state step of Fitch's rule, any = OR_k(a_k & b_k), is two quad_perm DPP ORs.
// Included inside an anonymous namespace of namespace mpf.
#pragma once

typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t rfl(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// old with lane `lane` (wave-uniform) replaced by val (wave-uniform): v_writelane_b32 (one scalar operand besides m0 on gfx9)
__device__ __forceinline__ int wlane(int val, int lane, int old)
{
  int keep;                                              // (m0 is the compiler's: handed back as it was)
  asm volatile("s_mov_b32 %1, m0\n\ts_mov_b32 m0, %3\n\tv_writelane_b32 %0, %2, m0\n\ts_mov_b32 m0, %1" : "+v"(old), "=&s"(keep) : "s"(val), "s"(lane));
  return old;
}

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

// One LDS atomic per wave whose result every lane needs (call with all lanes active).  Two obvious forms do not work here:
//  * `if (lane == 0) old = atomic(...); old = readfirstlane(old);` -- hipcc (ROCm 7.2) threads the lanes that skip the branch
//    past it into the next loop iteration with their own constant, so that they reach the readfirstlane without lane 0 (seen
//    in the ISA of the scan's task loop: an endless loop of lanes 1..63);
//  * issuing the atomic from all lanes with the value masked to lane 0 -- the atomic optimizer turns that into a loop over
//    the 64 active lanes (~2 us per call, measured: it was half of the scan phase).
// So the single-lane form is written out: EXEC narrowed to lane 0 around one ds instruction.
__device__ __forceinline__ uint32_t wave_fetch_add(uint32_t *p, uint32_t v, int lane)
{
  (void)lane;
  uint32_t old;
  unsigned long long sv;
  const uint32_t addr = (uint32_t)(uintptr_t)p;          // LDS offset (the low half of a flat LDS address)
  asm volatile("s_mov_b64 %1, exec\n\t"
               "s_mov_b64 exec, 1\n\t"
               "ds_add_rtn_u32 %0, %2, %3\n\t"
               "s_waitcnt lgkmcnt(0)\n\t"
               "s_mov_b64 exec, %1"
               : "=&v"(old), "=&s"(sv) : "v"(addr), "v"(v) : "memory");
  return rfl(old);
}
__device__ __forceinline__ uint32_t wave_fetch_sub(uint32_t *p, uint32_t v, int lane)
{
  (void)lane;
  uint32_t old;
  unsigned long long sv;
  const uint32_t addr = (uint32_t)(uintptr_t)p;
  asm volatile("s_mov_b64 %1, exec\n\t"
               "s_mov_b64 exec, 1\n\t"
               "ds_sub_rtn_u32 %0, %2, %3\n\t"
               "s_waitcnt lgkmcnt(0)\n\t"
               "s_mov_b64 exec, %1"
               : "=&v"(old), "=&s"(sv) : "v"(addr), "v"(v) : "memory");
  return rfl(old);
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

// the two other records of an inner record's node, in ring order
__device__ __forceinline__ void ring2(uint32_t c, uint32_t n, uint32_t &o1, uint32_t &o2)
{
  const uint32_t x = c - n, s = x - 3u * ((x * 43691u) >> 17);      // x mod 3 for x < 2^16
  o1 = s == 2u ? c - 2u : c + 1u;
  o2 = s == 0u ? c + 2u : c - 1u;
}

template <int KS, int VW>
struct QT { uint32_t v[KS][VW]; };

// KS == 4 is the WORD-MAJOR shape of four-state data (k_climb_many): a lane holds all four states of its VW words -- 64 * VW words
// per wavefront --, so Fitch's rule needs no step across lanes at all: 10 vector instructions per 64 words where the quad shape
// (KS == 1: the four lanes of a quad are the four states, 16 * VW words per wavefront) spends 24.  The quad shape exists for the
// single climb, which wants MANY small tiles (a workgroup each); a climb that is one workgroup wants the cheaper arithmetic.
template <int KS> constexpr bool kWordMajor = KS == 4;

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
#pragma unroll
      for (int h = 0; h < VW / 4; h++) {
        const v4u x = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[k] + 16u * (uint32_t)h, soff, 0);
        t.v[k][4 * h] = x[0]; t.v[k][4 * h + 1] = x[1]; t.v[k][4 * h + 2] = x[2]; t.v[k][4 * h + 3] = x[3];
      }
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
#pragma unroll
      for (int h = 0; h < VW / 4; h++) {
        v4u x; x[0] = t.v[k][4 * h]; x[1] = t.v[k][4 * h + 1]; x[2] = t.v[k][4 * h + 2]; x[3] = t.v[k][4 * h + 3];
        __builtin_amdgcn_raw_buffer_store_b128(x, rsrc, voff[k] + 16u * (uint32_t)h, soff, 0);
      }
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
    const uint32_t any = kWordMajor<KS> ? t : quad_or(t);
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
    const uint32_t any = kWordMajor<KS> ? t : quad_or(t);
    uint32_t hit = b3_fitch(u.v[0][j], d.v[0][j], any) & s.v[0][j];
#pragma unroll
    for (int k = 1; k < KS; k++) hit = b3_andor(b3_fitch(u.v[k][j], d.v[k][j], any), s.v[k][j], hit);
    if constexpr (!kWordMajor<KS>) hit = quad_or(hit);
    cost += (uint32_t)__builtin_popcount(~hit);
  }
  return cost;
}

