// ufboot.hip -- device side of the online UFBoot-MP bookkeeping (SURVEY §8f rank 1 + 2).
//
// Reference: for every insertion test of the SPR scan, IQTree::saveCurrentTree (iqtree.cpp:3271-3785) extracts the
// per-pattern parsimony lengths of the tentatively re-inserted tree (pllComputePatternParsimony,
// sprparsimony.cpp:3363-3392, fed by the 401 MB per-site counter arrays :294-376) and contracts them with every
// bootstrap weight vector (REPS, :3411-3449).
//
// Here neither the per-site counters nor a per-candidate pattern vector exist.  The per-site Fitch length of an
// unrooted tree does not depend on where it is rooted, so for the prune node p of a scan and any candidate edge e
//     pattern_pars(T with p re-inserted at e) = pattern_pars(T) - M(p, home) + M(p, e)
// where M(p, e) is the 1-bit-per-site "no common state" mask of joining p's subtree onto edge e -- exactly the
// word whose popcount the scan kernel already takes.  The scan kernel writes those masks (k_scan_walk<MASKS>),
// and the REPS of all candidates of a batch is ONE binary x small-integer matrix product
//     C[row][b] = sum_bit mask[row][bit] * w[b][bit]          (k_bitgemm, v_mfma_i32_16x16x64_i8)
//     rell(cand, b) = -( R_T[b] - C[home(cand)][b] + C[cand][b] )
// with R_T the REPS vector of the current tree.  The per-sample "new best or tie" decisions of the reference are
// order dependent (they draw random numbers on ties), so the device only extracts, per sample, the candidates
// that reach the running minimum in scan order (a chunked prefix-min), and the host replays just those events.
#include "ufboot.hpp"

#include <algorithm>
#include <atomic>
#include <cstdlib>

namespace mpf {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// ------------------------------------------------------------------------------------------------ join masks
// mask[op][w] = ~OR_k(vec[a]_k & vec[b]_k): the sites that mutate on the join (a, b) of a rooted traversal;
// their sum over the n-1 joins of the tree is the per-site length, so R_T = column sums of the product below.
__global__ __launch_bounds__(256) void k_join_masks(const uint32_t *__restrict__ vec, const EvOp *__restrict__ ops,
                                                    int n_ops, int S, int Wp, uint32_t *__restrict__ masks)
{
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  const int op = blockIdx.y;
  if (w >= Wp || op >= n_ops) return;
  const EvOp o = ops[op];
  const uint32_t *a = vec + (size_t)o.a * (size_t)(S * Wp) + w, *b = vec + (size_t)o.b * (size_t)(S * Wp) + w;
  uint32_t any = 0;
  for (int k = 0; k < S; k++) any |= a[(size_t)k * Wp] & b[(size_t)k * Wp];
  masks[(size_t)op * Wp + w] = ~any;
}

// ------------------------------------------------------------------------------------------------ bit GEMM
// A: masks [R][Wp] 32-bit words (row-major, bit j of word i = site 32 i + j), R padded to the workgroup tile.
// Wt: bootstrap weights as signed bytes (<= 127 per plane), laid out so that one k-block (64 sites = 2 words) of a
// 16-column group is the 1 KiB the MFMA B-fragments of a wave read lane-linearly:
//     Wt[((kblk * (Bp/16) + cg) * 4 + h) * 16 + c][j]  = weight of site 64 kblk + 16 h + j in sample 16 cg + c
// (h = lane >> 4, c = lane & 15, j = byte in the lane's 16-byte fragment).  The A fragment of lane (r = lane & 15, h)
// is the same 16 sites of row r, expanded from bits to bytes in registers: nibble * 0x00204081 & 0x01010101
// (both operands use the same (h, j) -> site map, so the sum over k is the intended one whatever order the
// hardware assigns to the 64 k values).
// Workgroup = 8 waves, tile 256 rows x 256 samples, wave tile 32 x 256 (2 x 16 MFMA tiles, 128 accumulator registers):
// each weight byte is fetched once per 256 rows and each expanded A fragment feeds 16 MFMAs.  KS k-blocks per stage, NS
// stages in an LDS ring; weights AND mask words arrive by LDS-DMA while older stages are multiplied (counted vmcnt waits and a
// raw s_barrier: __syncthreads() would drain the DMA requests of the younger stages).
// blockIdx.y splits K; partial products are added with integer atomics (exact, order-independent).
// MT x NT MFMA tiles per wave, WM x WN waves (tile = 16 MT WM rows x 16 NT WN samples).
// Where the time goes (profiles/r2/gemm_bounds.txt): the C3 sweep's main launch (110 592 rows) runs at 0.63 of the nominal
// int8 peak; with expansion, fragment reads and copies all removed the same loop reaches only 0.60 of it over the whole leg
// (0.45 with everything in), so what is left to gain inside the loop is the global -> LDS copy's issue cost (0.45 -> 0.57
// without it); tile shape, ring depth and k-blocks per stage all land within 0.43-0.49.
// EXPR (measurements only, results wrong): 1 = no bit -> byte expansion of the A fragments, 2 = B fragments read from LDS once
// per stage instead of once per k-block, 3 = no global -> LDS copies inside the loop, 4 = all three (bare MFMA stream);
// 6 (results right) = both waves of a SIMD issue their copies at the top of the stage, as before the staggering
template <int MT, int NT, int WM, int WN, int KS, int NS, int EXPR = 0, bool M32 = false>
__global__ __launch_bounds__(64 * WM * WN) void k_bitgemm(const uint32_t *__restrict__ masks, int Wp, const uint8_t *__restrict__ Wt,
                                                 int Bp, int32_t *__restrict__ C, int mult, int atomic, int row_blocks,
                                                 int kb_per_split, const uint32_t *__restrict__ rowsel, const uint32_t *__restrict__ row_limit)
{
  constexpr int TM = 16 * MT * WM, TN = 16 * NT * WN, NTH = 64 * WM * WN;
  constexpr int BT = TN * 64;                        // bytes of one k-block of the B tile (16 KiB)
  constexpr int LD = BT / (NTH * 16);                // 16-byte loads per thread per k-block
  constexpr int AL = KS / 2;                         // 16-byte pieces per row of the A stage tile
  constexpr int AT = AL * TM * 16;                   // bytes of the A tile of one stage
  static_assert(KS % 2 == 0 && TM % 64 == 0, "stage shape");
  extern __shared__ __attribute__((aligned(16))) uint8_t s_raw[];
  // NS stages in a ring: [NS][KS][BT] weights, then [NS][AL][TM][4] mask words -- both filled by LDS-DMA
  uint8_t *s_b = s_raw;
  uint8_t *s_a = s_raw + (size_t)NS * KS * BT;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WN, wc = wave % WN;
  constexpr bool STAGGER = EXPR != 6 && KS >= 4;
  // workgroups are dealt round-robin to the 8 XCDs: the 32 concurrent workgroups of an XCD share ONE column block,
  // so its slab of Wt streams through that XCD's L2 once per round
  const int col_blocks = Bp / TN;
  const int xcd = blockIdx.x & 7, grp = blockIdx.x >> 3;
  int cb, rb;
  if (col_blocks <= 8 && (8 % col_blocks) == 0) { cb = xcd % col_blocks; rb = grp * (8 / col_blocks) + xcd / col_blocks; }
  else { cb = (int)(blockIdx.x % (unsigned)col_blocks); rb = (int)(blockIdx.x / (unsigned)col_blocks); }
  if (rb >= row_blocks) return;
  // row_limit (a climb's batch): rows from *row_limit on lie behind the batch's certain end -- nobody reads their products
  if (row_limit && (uint32_t)(rb * TM) >= __builtin_nontemporal_load(row_limit)) return;
  const int r = lane & 15, h = lane >> 4;
  const int nkb = Wp >> 1;
  const int kb_begin = blockIdx.y * kb_per_split, kb_end = min(nkb, kb_begin + kb_per_split);
  if (kb_begin >= kb_end) return;

  // M32: the same wave tile as (MT / 2) x (NT / 2) tiles of v_mfma_i32_32x32x32_i8, two per 64-site k-block -- half the
  // matrix instructions per product (an MFMA holds the SIMD's vector issue for 8 cycles whatever its size)
  constexpr int MT2 = MT / 2, NT2 = NT / 2;
  static_assert(!M32 || (MT % 2 == 0 && NT % 2 == 0), "32x32 tiles pair the 16x16 ones");
  v4i acc[M32 ? 1 : MT][M32 ? 1 : NT];
  v16i acc32[M32 ? MT2 : 1][M32 ? NT2 : 1];
  if constexpr (M32) {
#pragma unroll
    for (int i = 0; i < MT2; i++)
#pragma unroll
      for (int j = 0; j < NT2; j++)
#pragma unroll
        for (int q = 0; q < 16; q++) acc32[i][j][q] = 0;
  } else {
#pragma unroll
    for (int i = 0; i < MT; i++)
#pragma unroll
      for (int j = 0; j < NT; j++) acc[i][j] = (v4i){0, 0, 0, 0};
  }
  const int c32 = lane & 31, hh = lane >> 5;       // M32: fragment row / column of the lane and its half of the 32 k-slots

  const uint8_t *wt_tile = Wt + (size_t)cb * (TN / 16) * 1024;
  const size_t wt_kstride = (size_t)(Bp / 16) * 1024;
  // output row i multiplies mask row rowsel[i] (cut-off filter: only the candidates that are saved), or row i itself
  const bool a_wave = tid < TM;                      // wave-uniform (TM % 64 == 0): these waves also fetch the mask rows
  const uint32_t arow_id = rowsel ? rowsel[rb * TM + (tid % TM)] : (uint32_t)(rb * TM + (tid % TM));
  const uint32_t *arow = masks + (size_t)arow_id * Wp;

  // Stage copy, all of it global -> LDS directly (LDS-DMA, 16 bytes per lane: lane i of a wave lands at the wave's LDS base
  // + 16 i): the weights in the tile's own linear layout, the mask words as [piece][row][4 words].  No staging registers,
  // no ds_write.  kb_ is a multiple of KS below nkb (nkb % KS == 0): blocks past kb_end are fetched but never multiplied.
  typedef __attribute__((address_space(3))) void lds_void;
#define MPF_GLOAD_KB(kq_, s_, st_)                                                                \
  do {                                                                                            \
    const uint4 *src_ = reinterpret_cast<const uint4 *>(wt_tile + (size_t)((kq_) + (s_)) * wt_kstride); \
    _Pragma("unroll") for (int i_ = 0; i_ < LD; i_++)                                             \
      __builtin_amdgcn_global_load_lds(src_ + tid + NTH * i_,                                     \
          (lds_void *)(s_b + ((size_t)(st_) * KS + (s_)) * BT + (size_t)((tid & ~63) + NTH * i_) * 16), 16, 0, 0); \
    if (((s_) & 1) == 0 && a_wave)                                                                \
      __builtin_amdgcn_global_load_lds(reinterpret_cast<const uint4 *>(arow + 2 * (kq_) + 2 * (s_)), \
          (lds_void *)(s_a + (size_t)(st_) * AT + (size_t)(((s_) >> 1) * TM + (tid & ~63)) * 16), 16, 0, 0); \
  } while (0)
#define MPF_GLOAD(kb_, st_)                                                                       \
  do {                                                                                            \
    const int kq0_ = min((kb_), nkb - KS);                                                        \
    _Pragma("unroll") for (int s0_ = 0; s0_ < KS; s0_++) MPF_GLOAD_KB(kq0_, s0_, st_);            \
  } while (0)
  // the stage needed next has landed when at most the NS - 2 younger stages' requests of this wave are outstanding
#define MPF_STAGE_WAIT()                                                                          \
  do {                                                                                            \
    if (a_wave) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * (KS * LD + AL)) : "memory");  \
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * (KS * LD)) : "memory");              \
  } while (0)
#pragma unroll
  for (int s = 0; s < NS - 1; s++) MPF_GLOAD(kb_begin + s * KS, s);
  MPF_STAGE_WAIT();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  int st = 0;
  for (int kb = kb_begin; kb < kb_end; kb += KS) {
    // clamped: the last stages prefetch valid blocks they never use.  The stage written is the one multiplied in the
    // previous iteration (every wave has passed the barrier behind it)
    // the copy's issue takes a wave off the matrix core for a good thousand cycles per stage: the two waves that share a
    // SIMD (w and w + 4) take turns -- one issues here, the other behind the first half of the stage's MFMAs
    const bool late = STAGGER && ((wave >> 2) & 1);
    if constexpr (EXPR != 3 && EXPR != 4) { if (!late) MPF_GLOAD(kb + (NS - 1) * KS, st == 0 ? NS - 1 : st - 1); }
    const uint8_t *sb = s_b + (size_t)st * KS * BT;
    const uint32_t *sa = reinterpret_cast<const uint32_t *>(s_a + (size_t)st * AT);
    // straight-line body: the mask words and B fragments of k-block s+1 are requested from LDS (words first: LDS answers in
    // order, and the words are what the next k-block needs first) while the MFMAs of k-block s run
    if constexpr (M32) {
      // k-block s = two MFMA k-steps (kk): k-slot (hh, j) of step kk is site 64 kblk + 16 (2 kk + hh) + j, for A and B alike.
      // The B fragment of lane (c32, hh) sits in the 16-column group c32 / 16 of the pair, row (2 kk + hh, c32 % 16) of its 1 KiB
      v4i bf[2][NT];
      uint32_t aw[2][2 * MT2];
      const uint32_t boff = (uint32_t)(((c32 >> 4) * 4 + hh) * 256 + (c32 & 15) * 16);
#define MPF_AWORD32(s_, kk_, i_) sa[(((2 * (s_) + (kk_)) >> 2) * TM + wr * 16 * MT + 32 * (i_) + c32) * 4 + ((2 * (s_) + (kk_)) & 3)]
#define MPF_BFRAG32(s_, kk_, j_) *reinterpret_cast<const v4i *>(sb + (size_t)(s_) * BT + (size_t)(wc * NT + 2 * (j_)) * 1024 + (kk_) * 512 + boff)
#pragma unroll
      for (int kk = 0; kk < 2; kk++)
#pragma unroll
        for (int i = 0; i < MT2; i++) aw[0][kk * MT2 + i] = MPF_AWORD32(0, kk, i);
#pragma unroll
      for (int kk = 0; kk < 2; kk++)
#pragma unroll
        for (int j = 0; j < NT2; j++) bf[0][kk * NT2 + j] = MPF_BFRAG32(0, kk, j);
#pragma unroll
      for (int s = 0; s < KS; s++) {
        if (s + 1 < KS) {
#pragma unroll
          for (int kk = 0; kk < 2; kk++)
#pragma unroll
            for (int i = 0; i < MT2; i++) aw[(s + 1) & 1][kk * MT2 + i] = MPF_AWORD32(s + 1, kk, i);
#pragma unroll
          for (int kk = 0; kk < 2; kk++)
#pragma unroll
            for (int j = 0; j < NT2; j++) bf[(s + 1) & 1][kk * NT2 + j] = MPF_BFRAG32(s + 1, kk, j);
        }
#pragma unroll
        for (int kk = 0; kk < 2; kk++)
#pragma unroll
          for (int i = 0; i < MT2; i++) {
            const uint32_t bits = (aw[s & 1][kk * MT2 + i] >> (hh * 16)) & 0xFFFFu;
            v4i af;
            af.x = (int)((((bits)&0xFu) * 0x00204081u) & 0x01010101u);
            af.y = (int)((((bits >> 4) & 0xFu) * 0x00204081u) & 0x01010101u);
            af.z = (int)((((bits >> 8) & 0xFu) * 0x00204081u) & 0x01010101u);
            af.w = (int)((((bits >> 12) & 0xFu) * 0x00204081u) & 0x01010101u);
#pragma unroll
            for (int j = 0; j < NT2; j++) acc32[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af, bf[s & 1][kk * NT2 + j], acc32[i][j], 0, 0, 0);
          }
        if constexpr (EXPR != 3 && EXPR != 4) { if (s == KS / 2 - 1 && late) MPF_GLOAD(kb + (NS - 1) * KS, st == 0 ? NS - 1 : st - 1); }
      }
#undef MPF_AWORD32
#undef MPF_BFRAG32
    } else {
    v4i bf[2][NT];
      uint32_t aw[2][MT];
  #define MPF_AWORD(s_, i_) sa[(((2 * (s_) + (h >> 1)) >> 2) * TM + wr * 16 * MT + 16 * (i_) + r) * 4 + ((2 * (s_) + (h >> 1)) & 3)]
  #pragma unroll
      for (int i = 0; i < MT; i++) aw[0][i] = MPF_AWORD(0, i);
  #pragma unroll
      for (int j = 0; j < NT; j++)
        bf[0][j] = *reinterpret_cast<const v4i *>(sb + (size_t)(wc * NT + j) * 1024 + (size_t)lane * 16);
  #pragma unroll
      for (int s = 0; s < KS; s++) {
        if (s + 1 < KS) {
  #pragma unroll
          for (int i = 0; i < MT; i++) aw[(s + 1) & 1][i] = MPF_AWORD(s + 1, i);
  #pragma unroll
          for (int j = 0; j < NT; j++) {
            if constexpr (EXPR == 2 || EXPR == 4) bf[(s + 1) & 1][j] = bf[s & 1][j];
            else bf[(s + 1) & 1][j] = *reinterpret_cast<const v4i *>(sb + (size_t)(s + 1) * BT + (size_t)(wc * NT + j) * 1024 + (size_t)lane * 16);
          }
        }
  #pragma unroll
        for (int i = 0; i < MT; i++) {
          const uint32_t bits = (aw[s & 1][i] >> ((h & 1) * 16)) & 0xFFFFu;
          v4i af;
          if constexpr (EXPR == 1 || EXPR == 4) { af.x = af.y = af.z = af.w = (int)aw[s & 1][i]; } else {
          af.x = (int)((((bits)&0xFu) * 0x00204081u) & 0x01010101u);
          af.y = (int)((((bits >> 4) & 0xFu) * 0x00204081u) & 0x01010101u);
          af.z = (int)((((bits >> 8) & 0xFu) * 0x00204081u) & 0x01010101u);
          af.w = (int)((((bits >> 12) & 0xFu) * 0x00204081u) & 0x01010101u);
          }
  #pragma unroll
          for (int j = 0; j < NT; j++) acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af, bf[s & 1][j], acc[i][j], 0, 0, 0);
        }
        if constexpr (EXPR != 3 && EXPR != 4) { if (s == KS / 2 - 1 && late) MPF_GLOAD(kb + (NS - 1) * KS, st == 0 ? NS - 1 : st - 1); }
      }
    }
#undef MPF_AWORD
    MPF_STAGE_WAIT();
    // a raw barrier: __syncthreads() would drain the LDS-DMA requests of the younger stages (they count as pending LDS
    // writes on the vector-memory counter) and with them the whole point of the ring
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    st = st + 1 == NS ? 0 : st + 1;
  }
#undef MPF_GLOAD
#undef MPF_GLOAD_KB
#undef MPF_STAGE_WAIT

  const int row0 = rb * TM + wr * 16 * MT, col0 = cb * TN + wc * 16 * NT;
  if constexpr (M32) {
    // D layout of the 32x32 forms: column = lane & 31, row = 8 (reg / 4) + 4 (lane >> 5) + reg % 4
#pragma unroll
    for (int i = 0; i < MT2; i++)
#pragma unroll
      for (int j = 0; j < NT2; j++) {
        const int col = col0 + 32 * j + c32;
#pragma unroll
        for (int q = 0; q < 16; q++) {
          const int row = row0 + 32 * i + 8 * (q >> 2) + 4 * hh + (q & 3);
          int32_t *p = C + (size_t)row * Bp + col;
          const int v = acc32[i][j][q] * mult;
          if (atomic) { if (v) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
          else *p = v;
        }
      }
    return;
  }
  // D layout (all 16x16 MFMA forms on gfx950): column = lane & 15, row = 4 (lane >> 4) + reg
#pragma unroll
  for (int i = 0; i < MT; i++)
#pragma unroll
    for (int j = 0; j < NT; j++) {
      const int col = col0 + 16 * j + r;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int row = row0 + 16 * i + 4 * h + q;
        int32_t *p = C + (size_t)row * Bp + col;
        const int v = acc[i][j][q] * mult;
        if (atomic) { if (v) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        else *p = v;
      }
    }
}
template <int MT, int NT, int WM, int WN, int KS, int NS>
constexpr size_t gemm_lds() { return (size_t)NS * KS * (16 * NT * WN) * 64 + (size_t)NS * (KS / 2) * (16 * MT * WM) * 16; }

// R_T[b] = sum over rows of C[row][b]   (rt zeroed by the launcher; 64 rows per thread, integer atomics)
__global__ __launch_bounds__(256) void k_colsum(const int32_t *__restrict__ C, int rows, int Bp, int32_t *__restrict__ rt)
{
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= Bp) return;
  const int r0 = blockIdx.y * 64, r1 = min(rows, r0 + 64);
  int32_t s = 0;
  for (int r = r0; r < r1; r++) s += C[(size_t)r * Bp + b];
  if (s) __hip_atomic_fetch_add(rt + b, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// R_T += C[row] - C[home]   (the accepted move's candidate becomes the current tree)
__global__ __launch_bounds__(256) void k_rt_update(int32_t *__restrict__ rt, const int32_t *__restrict__ C, int Bp,
                                                   uint32_t row, uint32_t home)
{
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= Bp) return;
  rt[b] += C[(size_t)row * Bp + b] - C[(size_t)home * Bp + b];
}

// ------------------------------------------------------------------------------------------------ events
// Index space: the scan's output indices (candidates in the reference's order, each part followed by its home
// slot).  info[i] = (mask row, part) for a candidate, (.., 0xFFFFFFFF) for a home slot; a candidate takes part in
// the bookkeeping iff cost[i] < thr[part] (the logl_cutoff filter, iqtree.cpp:3343; thr = largest admissible cost + 1).
// score(i, b) = R_T[b] - C[home(part)][b] + C[row(i)][b]   (parsimony length of candidate i under sample b)
// crow (optional): scan output index -> row of C when only the saved candidates were multiplied
__device__ __forceinline__ bool ufb_score(uint32_t i, int b, const uint2 *__restrict__ info, const uint32_t *__restrict__ cost,
                                          const uint32_t *__restrict__ thr, const uint32_t *__restrict__ home,
                                          const uint32_t *__restrict__ crow, const int32_t *__restrict__ C, int Bp, int32_t rt,
                                          int32_t &s)
{
  const uint2 in = info[i];
  if (in.y == 0xFFFFFFFFu) return false;
  if (in.y == 0xFFFFFFFEu) { s = rt; return true; }      // the current tree itself (booked once per prune node): R_T
  if (cost[i] >= thr[in.y]) return false;
  const uint32_t hi = home[in.y];
  const uint32_t rc = crow ? crow[i] : in.x, rh = crow ? crow[hi] : hi;
  s = rt - C[(size_t)rh * Bp + b] + C[(size_t)rc * Bp + b];
  return true;
}

constexpr int kUfbChunk = 256;

__global__ __launch_bounds__(256) void k_ufb_chunkmin(const uint2 *__restrict__ info, const uint32_t *__restrict__ cost,
                                                      const uint32_t *__restrict__ thr, const uint32_t *__restrict__ home,
                                                      const uint32_t *__restrict__ crow, const int32_t *__restrict__ C, int Bp,
                                                      const int32_t *__restrict__ rt, uint32_t n_idx, uint32_t *__restrict__ cmin)
{
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t chunk = blockIdx.y;
  if (b >= Bp) return;
  const uint32_t i0 = chunk * kUfbChunk, i1 = min(n_idx, i0 + kUfbChunk);
  const int32_t r = rt[b];
  uint32_t m = 0xFFFFFFFFu;
  for (uint32_t i = i0; i < i1; i++) {
    int32_t s;
    if (ufb_score(i, b, info, cost, thr, home, crow, C, Bp, r, s)) m = min(m, (uint32_t)s);
  }
  cmin[(size_t)chunk * Bp + b] = m;
}

// pre[chunk][b] = min(best[b], cmin[0..chunk)[b])
__global__ __launch_bounds__(256) void k_ufb_prefix(const uint32_t *__restrict__ cmin, const uint32_t *__restrict__ best, int Bp,
                                                    uint32_t n_chunks, uint32_t *__restrict__ pre)
{
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= Bp) return;
  uint32_t run = best[b];
  for (uint32_t c = 0; c < n_chunks; c++) {
    pre[(size_t)c * Bp + b] = run;
    run = min(run, cmin[(size_t)c * Bp + b]);
  }
}

// every (candidate, sample) whose score reaches the running minimum in scan order: the only pairs for which the
// reference's update rule can fire (rell > boot_logl - epsilon with 0 < epsilon < 1 and integer scores)
__global__ __launch_bounds__(256) void k_ufb_events(const uint2 *__restrict__ info, const uint32_t *__restrict__ cost,
                                                    const uint32_t *__restrict__ thr, const uint32_t *__restrict__ home,
                                                    const uint32_t *__restrict__ crow, const int32_t *__restrict__ C, int Bp, int B,
                                                    const int32_t *__restrict__ rt, uint32_t n_idx, const uint32_t *__restrict__ pre,
                                                    UfbEvent *__restrict__ ev, uint32_t ev_cap, uint32_t *__restrict__ ev_count,
                                                    const uint32_t *__restrict__ fixed)
{
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t chunk = blockIdx.y;
  if (b >= B) return;
  const uint32_t i0 = chunk * kUfbChunk, i1 = min(n_idx, i0 + kUfbChunk);
  const int32_t r = rt[b];
  // fixed (the top-N rules): the bound is the sample's threshold at the start of the batch and does not follow the scores --
  // a list's threshold moves by the host's rule, not to the score that beat it
  uint32_t run = fixed ? fixed[b] : pre[(size_t)chunk * Bp + b];
  for (uint32_t i = i0; i < i1; i++) {
    int32_t s;
    if (!ufb_score(i, b, info, cost, thr, home, crow, C, Bp, r, s)) continue;
    if ((uint32_t)s <= run) {
      const uint32_t at = atomicAdd(ev_count, 1u);
      if (at < ev_cap) ev[at] = UfbEvent{i, (uint32_t)b, (uint32_t)s};
      if (!fixed) run = (uint32_t)s;
    }
  }
}

// The batches inside a climb (a few hundred output indices: two to four chunks): the same events in ONE launch -- every chunk's
// threads first run the minimum over the chunks in front of theirs themselves (at most three chunks of re-reads from L2) instead
// of waiting for two more dispatches (chunk minima, prefix), each of which costs more than the re-reads.
__global__ __launch_bounds__(256) void k_ufb_events_fused(const uint2 *__restrict__ info, const uint32_t *__restrict__ cost,
                                                          const uint32_t *__restrict__ thr, const uint32_t *__restrict__ home,
                                                          const uint32_t *__restrict__ crow, const int32_t *__restrict__ C, int Bp, int B,
                                                          const int32_t *__restrict__ rt, uint32_t n_idx, const uint32_t *__restrict__ best,
                                                          UfbEvent *__restrict__ ev, uint32_t ev_cap, uint32_t *__restrict__ ev_count)
{
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t chunk = blockIdx.y;
  if (b >= B) return;
  const uint32_t i0 = chunk * kUfbChunk, i1 = min(n_idx, i0 + kUfbChunk);
  const int32_t r = rt[b];
  uint32_t run = best[b];
  for (uint32_t i = 0; i < i0; i++) {
    int32_t s;
    if (ufb_score(i, b, info, cost, thr, home, crow, C, Bp, r, s)) run = min(run, (uint32_t)s);
  }
  for (uint32_t i = i0; i < i1; i++) {
    int32_t s;
    if (!ufb_score(i, b, info, cost, thr, home, crow, C, Bp, r, s)) continue;
    if ((uint32_t)s <= run) {
      const uint32_t at = atomicAdd(ev_count, 1u);
      if (at < ev_cap) ev[at] = UfbEvent{i, (uint32_t)b, (uint32_t)s};
      run = (uint32_t)s;
    }
  }
}

// ---- the batches inside a climb: ONE extraction launch that also hands the results to the host --------------------------------
// k_ufb_events gives every sample one thread that walks a chunk of 256 indices alone: four to eight workgroups, each thread a
// chain of 256 dependent-looking iterations and one same-address atomic per event -- 34-82 us for the few hundred candidates of a
// climb's batch, more than the product in front of it.  Here a workgroup owns 64 samples (one per lane: the rows of C are read
// as 256-byte runs) and its 16 waves share the indices: a wave scores kUfbSlice consecutive indices into registers, the
// slices' minima meet in LDS, every wave takes the running minimum of the slices in front of its own from there and emits its
// events with ONE atomic per wave (the order of events in the buffer is free: the host sorts them).
// The workgroup that finishes last publishes to the host's pinned buffers without a copy dispatch: the first h_ev_cap events,
// up to three word ranges (the scan's costs and the refresh's mutation counts, info, R_T), the event count and a flag word.
constexpr int kUfbSlice = 16;
struct UfbPublish {
  const uint32_t *src[3];
  uint32_t *dst[3];
  uint32_t words[3];
  UfbEvent *h_ev;
  uint32_t h_ev_cap;
  uint32_t *h_flag;                                // [0] = number of events, [1] = 1 when everything above has arrived
  uint32_t *done;                                  // zeroed device word, left zeroed
};

// (one workgroup copying 70 KB with a load -> store loop of a dozen dependent iterations took 19 us: the ranges are copied by ALL
//  workgroups, a slice each, in front of the ticket; k_ufb_events2 writes its events to the host as it emits them)
__device__ __forceinline__ void ufb_publish_ranges(const UfbPublish &pb, uint32_t part, uint32_t n_parts)
{
  for (int k = 0; k < 3; k++) {
    const uint32_t n = pb.words[k], per = (n + n_parts - 1) / n_parts;
    const uint32_t lo = min(n, part * per), hi = min(n, lo + per);
    for (uint32_t i = lo + threadIdx.x; i < hi; i += blockDim.x) pb.dst[k][i] = __builtin_nontemporal_load(pb.src[k] + i);
  }
}

template <bool COPY_EVENTS>
__device__ __forceinline__ void ufb_publish(const UfbPublish &pb, const UfbEvent *__restrict__ ev, const uint32_t *__restrict__ ev_count,
                                            uint32_t n_tickets)
{
  __shared__ int s_last;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");   // every wave: its own words (device and host) have arrived
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t ticket = __hip_atomic_fetch_add(pb.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = ticket == n_tickets - 1;
  }
  __syncthreads();
  if (!s_last) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  const uint32_t n_ev = __hip_atomic_load(ev_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (COPY_EVENTS) {
    const uint32_t n_copy = min(n_ev, pb.h_ev_cap) * 3u;
    const uint32_t *evw = reinterpret_cast<const uint32_t *>(ev);
    uint32_t *hw = reinterpret_cast<uint32_t *>(pb.h_ev);
    for (uint32_t i = threadIdx.x; i < n_copy; i += blockDim.x) hw[i] = __builtin_nontemporal_load(evw + i);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    __hip_atomic_store(pb.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    pb.h_flag[0] = n_ev;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    __hip_atomic_store(pb.h_flag + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

template <bool FIXED>
__global__ __launch_bounds__(1024) void k_ufb_events2(const uint2 *__restrict__ info, const uint32_t *__restrict__ cost,
                                                      const uint32_t *__restrict__ thr, const uint32_t *__restrict__ home,
                                                      const uint32_t *__restrict__ crow, const int32_t *__restrict__ C, int Bp, int B,
                                                      const int32_t *__restrict__ rt, uint32_t n_idx, const uint32_t *__restrict__ best,
                                                      UfbEvent *__restrict__ ev, uint32_t ev_cap, uint32_t *__restrict__ ev_count,
                                                      UfbPublish pb, int clamp_rt, const uint32_t *__restrict__ cut)
{
  if (cut) n_idx = min(n_idx, __builtin_nontemporal_load(cut));       // (the batch's certain end, k_ufb_mid)
  __shared__ uint32_t wmin[16][64];
  __shared__ uint32_t wtot[16];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int b = blockIdx.x * 64 + lane;
  const bool live = b < B;
  const int bb = min(b, Bp - 1);
  const int32_t r = rt[bb];
  uint32_t run0 = best[bb];
  // clamp_rt (a batch launched before the replay of the one in front): best[] is one batch old -- but the current tree
  // was offered to every sample when it was accepted, so no sample's best is worse than R_T, and a bound that is too high only
  // costs events the host ignores
  if (clamp_rt) run0 = min(run0, (uint32_t)r);
  if (pb.h_flag) ufb_publish_ranges(pb, blockIdx.x, gridDim.x);
  for (uint32_t base = 0; base < n_idx; base += 16u * kUfbSlice) {
    const uint32_t i0 = base + (uint32_t)w * kUfbSlice;
    // the slice's metadata by lanes 0..15 (info -> thr / home -> rows: three dependent round trips per pass, not per index),
    // then the 2 x 16 reads of C go out together -- taking an index one at a time behind its own tests made a pass 10 us long
    uint32_t my_rc = 0, my_rh = 0, my_mode = 0;    // mode: 0 = takes no part, 1 = the current tree (score R_T), 2 = candidate
    {
      const uint32_t mi = i0 + (uint32_t)(lane & (kUfbSlice - 1));
      if (mi < n_idx) {
        const uint2 in = info[mi];
        if (in.y == 0xFFFFFFFEu) my_mode = 1;
        else if (in.y != 0xFFFFFFFFu) {
          const uint32_t hm = home[in.y];
          if (cost[mi] < thr[in.y]) {
            my_mode = 2;
            my_rc = crow ? crow[mi] : in.x;
            my_rh = crow ? crow[hm] : hm;
          }
        }
      }
    }
    uint32_t sc[kUfbSlice];
    int32_t ch[kUfbSlice], cc[kUfbSlice];
#pragma unroll
    for (int k = 0; k < kUfbSlice; k++) {
      const uint32_t rh = (uint32_t)__builtin_amdgcn_readlane((int)my_rh, k), rc = (uint32_t)__builtin_amdgcn_readlane((int)my_rc, k);
      ch[k] = C[(size_t)rh * Bp + bb];
      cc[k] = C[(size_t)rc * Bp + bb];
    }
    uint32_t m = 0xFFFFFFFFu;
#pragma unroll
    for (int k = 0; k < kUfbSlice; k++) {
      const uint32_t mode = (uint32_t)__builtin_amdgcn_readlane((int)my_mode, k);
      sc[k] = mode == 0 ? 0xFFFFFFFFu : mode == 1 ? (uint32_t)r : (uint32_t)(r - ch[k] + cc[k]);
      m = min(m, sc[k]);
    }
    wmin[w][lane] = m;
    __syncthreads();
    uint32_t run = run0, tot = run0;
    for (int w2 = 0; w2 < 16; w2++) {
      const uint32_t v = wmin[w2][lane];
      if (!FIXED && w2 < w) run = min(run, v);
      tot = min(tot, v);
    }
    // this lane's events of the slice (a fixed bound does not follow the scores)
    uint32_t hit = 0, cur = run;
#pragma unroll
    for (int k = 0; k < kUfbSlice; k++)
      if (live && sc[k] != 0xFFFFFFFFu && sc[k] <= cur) { hit |= 1u << k; if (!FIXED) cur = sc[k]; }
    // one reservation per WORKGROUP and pass (same-address atomics are served one after the other, ~70 ns each: one per wave
    // made this kernel 25 us long): the waves' totals meet in LDS, wave 0 reserves, every wave starts behind the waves in front
    // of it.  Inside a wave the events lie in (index, sample) order -- ballots, no lane scan.
    uint32_t wave_total = 0;
    uint32_t pos[kUfbSlice];
#pragma unroll
    for (int k = 0; k < kUfbSlice; k++) {
      const unsigned long long bal = __ballot((hit >> k) & 1u);
      pos[k] = wave_total + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
      wave_total += (uint32_t)__popcll(bal);
    }
    if (lane == 0) wtot[w] = wave_total;
    __syncthreads();
    if (w == 0) {
      uint32_t t = lane < 16 ? wtot[lane] : 0u, incl = t;
#pragma unroll
      for (int d = 1; d < 16; d <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, d, 64);
        if (lane >= d) incl += up;
      }
      const uint32_t all = (uint32_t)__shfl((int)incl, 15, 64);
      uint32_t base0 = 0;
      if (lane == 0 && all) base0 = atomicAdd(ev_count, all);
      base0 = (uint32_t)__shfl((int)base0, 0, 64);
      if (lane < 16) wtot[lane] = base0 + incl - t;
    }
    __syncthreads();
    const uint32_t at0 = wtot[w];
#pragma unroll
    for (int k = 0; k < kUfbSlice; k++)
      if (hit & (1u << k)) {
        const uint32_t at = at0 + pos[k];
        const UfbEvent e{i0 + (uint32_t)k, (uint32_t)b, sc[k]};
        if (at < ev_cap) ev[at] = e;
        if (at < pb.h_ev_cap) pb.h_ev[at] = e;       // (h_ev_cap = 0 without a host to publish to)
      }
    if (!FIXED) run0 = tot;
    __syncthreads();
  }
  if (pb.h_flag) ufb_publish<false>(pb, ev, ev_count, gridDim.x);
}

// the publishing tail alone, behind the chunked kernels of a large batch
__global__ __launch_bounds__(1024) void k_ufb_publish(const UfbEvent *__restrict__ ev, const uint32_t *__restrict__ ev_count, UfbPublish pb)
{
  ufb_publish_ranges(pb, blockIdx.x, gridDim.x);
  ufb_publish<true>(pb, ev, ev_count, gridDim.x);
}

// k_ufb_prep + the publication of the SCAN's results (costs with the refresh's mutation counts, info) behind a flag of their own:
// the host can take the search's decision from the costs while the product and the extraction are still running
__global__ __launch_bounds__(256) void k_ufb_mid(uint4 *__restrict__ C4, uint32_t n4, uint2 *__restrict__ info, const uint32_t *__restrict__ idx,
                                                 uint32_t n_self, uint32_t code, uint32_t *__restrict__ ev_count, UfbPublish pb,
                                                 const uint32_t *__restrict__ cost, const uint32_t *__restrict__ home,
                                                 const uint32_t *__restrict__ plan_end, uint32_t n_idx, uint32_t *__restrict__ cut)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n4) C4[i] = make_uint4(0u, 0u, 0u, 0u);
  if (i == 0) *ev_count = 0u;
  if (blockIdx.x == 0) {
    // the current tree's slots first (what an earlier batch left in them must not be taken for a candidate below)
    for (uint32_t k = threadIdx.x; k < n_self; k += blockDim.x) info[idx[k]] = make_uint2(0u, code);
    // the batch's certain end: a candidate is strictly better than the current tree <=> its cost is below the cost of its part's
    // home edge (both are the join onto an edge of the tree without the pruned subtree; the rest of the length is common), and
    // the first prune node that has one ends the batch at the latest.  cut = end of that prune node's index range: the
    // product's row blocks and the extraction stop there.
    __shared__ uint32_t s_cut;
    if (threadIdx.x == 0) s_cut = n_idx;
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < n_idx; k += blockDim.x) {
      const uint2 in = info[k];
      if (in.y >= 0xFFFFFFFEu) continue;
      if (cost[k] < cost[home[in.y]]) atomicMin(&s_cut, plan_end[in.y]);
    }
    __syncthreads();
    if (threadIdx.x == 0) { *cut = s_cut; pb.h_flag[2] = s_cut; }
  }
  const uint32_t pub = min(gridDim.x, 16u);        // publishing workgroups (a ticket per workgroup of a 400-workgroup launch would take longer than the copy)
  if (blockIdx.x < pub) {
    ufb_publish_ranges(pb, blockIdx.x, pub);
    ufb_publish<false>(pb, nullptr, ev_count, pub);
  }
}

// in front of the product of a climb's batch, one launch: C <- 0 (the K-split product adds into it), the event counter <- 0,
// and the current tree's slots info[idx[i]] = (0, code) as k_ufb_self writes them
__global__ __launch_bounds__(256) void k_ufb_prep(uint4 *__restrict__ C4, uint32_t n4, uint2 *__restrict__ info, const uint32_t *__restrict__ idx,
                                                  uint32_t n_self, uint32_t code, uint32_t *__restrict__ ev_count)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n4) C4[i] = make_uint4(0u, 0u, 0u, 0u);
  if (i < n_self) info[idx[i]] = make_uint2(0u, code);
  if (i == 0) *ev_count = 0u;
}

// info[idx[i]] = (0, code): the slots reserved for the current tree in front of every prune node's candidates
// (code 0xFFFFFFFE = takes part with score R_T, 0xFFFFFFFF = does not: the current tree fails the cut-off)
__global__ __launch_bounds__(256) void k_ufb_self(uint2 *__restrict__ info, const uint32_t *__restrict__ idx, uint32_t n, uint32_t code)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) info[idx[i]] = make_uint2(0u, code);
}

// ------------------------------------------------------------------------------------------------ weight layout
// The product's right-hand side holds, for every 32-site word pair, the weights of the samples at the FIRST expanded
// site of each pattern (the site pllComputePatternParsimony reads, sprparsimony.cpp:3382-3387).  The site of a pattern
// moves whenever the alignment is re-weighted (ratchet climbs re-pack the sites), so the layout is a device pass over
// the samples as uploaded: src[col][ptn] uint16, col = local sample (or the extra column of the original pattern
// frequencies), first[ptn] = first expanded site of the pattern in the packing in force (-1: not packed), cur[ptn] = its
// current weight (0: it owns no site).  Wt is zeroed by the caller.
__global__ __launch_bounds__(256) void k_ufb_layout(const uint16_t *__restrict__ src, int n_cols, int P, const int32_t *__restrict__ first,
                                                    const int32_t *__restrict__ cur, uint8_t *__restrict__ Wt, int Bp, int planes,
                                                    size_t plane_bytes)
{
  // a workgroup = 16 patterns x 16 columns: the 16 x 16 bytes of a column group's 16 consecutive sites are ONE 256-byte piece of Wt
  // (16 bytes per column), so a wave -- 4 columns x 16 patterns -- writes 64 consecutive bytes.  (A thread per pattern and a grid row per
  // column, as before round 6's end, left every wave writing four 16-byte pieces 256 bytes apart, a byte per lane: 0.6 ms per call at C4.)
  const int ptn = blockIdx.x * 16 + (threadIdx.x & 15);
  const int col = blockIdx.y * 16 + (threadIdx.x >> 4);
  if (ptn >= P || col >= n_cols) return;
  const uint32_t w = src[(size_t)col * P + ptn];
  const int site = first[ptn];
  if (!w || site < 0 || cur[ptn] <= 0) return;
  const int word = site >> 5, bit = site & 31;
  const int kb = word >> 1, within = (word & 1) * 32 + bit, h = within >> 4, j = within & 15;
  const size_t off = (size_t)kb * ((size_t)(Bp / 16) * 1024) + (size_t)h * 256 + (size_t)j + (size_t)(col >> 4) * 1024 + (size_t)(col & 15) * 16;
  for (int pl = 0; pl < planes; pl++) Wt[(size_t)pl * plane_bytes + off] = (uint8_t)((w >> (7 * pl)) & 0x7Fu);
}

// out[i] = C[i][col]: one column of the product, contiguous (the original-frequency column of a re-weighted climb)
__global__ __launch_bounds__(256) void k_ufb_column(const int32_t *__restrict__ C, int Bp, int col, uint32_t rows, int32_t *__restrict__ out)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < rows) out[i] = C[(size_t)i * Bp + col];
}

// ------------------------------------------------------------------------------------------------ weighted engine: bit planes
// vals[row][npat] (16-bit per-pattern lengths of a tentative tree, from k_snk_scan) -> planes[j][row][Wp] with bit i of word w
// of plane j = bit j of vals[row][32 w + i]: REPS = sum_j 2^j * (plane j x weights), one k_bitgemm per plane.
// One wave per (row, 64 patterns); rows_p / Wp are the padded sizes of the product (the padding is zeroed by the caller).
__global__ __launch_bounds__(256) void k_vals_planes(const uint16_t *__restrict__ vals, uint32_t rows, uint32_t npat, int K,
                                                     uint32_t *__restrict__ planes, uint32_t rows_p, uint32_t Wp)
{
  const int lane = threadIdx.x & 63;
  const uint32_t tile = blockIdx.x * 4 + (threadIdx.x >> 6), row = blockIdx.y;
  if (row >= rows || tile * 64 >= npat) return;    // wave-uniform
  const uint32_t j = tile * 64 + (uint32_t)lane;
  const uint32_t v = j < npat ? vals[(size_t)row * npat + j] : 0u;
  for (int k = 0; k < K; k++) {
    const unsigned long long bal = __ballot((int)((v >> k) & 1u));
    if (lane == 0)
      *reinterpret_cast<uint2 *>(planes + ((size_t)k * rows_p + row) * Wp + 2 * tile) = make_uint2((uint32_t)bal, (uint32_t)(bal >> 32));
  }
}

hipError_t launch_vals_planes(hipStream_t st, const uint16_t *vals, uint32_t rows, uint32_t npat, int K, uint32_t *planes, uint32_t rows_p,
                              uint32_t Wp)
{
  if (!rows || K <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_vals_planes, dim3((npat + 255) / 256, rows), dim3(256), 0, st, vals, rows, npat, K, planes, rows_p, Wp);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ launchers
hipError_t launch_join_masks(hipStream_t st, const Geometry &g, const uint32_t *vec, const EvOp *ops, int n_ops, uint32_t *masks)
{
  if (n_ops <= 0) return hipSuccess;
  const int S = g.S;
  hipLaunchKernelGGL(k_join_masks, dim3((g.Wp + 255) / 256, n_ops), dim3(256), 0, st, vec, ops, n_ops, S, g.Wp, masks);
  return hipGetLastError();
}

template <int MT, int NT, int WM, int WN, int KS, int NS, int EXPR = 0, bool M32 = false>
static hipError_t launch_bitgemm_t(hipStream_t st, const uint32_t *masks, int rows_padded, int Wp, const uint8_t *Wt, int Bp, int32_t *C,
                                   int mult, int accumulate, const uint32_t *rowsel, const uint32_t *row_limit, long want_default = 192)
{
  constexpr int TM = 16 * MT * WM;
  const int row_blocks = rows_padded / TM, col_blocks = Bp / (16 * NT * WN);
  const int nkb = Wp / 2;
  // small batches: split K so that the launch still has a few workgroups per CU
  long tiles = (long)row_blocks * col_blocks;
  // (measured on the batches of a C3 climb with 1000 samples: 512 workgroups 0.50 s of product kernels per climb, 192: 0.30 s
  //  -- every K-split multiplies the atomic adds into C, and 192 workgroups already stream the weight matrix at full rate)
  // (128-row tiles, the batches inside a climb: 256 workgroups 0.234 s, 192: 0.261 s, 512: 0.233 s)
  static const long want_env = std::getenv("MPF_GEMM_WANT") ? std::atol(std::getenv("MPF_GEMM_WANT")) : 0;
  const long want = want_env > 0 ? want_env : want_default;
  int ksplit = 1;
  if (tiles < want) ksplit = (int)std::min<long>((want + tiles - 1) / tiles, std::max(1, nkb / 16));
  int per = (nkb + ksplit - 1) / ksplit;
  per = (per + KS - 1) / KS * KS;                        // stages start on multiples of the stage depth
  ksplit = (nkb + per - 1) / per;
  const int atomic = (ksplit > 1 || accumulate) ? 1 : 0;
  if (atomic && !accumulate) {
    hipError_t e = hipMemsetAsync(C, 0, (size_t)rows_padded * (size_t)Bp * sizeof(int32_t), st);
    if (e != hipSuccess) return e;
  }
  // grid.x covers the XCD-aware (row block, column block) map of the kernel
  unsigned gx;
  if (col_blocks <= 8 && 8 % col_blocks == 0) { const int per_x = 8 / col_blocks; gx = (unsigned)(((row_blocks + per_x - 1) / per_x) * 8); }
  else gx = (unsigned)(row_blocks * col_blocks);
  // > 64 KiB of dynamic LDS needs the opt-in, once per device (engines of one process may sit on different GPUs)
  static std::atomic<bool> attr_set[64];                  // (engines on several host threads share a device)
  int dev = 0;
  (void)hipGetDevice(&dev);
  constexpr size_t lds = gemm_lds<MT, NT, WM, WN, KS, NS>();
  if (dev < 0 || dev >= 64 || !attr_set[dev].load(std::memory_order_acquire)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_bitgemm<MT, NT, WM, WN, KS, NS, EXPR, M32>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64) attr_set[dev].store(true, std::memory_order_release);
  }
  hipLaunchKernelGGL((k_bitgemm<MT, NT, WM, WN, KS, NS, EXPR, M32>), dim3(gx, (unsigned)ksplit), dim3(64 * WM * WN), lds, st, masks, Wp, Wt, Bp, C, mult, atomic, row_blocks, per, rowsel, row_limit);
  return hipGetLastError();
}

hipError_t launch_bitgemm(hipStream_t st, const uint32_t *masks, int rows_padded, int Wp, const uint8_t *Wt, int Bp, int32_t *C,
                          int mult, int accumulate, const uint32_t *rowsel, const uint32_t *row_limit)
{
  if (rows_padded <= 0) return hipSuccess;
  // 256-sample column blocks when the (padded) sample count allows, else 128-sample blocks with twice the rows per
  // workgroup (sample-sharded runs: 1000 samples over 8 GPUs = 125 per rank)
  static int variant = -1;
  if (variant < 0) { const char *v = std::getenv("MPF_GEMM_VARIANT"); variant = v ? std::atoi(v) : 0; }
  if (Bp % 256 == 0) {
    // 8 x 1 waves of 32 rows x 256 samples: an expanded A fragment feeds 16 MFMAs (0.8 vector instructions per MFMA)
    // few rows (the batches inside a climb): 512-row x 128-sample tiles -- twice the column blocks, half the K-splits
    // (C3 climb from a random tree, 1000 samples, product kernels in total: 512 x 128 tiles 0.298 s, 256 x 128 0.253 s, 128 x 128
    //  0.234 s, 128 x 256 0.264 s -- fewer K-splits per output element, i.e. fewer atomic adds into C, and all CUs busy)
    static const int small_v = std::getenv("MPF_GEMM_SMALL") ? std::atoi(std::getenv("MPF_GEMM_SMALL")) : 2;
    if (small_v == 1 && rows_padded <= 1024) return launch_bitgemm_t<4, 8, 8, 1, 4, 2>(st, masks, rows_padded, Wp, Wt, Bp, C, mult, accumulate, rowsel, row_limit);
    if (small_v == 2 && rows_padded <= 2048) return launch_bitgemm_t<1, 8, 8, 1, 4, 2>(st, masks, rows_padded, Wp, Wt, Bp, C, mult, accumulate, rowsel, row_limit, 256);
    if (small_v == 3 && rows_padded <= 2048) return launch_bitgemm_t<2, 8, 8, 1, 4, 2>(st, masks, rows_padded, Wp, Wt, Bp, C, mult, accumulate, rowsel, row_limit);
    if (small_v == 4 && rows_padded <= 2048) return launch_bitgemm_t<1, 16, 8, 1, 4, 2>(st, masks, rows_padded, Wp, Wt, Bp, C, mult, accumulate, rowsel, row_limit);
#ifdef MPF_EXPERIMENTS                     // (knock-out variants of tools/gemm_bounds.sh, wrong results on purpose: `make EXPERIMENTS=1` only)
    static const int expr = std::getenv("MPF_GEMM_EXPERIMENT") ? std::atoi(std::getenv("MPF_GEMM_EXPERIMENT")) : 0;
    if (expr == 1) return launch_bitgemm_t<2, 16, 8, 1, 4, 2, 1>(st, masks, rows_padded, Wp, Wt, Bp, C, mult, accumulate, rowsel, row_limit);
    if (expr == 2) return launch_bitgemm_t<2, 16, 8, 1, 4, 2, 2>(st, masks, rows_padded, Wp, Wt, Bp, C, mult, accumulate, rowsel, row_limit);
    if (expr == 3) return launch_bitgemm_t<2, 16, 8, 1, 4, 2, 3>(st, masks, rows_padded, Wp, Wt, Bp, C, mult, accumulate, rowsel, row_limit);
    if (expr == 6) return launch_bitgemm_t<2, 16, 8, 1, 4, 2, 6>(st, masks, rows_padded, Wp, Wt, Bp, C, mult, accumulate, rowsel, row_limit);
    if (expr == 4) return launch_bitgemm_t<2, 16, 8, 1, 4, 2, 4>(st, masks, rows_padded, Wp, Wt, Bp, C, mult, accumulate, rowsel, row_limit);
#endif
    // variants (all within 0.43-0.49 of the nominal peak, tools/gemm_bounds.sh): 0 = two k-blocks per stage, four stages in
    // the ring; 1 / 3 = 4 x 2 waves of 64 rows x 128 samples; default = 8 x 1 waves of 32 rows x 256 samples, four k-blocks
    // per stage, two stages
    if (variant == 4) return launch_bitgemm_t<2, 16, 8, 1, 4, 2, 0, true>(st, masks, rows_padded, Wp, Wt, Bp, C, mult, accumulate, rowsel, row_limit);
    if (variant == 5) return launch_bitgemm_t<4, 8, 4, 2, 4, 2, 0, true>(st, masks, rows_padded, Wp, Wt, Bp, C, mult, accumulate, rowsel, row_limit);
    if (variant == 1) return launch_bitgemm_t<2, 16, 8, 1, 2, 4>(st, masks, rows_padded, Wp, Wt, Bp, C, mult, accumulate, rowsel, row_limit);
    if (variant == 2) return launch_bitgemm_t<4, 8, 4, 2, 2, 4>(st, masks, rows_padded, Wp, Wt, Bp, C, mult, accumulate, rowsel, row_limit);
    if (variant == 3) return launch_bitgemm_t<4, 8, 4, 2, 4, 2>(st, masks, rows_padded, Wp, Wt, Bp, C, mult, accumulate, rowsel, row_limit);
    return launch_bitgemm_t<2, 16, 8, 1, 4, 2>(st, masks, rows_padded, Wp, Wt, Bp, C, mult, accumulate, rowsel, row_limit);
  }
  return launch_bitgemm_t<4, 8, 8, 1, 4, 2>(st, masks, rows_padded, Wp, Wt, Bp, C, mult, accumulate, rowsel, row_limit);
}

int ufb_row_padding(int rows, int Bp)
{
  static const int small_v = std::getenv("MPF_GEMM_SMALL") ? std::atoi(std::getenv("MPF_GEMM_SMALL")) : 2;
  const int r128 = (std::max(rows, 1) + 127) / 128 * 128;
  return (Bp % 256 == 0 && small_v == 2 && r128 <= 2048) ? 128 : kUfbRowTile;     // (launch_bitgemm's 128-row variant)
}

hipError_t launch_colsum(hipStream_t st, const int32_t *C, int rows, int Bp, int32_t *rt)
{
  hipError_t e = hipMemsetAsync(rt, 0, (size_t)Bp * sizeof(int32_t), st);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_colsum, dim3((Bp + 255) / 256, (rows + 63) / 64), dim3(256), 0, st, C, rows, Bp, rt);
  return hipGetLastError();
}

hipError_t launch_rt_update(hipStream_t st, int32_t *rt, const int32_t *C, int Bp, uint32_t row, uint32_t home)
{
  hipLaunchKernelGGL(k_rt_update, dim3((Bp + 255) / 256), dim3(256), 0, st, rt, C, Bp, row, home);
  return hipGetLastError();
}

uint32_t ufb_chunks(uint32_t n_idx) { return (n_idx + kUfbChunk - 1) / kUfbChunk; }

hipError_t launch_ufb_events(hipStream_t st, const uint2 *info, const uint32_t *cost, const uint32_t *thr, const uint32_t *home,
                             const uint32_t *crow, const int32_t *C, int Bp, int B, const int32_t *rt, const uint32_t *best, uint32_t n_idx,
                             uint32_t *cmin, uint32_t *pre, UfbEvent *ev, uint32_t ev_cap, uint32_t *ev_count, int fixed_bound)
{
  if (n_idx == 0) return hipSuccess;
  const uint32_t nc = ufb_chunks(n_idx);
  dim3 grid((Bp + 255) / 256, nc), block(256);
  if (!fixed_bound && nc > 1 && nc <= 4) {
    hipLaunchKernelGGL(k_ufb_events_fused, grid, block, 0, st, info, cost, thr, home, crow, C, Bp, B, rt, n_idx, best, ev, ev_cap, ev_count);
    return hipGetLastError();
  }
  // a single chunk: its running minimum starts at best[] itself, no prefix pass
  const bool prefix = !fixed_bound && nc > 1;
  if (prefix) {
    hipLaunchKernelGGL(k_ufb_chunkmin, grid, block, 0, st, info, cost, thr, home, crow, C, Bp, rt, n_idx, cmin);
    hipLaunchKernelGGL(k_ufb_prefix, dim3((Bp + 255) / 256), block, 0, st, cmin, best, Bp, nc, pre);
  }
  hipLaunchKernelGGL(k_ufb_events, grid, block, 0, st, info, cost, thr, home, crow, C, Bp, B, rt, n_idx, prefix ? pre : best, ev, ev_cap, ev_count,
                     fixed_bound ? best : (const uint32_t *)nullptr);
  return hipGetLastError();
}

hipError_t launch_ufb_prep(hipStream_t st, int32_t *C, size_t c_words, uint2 *info, const uint32_t *self_idx, uint32_t n_self, uint32_t code,
                           uint32_t *ev_count)
{
  const uint32_t n4 = (uint32_t)((c_words + 3) / 4);                     // (C is reserved in whole row tiles: a multiple of four words)
  const uint32_t n = std::max(std::max(n4, n_self), 1u);
  hipLaunchKernelGGL(k_ufb_prep, dim3((n + 255) / 256), dim3(256), 0, st, reinterpret_cast<uint4 *>(C), n4, info, self_idx, n_self, code, ev_count);
  return hipGetLastError();
}

static UfbPublish publish_of(const UfbPublishArgs &a)
{
  UfbPublish pb;
  for (int k = 0; k < 3; k++) { pb.src[k] = a.src[k]; pb.dst[k] = a.dst[k]; pb.words[k] = a.words[k]; }
  pb.h_ev = a.h_ev;
  pb.h_ev_cap = a.h_ev_cap;
  pb.h_flag = a.h_flag;
  pb.done = a.done;
  return pb;
}

hipError_t launch_ufb_mid(hipStream_t st, int32_t *C, size_t c_words, uint2 *info, const uint32_t *self_idx, uint32_t n_self, uint32_t code,
                          uint32_t *ev_count, const UfbPublishArgs &a, const uint32_t *cost, const uint32_t *home, const uint32_t *plan_end,
                          uint32_t n_idx, uint32_t *cut)
{
  const uint32_t n4 = (uint32_t)((c_words + 3) / 4);
  const uint32_t n = std::max(n4, 1u);
  hipLaunchKernelGGL(k_ufb_mid, dim3((n + 255) / 256), dim3(256), 0, st, reinterpret_cast<uint4 *>(C), n4, info, self_idx, n_self, code, ev_count, publish_of(a),
                     cost, home, plan_end, n_idx, cut);
  return hipGetLastError();
}

hipError_t launch_ufb_events_publish(hipStream_t st, const uint2 *info, const uint32_t *cost, const uint32_t *thr, const uint32_t *home,
                                     const uint32_t *crow, const int32_t *C, int Bp, int B, const int32_t *rt, const uint32_t *best,
                                     uint32_t n_idx, uint32_t *cmin, uint32_t *pre, UfbEvent *ev, uint32_t ev_cap, uint32_t *ev_count,
                                     int fixed_bound, const UfbPublishArgs &a, int clamp_rt, const uint32_t *cut)
{
  const UfbPublish pb = publish_of(a);
  if (n_idx > 0 && n_idx <= kUfbEvents2Max) {
    dim3 grid((unsigned)((B + 63) / 64)), block(1024);
    if (fixed_bound) hipLaunchKernelGGL(k_ufb_events2<true>, grid, block, 0, st, info, cost, thr, home, crow, C, Bp, B, rt, n_idx, best, ev, ev_cap, ev_count, pb, 0, cut);
    else hipLaunchKernelGGL(k_ufb_events2<false>, grid, block, 0, st, info, cost, thr, home, crow, C, Bp, B, rt, n_idx, best, ev, ev_cap, ev_count, pb, clamp_rt, cut);
    return hipGetLastError();
  }
  hipError_t e = launch_ufb_events(st, info, cost, thr, home, crow, C, Bp, B, rt, best, n_idx, cmin, pre, ev, ev_cap, ev_count, fixed_bound);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_ufb_publish, dim3(16), dim3(1024), 0, st, ev, ev_count, pb);      // (whole sweeps: 1e5 costs, 2e5 info words)
  return hipGetLastError();
}

hipError_t launch_ufb_layout(hipStream_t st, const uint16_t *src, int n_cols, int P, const int32_t *first, const int32_t *cur, uint8_t *Wt,
                             int Bp, int planes, size_t plane_bytes)
{
  hipError_t e = hipMemsetAsync(Wt, 0, plane_bytes * (size_t)planes, st);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_ufb_layout, dim3((P + 15) / 16, (n_cols + 15) / 16), dim3(256), 0, st, src, n_cols, P, first, cur, Wt, Bp, planes, plane_bytes);
  return hipGetLastError();
}

hipError_t launch_ufb_self(hipStream_t st, uint2 *info, const uint32_t *idx, uint32_t n, uint32_t code)
{
  if (!n) return hipSuccess;
  hipLaunchKernelGGL(k_ufb_self, dim3((n + 255) / 256), dim3(256), 0, st, info, idx, n, code);
  return hipGetLastError();
}

hipError_t launch_ufb_column(hipStream_t st, const int32_t *C, int Bp, int col, uint32_t rows, int32_t *out)
{
  if (!rows) return hipSuccess;
  hipLaunchKernelGGL(k_ufb_column, dim3((rows + 255) / 256), dim3(256), 0, st, C, Bp, col, rows, out);
  return hipGetLastError();
}

}  // namespace mpf
