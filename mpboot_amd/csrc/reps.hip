// reps.hip -- REPS (resampling parsimony score) contraction, the reference's K10:
//     rell[m][b] = - sum_ptn pattern_pars[m][ptn] * boot_sample[b][ptn]
// (IQTree::saveCurrentTree, reference iqtree.cpp:3411-3449, Vec16us lanes + per-segment horizontal adds).
// Exact 32-bit integer sums here (the reference's 16-bit segment sums are equal whenever they do not wrap,
// SURVEY parity hazard 4); the branch-and-bound skip (:3435-3445) is an optimisation of the CPU loop and is
// not needed.  A dense small-integer contraction: v_dot2_u32_u16 (two u16 MACs per lane per instruction),
// one wave = a 4 x 4 tile of (trees x samples), lanes stride over pattern pairs.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <new>
#include <string>
#include <vector>

#include "../../include/mpfitch.h"

namespace mpf {
void set_error(const std::string &msg);

typedef unsigned short us2 __attribute__((ext_vector_type(2)));
constexpr int kMT = 4, kBT = 4;

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += (uint32_t)__shfl_xor((int)v, off, 64);
  return v;
}

// pars: [M][Pp] u16, boot: [B][Pp] u16, Pp even (zero padded)
__global__ __launch_bounds__(256) void k_reps(const uint16_t *__restrict__ pars, const uint16_t *__restrict__ boot,
                                              int M, int B, int Pp, int32_t *__restrict__ rell)
{
  const int lane = threadIdx.x & 63;
  int gw = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  gw = __builtin_amdgcn_readfirstlane(gw);
  const int bt = (B + kBT - 1) / kBT, mt = (M + kMT - 1) / kMT;
  if (gw >= bt * mt) return;
  const int m0 = (gw / bt) * kMT, b0 = (gw % bt) * kBT;
  uint32_t acc[kMT][kBT];
#pragma unroll
  for (int i = 0; i < kMT; i++)
#pragma unroll
    for (int j = 0; j < kBT; j++) acc[i][j] = 0;
  const int pairs = Pp >> 1;
  for (int p = lane; p < pairs; p += 64) {
    us2 a[kMT], w[kBT];
#pragma unroll
    for (int i = 0; i < kMT; i++) {
      const int m = min(m0 + i, M - 1);
      a[i] = *reinterpret_cast<const us2 *>(pars + (size_t)m * Pp + 2 * p);
    }
#pragma unroll
    for (int j = 0; j < kBT; j++) {
      const int b = min(b0 + j, B - 1);
      w[j] = *reinterpret_cast<const us2 *>(boot + (size_t)b * Pp + 2 * p);
    }
#pragma unroll
    for (int i = 0; i < kMT; i++)
#pragma unroll
      for (int j = 0; j < kBT; j++) acc[i][j] = __builtin_amdgcn_udot2(a[i], w[j], acc[i][j], false);
  }
#pragma unroll
  for (int i = 0; i < kMT; i++)
#pragma unroll
    for (int j = 0; j < kBT; j++) {
      const uint32_t tot = wave_sum_u32(acc[i][j]);
      if (lane == 0 && m0 + i < M && b0 + j < B) rell[(size_t)(m0 + i) * B + (b0 + j)] = -(int32_t)tot;
    }
}

}  // namespace mpf

struct mpf_reps {
  int dev = 0, B = 0, P = 0, Pp = 0;
  uint16_t *d_boot = nullptr, *d_pars = nullptr;
  int32_t *d_rell = nullptr;
  size_t cap_trees = 0;
  hipStream_t st = nullptr;
  std::vector<uint16_t> stage;
};

#define RCHK(expr)                                                                                    \
  do {                                                                                                \
    hipError_t e__ = (expr);                                                                          \
    if (e__ != hipSuccess) { mpf::set_error(std::string(#expr) + ": " + hipGetErrorString(e__)); return MPF_E_HIP; } \
  } while (0)

extern "C" {

int mpf_reps_create(mpf_reps **out, int32_t device, int32_t n_samples, int32_t n_patterns, const uint16_t *boot)
{
  if (!out || !boot || n_samples < 1 || n_patterns < 1) { mpf::set_error("mpf_reps_create: bad argument"); return MPF_E_INVALID; }
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { mpf::set_error("no HIP device available: libmpfitch has no CPU fallback"); return MPF_E_NO_DEVICE; }
  if (device < 0 || device >= ndev) { mpf::set_error("device ordinal out of range"); return MPF_E_INVALID; }
  mpf_reps *r = new (std::nothrow) mpf_reps();
  if (!r) return MPF_E_NOMEM;
  r->dev = device; r->B = n_samples; r->P = n_patterns; r->Pp = (n_patterns + 1) & ~1;
  RCHK(hipSetDevice(device));
  RCHK(hipStreamCreateWithFlags(&r->st, hipStreamNonBlocking));
  RCHK(hipMalloc((void **)&r->d_boot, (size_t)r->B * r->Pp * sizeof(uint16_t)));
  RCHK(hipMemsetAsync(r->d_boot, 0, (size_t)r->B * r->Pp * sizeof(uint16_t), r->st));
  RCHK(hipMemcpy2DAsync(r->d_boot, (size_t)r->Pp * 2, boot, (size_t)r->P * 2, (size_t)r->P * 2, (size_t)r->B, hipMemcpyHostToDevice, r->st));
  RCHK(hipStreamSynchronize(r->st));
  *out = r;
  return MPF_OK;
}

int mpf_reps_scores(mpf_reps *r, int32_t n_trees, const uint16_t *pattern_pars, int32_t *rell)
{
  if (!r || !pattern_pars || !rell || n_trees < 1) { mpf::set_error("mpf_reps_scores: bad argument"); return MPF_E_INVALID; }
  RCHK(hipSetDevice(r->dev));
  if ((size_t)n_trees > r->cap_trees) {
    if (r->d_pars) (void)hipFree(r->d_pars);
    if (r->d_rell) (void)hipFree(r->d_rell);
    r->cap_trees = (size_t)n_trees + 16;
    RCHK(hipMalloc((void **)&r->d_pars, r->cap_trees * r->Pp * sizeof(uint16_t)));
    RCHK(hipMalloc((void **)&r->d_rell, r->cap_trees * r->B * sizeof(int32_t)));
  }
  RCHK(hipMemsetAsync(r->d_pars, 0, (size_t)n_trees * r->Pp * sizeof(uint16_t), r->st));
  RCHK(hipMemcpy2DAsync(r->d_pars, (size_t)r->Pp * 2, pattern_pars, (size_t)r->P * 2, (size_t)r->P * 2, (size_t)n_trees, hipMemcpyHostToDevice, r->st));
  const int bt = (r->B + mpf::kBT - 1) / mpf::kBT, mt = (n_trees + mpf::kMT - 1) / mpf::kMT;
  const long waves = (long)bt * mt;
  hipLaunchKernelGGL(mpf::k_reps, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, r->st, r->d_pars, r->d_boot, n_trees, r->B, r->Pp, r->d_rell);
  RCHK(hipGetLastError());
  RCHK(hipMemcpyAsync(rell, r->d_rell, (size_t)n_trees * r->B * sizeof(int32_t), hipMemcpyDeviceToHost, r->st));
  RCHK(hipStreamSynchronize(r->st));
  return MPF_OK;
}

void mpf_reps_destroy(mpf_reps *r)
{
  if (!r) return;
  if (r->d_boot) (void)hipFree(r->d_boot);
  if (r->d_pars) (void)hipFree(r->d_pars);
  if (r->d_rell) (void)hipFree(r->d_rell);
  if (r->st) (void)hipStreamDestroy(r->st);
  delete r;
}

}  // extern "C"
