// Load rate of a CU from a small hot buffer (vector L1 hits) by access shape: what the planned scan's child-vector loads cost.
//   shape 0: four buffer_load_dword per "vector" (row-major store: 4 state rows, 64 consecutive words each: 4 x 256 B)
//   shape 1: two buffer_load_dwordx2 (two rows, 128 words)          -- same bytes per vector
//   shape 2: one buffer_load_dwordx4 per vector (word-major store: the four state words of a site word side by side, 1 KB contiguous)
// Build: hipcc --offload-arch=gfx950 -O3 -o l1_rate l1_rate.hip ; run: ./l1_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(64, 8) void k(const uint32_t *buf, uint32_t *out, int iters, uint32_t nvec_mask)
{
  const int lane = threadIdx.x;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)buf, 0, 0x7FFFFFFF, 0x00020000);
  uint32_t acc = 0, v = blockIdx.x * 7u;
  for (int i = 0; i < iters; i++) {
    // two "vectors" of 1 KB per step, like one expansion of the scan; which ones: a cheap LCG, wave-uniform
    v = v * 1664525u + 1013904223u;
    const uint32_t a = (v >> 8) & nvec_mask, b = (v >> 16) & nvec_mask;
    if constexpr (SHAPE == 0) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        acc ^= __builtin_amdgcn_raw_buffer_load_b32(rsrc, (uint32_t)(lane * 4 + r * 256), a * 1024u, 0);
        acc ^= __builtin_amdgcn_raw_buffer_load_b32(rsrc, (uint32_t)(lane * 4 + r * 256), b * 1024u, 0);
      }
    } else if constexpr (SHAPE == 1) {
#pragma unroll
      for (int r = 0; r < 2; r++) {
        const v2u x = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (uint32_t)(lane * 8 + r * 512), a * 1024u, 0);
        const v2u y = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (uint32_t)(lane * 8 + r * 512), b * 1024u, 0);
        acc ^= x[0] ^ x[1] ^ y[0] ^ y[1];
      }
    } else {
      const v4u x = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (uint32_t)(lane * 16), a * 1024u, 0);
      const v4u y = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (uint32_t)(lane * 16), b * 1024u, 0);
      acc ^= x[0] ^ x[1] ^ x[2] ^ x[3] ^ y[0] ^ y[1] ^ y[2] ^ y[3];
    }
  }
  out[blockIdx.x * 64 + lane] = acc;
}

template <int SHAPE>
void run(const char *name, uint32_t *buf, uint32_t *out)
{
  printf("%-58s", name);
  for (uint32_t mask : {1u, 15u, 1023u, 65535u}) {      // 2 KB / 16 KB (L1) / 1 MB (L2) / 64 MB (L2 + MALL) of hot vectors
    const int iters = 4000, blocks = 256 * 32;            // eight waves per SIMD
    hipLaunchKernelGGL(k<SHAPE>, dim3(blocks), dim3(64), 0, 0, buf, out, 10, mask);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<SHAPE>, dim3(blocks), dim3(64), 0, 0, buf, out, iters, mask);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)blocks * iters * 2048.0;
    printf("  %6u KB: %6.1f GB/s/CU", (mask + 1), bytes / (ms * 1e-3) / 1e9 / 256.0);
  }
  printf("\n");
}

int main()
{
  uint32_t *buf, *out;
  hipMalloc(&buf, 65536u * 1024u + 4096); hipMalloc(&out, 256 * 32 * 64 * 4);
  hipMemset(buf, 1, 65536u * 1024u + 4096);
  run<0>("4 x buffer_load_dword per 1-KB vector (row-major)", buf, out);
  run<1>("2 x buffer_load_dwordx2", buf, out);
  run<2>("1 x buffer_load_dwordx4 (word-major)", buf, out);
  printf("(hot set per column: that many KB of vectors, picked at random per step; 64 B/clk/CU at 2.4 GHz = 153.6 GB/s/CU)\n");
  return 0;
}
