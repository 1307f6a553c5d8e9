// Issue rate of the integer VALU instructions the Fitch kernels are made of (gfx950), one to eight waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run: ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define REP16(X) X X X X X X X X X X X X X X X X
template <int KIND>
__global__ __launch_bounds__(1024) void k(uint32_t *out, unsigned long long *cyc, int iters)
{
  uint32_t a0 = threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, b0 = a0 ^ 0x55, b1 = a1 ^ 0x33, b2 = a2 ^ 0x0f, b3 = a3 ^ 0xff;
  uint32_t c0 = 1, c1 = 2, c2 = 3, c3 = 4;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) {
    if constexpr (KIND == 0) {        // VOP2 v_and_b32, 4 independent chains
      REP16(asm volatile("v_and_b32 %0, %0, %4\n v_and_b32 %1, %1, %5\n v_and_b32 %2, %2, %6\n v_and_b32 %3, %3, %7" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));)
    } else if constexpr (KIND == 1) { // v_bitop3_b32, three VGPR sources
      REP16(asm volatile("v_bitop3_b32 %0, %4, %8, %0 bitop3:0xea\n v_bitop3_b32 %1, %5, %9, %1 bitop3:0xea\n v_bitop3_b32 %2, %6, %10, %2 bitop3:0xea\n v_bitop3_b32 %3, %7, %11, %3 bitop3:0xea" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
    } else if constexpr (KIND == 2) { // v_and_or_b32 (VOP3, three VGPR sources)
      REP16(asm volatile("v_and_or_b32 %0, %4, %8, %0\n v_and_or_b32 %1, %5, %9, %1\n v_and_or_b32 %2, %6, %10, %2\n v_and_or_b32 %3, %7, %11, %3" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
    } else if constexpr (KIND == 3) { // v_bitop3_b32 with two VGPR sources (src2 == src0)
      REP16(asm volatile("v_bitop3_b32 %0, %0, %4, %0 bitop3:0xea\n v_bitop3_b32 %1, %1, %5, %1 bitop3:0xea\n v_bitop3_b32 %2, %2, %6, %2 bitop3:0xea\n v_bitop3_b32 %3, %3, %7, %3 bitop3:0xea" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));)
    } else if constexpr (KIND == 4) { // v_bcnt_u32_b32 accumulate
      REP16(asm volatile("v_bcnt_u32_b32 %0, %4, %0\n v_bcnt_u32_b32 %1, %5, %1\n v_bcnt_u32_b32 %2, %6, %2\n v_bcnt_u32_b32 %3, %7, %3" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));)
    } else if constexpr (KIND == 5) { // DPP adds on four independent registers
      REP16(asm volatile("v_add_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3));)
    } else if constexpr (KIND == 6) { // v_or3_b32
      REP16(asm volatile("v_or3_b32 %0, %4, %8, %0\n v_or3_b32 %1, %5, %9, %1\n v_or3_b32 %2, %6, %10, %2\n v_or3_b32 %3, %7, %11, %3" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
    } else if constexpr (KIND == 7) { // bitop3 with source registers in distinct banks (v4n, v4n+1, v4n+2): explicit registers
      REP16(asm volatile("v_bitop3_b32 v40, v44, v49, v40 bitop3:0xea\n v_bitop3_b32 v41, v45, v50, v41 bitop3:0xea\n v_bitop3_b32 v42, v46, v51, v42 bitop3:0xea\n v_bitop3_b32 v43, v47, v48, v43 bitop3:0xea" ::: "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51");)
    } else if constexpr (KIND == 8) { // bitop3 with all three sources in the same bank (v40, v44, v48)
      REP16(asm volatile("v_bitop3_b32 v40, v44, v48, v40 bitop3:0xea\n v_bitop3_b32 v41, v45, v49, v41 bitop3:0xea\n v_bitop3_b32 v42, v46, v50, v42 bitop3:0xea\n v_bitop3_b32 v43, v47, v51, v43 bitop3:0xea" ::: "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51");)
    } else if constexpr (KIND == 9) { // v_pk_add_u16, VGPR sources (the weighted engine's min-plus arithmetic)
      REP16(asm volatile("v_pk_add_u16 %0, %0, %4\n v_pk_add_u16 %1, %1, %5\n v_pk_add_u16 %2, %2, %6\n v_pk_add_u16 %3, %3, %7" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));)
    } else if constexpr (KIND == 10) { // v_pk_min_u16
      REP16(asm volatile("v_pk_min_u16 %0, %0, %4\n v_pk_min_u16 %1, %1, %5\n v_pk_min_u16 %2, %2, %6\n v_pk_min_u16 %3, %3, %7" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));)
    } else if constexpr (KIND == 11) { // v_pk_add_u16 with a scalar source (a cost from the scalar cache)
      REP16(asm volatile("v_pk_add_u16 %0, %4, s20\n v_pk_add_u16 %1, %5, s21\n v_pk_add_u16 %2, %6, s22\n v_pk_add_u16 %3, %7, s23" : "=v"(c0), "=v"(c1), "=v"(c2), "=v"(c3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "s20", "s21", "s22", "s23");)
    } else if constexpr (KIND == 12) { // the pair as the kernel issues it: t = a + cost (scalar); m = min(m, t)
      REP16(asm volatile("v_pk_add_u16 %4, %8, s20\n v_pk_min_u16 %0, %0, %4\n v_pk_add_u16 %5, %9, s21\n v_pk_min_u16 %1, %1, %5\n v_pk_add_u16 %6, %10, s22\n v_pk_min_u16 %2, %2, %6\n v_pk_add_u16 %7, %11, s23\n v_pk_min_u16 %3, %3, %7" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "s20", "s21", "s22", "s23");)
    } else if constexpr (KIND == 13) { // 32-bit: v_add_u32 + v_min_u32
      REP16(asm volatile("v_add_u32 %4, %8, s20\n v_min_u32 %0, %0, %4\n v_add_u32 %5, %9, s21\n v_min_u32 %1, %1, %5\n v_add_u32 %6, %10, s22\n v_min_u32 %2, %2, %6\n v_add_u32 %7, %11, s23\n v_min_u32 %3, %3, %7" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "s20", "s21", "s22", "s23");)
    } else if constexpr (KIND == 14) { // v_min3_u32
      REP16(asm volatile("v_min3_u32 %0, %0, %4, %8\n v_min3_u32 %1, %1, %5, %9\n v_min3_u32 %2, %2, %6, %10\n v_min3_u32 %3, %3, %7, %11" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0 ^ c1 ^ c2 ^ c3;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND>
void run(const char *name)
{
  uint32_t *out; unsigned long long *cyc;
  hipMalloc(&out, 256 * 2048 * 4); hipMalloc(&cyc, 256 * 32 * 8);
  const int iters = 2000;
  printf("%-44s", name);
  for (int wps : {1, 2, 4, 8}) {                   // waves per SIMD: block = 256 * wps threads, one block per CU
    const int threads = 256 * wps > 1024 ? 1024 : 256 * wps;
    const int blocks = 256 * (256 * wps / threads);          // 8 waves per SIMD: two 1024-thread blocks per CU
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 10);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (hipGetLastError() != hipSuccess) printf(" [launch error]");
    // instructions per SIMD = wps waves * iters * 64; at 4 cycles each and f GHz ...
    const double instr_per_simd = (double)wps * iters * 64.0 * ((KIND == 12 || KIND == 13) ? 2.0 : 1.0);
    printf("  %dw: %.2f ns/instr/SIMD", wps, ms * 1e6 / instr_per_simd);
  }
  printf("\n");
  hipFree(out); hipFree(cyc);
}

int main()
{
  run<0>("v_and_b32 (VOP2)");
  run<1>("v_bitop3_b32 3 VGPR srcs");
  run<2>("v_and_or_b32 3 VGPR srcs");
  run<3>("v_bitop3_b32 2 distinct VGPR srcs");
  run<4>("v_bcnt_u32_b32");
  run<5>("v_add_u32_dpp quad_perm");
  run<6>("v_or3_b32 3 VGPR srcs");
  run<7>("v_bitop3_b32 srcs in 3 different banks");
  run<8>("v_bitop3_b32 srcs in the same bank");
  run<9>("v_pk_add_u16 VGPR srcs");
  run<10>("v_pk_min_u16 VGPR srcs");
  run<11>("v_pk_add_u16 VGPR + SGPR src");
  run<12>("v_pk_add_u16 (SGPR) + v_pk_min_u16 pairs (per instr)");
  run<13>("v_add_u32 (SGPR) + v_min_u32 pairs (per instr)");
  run<14>("v_min3_u32 3 VGPR srcs");
  printf("(4 cycles at 2.4 GHz = 1.67 ns)\n");
  return 0;
}
