// lcg_block.hpp -- eight draws of the tie stream at once (x86-64 with AVX-512 F + DQ; callers ask lcg64_have_avx512() first).
//
// random_double() is SPRNG's 64-bit LCG (rng.hpp, TieRng): state' = state * A + C, value = state' * 2^-64.  A chain of dependent
// multiply-adds -- four cycles a draw -- is what the replay of the current tree's bookings over 1000 samples waits for at every
// prune-node visit.  x(n+k) = A^k x(n) + C_k, so draws n+1 .. n+8 are ONE vector multiply-add of the state with constants; the
// conversion to double and the scaling are the same IEEE operations as the scalar code's (vcvtuqq2pd rounds to nearest even like
// the scalar conversion, 2^-64 is exact), so every value is bit for bit the scalar one (tests/test_lcg_block.py compares them).
#pragma once
#include <immintrin.h>
#include <stdint.h>

namespace mpf {

constexpr uint64_t kLcg64A = 0x27bb2ee687b0b0fdULL, kLcg64C = 3037000493ULL;

struct Lcg64Jump8 {
  alignas(64) uint64_t a[8], c[8];                 // a[k] = A^(k+1), c[k] = C * (A^k + ... + 1)
  Lcg64Jump8()
  {
    uint64_t ak = 1, ck = 0;
    for (int k = 0; k < 8; k++) { ck = ck * kLcg64A + kLcg64C; ak *= kLcg64A; a[k] = ak; c[k] = ck; }
  }
};

inline bool lcg64_have_avx512()
{
  static const bool ok = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq") && __builtin_cpu_supports("avx512vl");
  return ok;
}

// values of the draws 1 .. 8 behind `state` (lane k = draw k + 1); returns the state after the eighth
__attribute__((target("avx512f,avx512dq,avx512vl"))) inline uint64_t lcg64_draw8(uint64_t state, const Lcg64Jump8 &j, __m512d *values)
{
  const __m512i s = _mm512_add_epi64(_mm512_mullo_epi64(_mm512_set1_epi64((long long)state), _mm512_load_si512((const void *)j.a)),
                                     _mm512_load_si512((const void *)j.c));
  *values = _mm512_mul_pd(_mm512_cvtepu64_pd(s), _mm512_set1_pd(5.4210108624275222e-20));
  return state * j.a[7] + j.c[7];
}

}  // namespace mpf
