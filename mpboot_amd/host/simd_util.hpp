// simd_util.hpp -- the two host loops that touch every candidate of a sweep (their minimum; the next candidate that is not
// worse than the best so far), eight at a time where the CPU has AVX2 (checked once at run time).
#pragma once
#include <immintrin.h>
#include <stdint.h>

namespace mpf {

inline bool cpu_has_avx2()
{
  static const bool v = __builtin_cpu_supports("avx2");
  return v;
}

__attribute__((target("avx2"))) inline uint32_t min_u32_avx2(const uint32_t *o, int cnt)
{
  __m256i m = _mm256_set1_epi32(-1);
  int c = 0;
  for (; c + 8 <= cnt; c += 8) m = _mm256_min_epu32(m, _mm256_loadu_si256(reinterpret_cast<const __m256i *>(o + c)));
  __m128i h = _mm_min_epu32(_mm256_castsi256_si128(m), _mm256_extracti128_si256(m, 1));
  h = _mm_min_epu32(h, _mm_shuffle_epi32(h, 0x4E));
  h = _mm_min_epu32(h, _mm_shuffle_epi32(h, 0xB1));
  uint32_t r = (uint32_t)_mm_cvtsi128_si32(h);
  for (; c < cnt; c++) r = o[c] < r ? o[c] : r;
  return r;
}

inline uint32_t min_u32(const uint32_t *o, int cnt)
{
  if (cpu_has_avx2()) return min_u32_avx2(o, cnt);
  uint32_t r = 0xFFFFFFFFu;
  for (int c = 0; c < cnt; c++) r = o[c] < r ? o[c] : r;
  return r;
}

// smallest index in [k, cnt) with o[index] <= thr, cnt if there is none
__attribute__((target("avx2"))) inline int first_le_avx2(const uint32_t *o, int k, int cnt, uint32_t thr)
{
  const __m256i t = _mm256_set1_epi32((int)thr);
  for (; k + 8 <= cnt; k += 8) {
    const __m256i v = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(o + k));
    const int mask = _mm256_movemask_ps(_mm256_castsi256_ps(_mm256_cmpeq_epi32(_mm256_min_epu32(v, t), v)));   // v <= t
    if (mask) return k + __builtin_ctz((unsigned)mask);
  }
  for (; k < cnt; k++)
    if (o[k] <= thr) return k;
  return cnt;
}

inline int first_le(const uint32_t *o, int k, int cnt, uint32_t thr)
{
  if (cpu_has_avx2()) return first_le_avx2(o, k, cnt, thr);
  for (; k < cnt; k++)
    if (o[k] <= thr) return k;
  return cnt;
}

}  // namespace mpf
