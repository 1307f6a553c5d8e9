// rng.hpp -- the two generators the reference's parsimony path draws from (product copy).
//
// TieRng  : random_double() of the reference (tools.cpp:3363-3368) = SPRNG 64-bit LCG with prime
//           addend, stream 0 of 1 as created by init_random(seed) (tools.cpp:3320-3331):
//             state0 = 0x2bc6ffff8cfe166d ^ (seed << 33)              sprng/lcg64.c:199-204
//             state  = state * 0x27bb2ee687b0b0fd + 3037000493        sprng/lcg64.c:220 (+ :63, primelist-lcg64.h:7)
//             value  = state * 2^-64                                   sprng/lcg64.c:268
// randum(): PLL's addition-order generator (pllrepo/src/utils.c:335-358).
#pragma once
#include <cstdint>

namespace mpf {

struct TieRng {
  uint64_t state = 0;
  void seed(int32_t s) { state = 0x2bc6ffff8cfe166dULL ^ ((uint64_t)(int64_t)s << 33); }
  double next()
  {
    state = state * 0x27bb2ee687b0b0fdULL + 3037000493ULL;
    return (double)state * 5.4210108624275222e-20;
  }
};

inline double randum(int64_t *seed)
{
  const int64_t m0 = 1549, m1 = 406;
  int64_t s0 = *seed & 4095, s1 = (*seed >> 12) & 4095, s2 = (*seed >> 24) & 255;
  int64_t sum = m0 * s0;
  const int64_t n0 = sum & 4095;
  sum >>= 12;
  sum += m0 * s1 + m1 * s0;
  const int64_t n1 = sum & 4095;
  sum >>= 12;
  sum += m0 * s2 + m1 * s1;
  const int64_t n2 = sum & 255;
  *seed = n2 << 24 | n1 << 12 | n0;
  return 0.00390625 * ((double)n2 + 0.000244140625 * ((double)n1 + 0.000244140625 * (double)n0));
}

}  // namespace mpf
