/*
 * rng.h -- the two generators the reference's parsimony path draws from.
 * TEST INFRASTRUCTURE (oracle) -- the product has its own copy in
 * mpboot_amd/host/rng.hpp; this one is the checker.
 *
 * (1) tie-break generator = random_double() (tools.cpp:3363-3368) = SPRNG's
 *     64-bit LCG with prime addend (sprng/lcg64.c):
 *       init : lcg64.c:199-204   state = 0x2bc6ffff8cfe166d ^ ((u64)seed << 33 | stream)
 *              multiplier = PARAMLIST[0] = 0x27bb2ee687b0b0fd (lcg64.c:63, :197)
 *              prime      = prime_list[stream] (primes-lcg64.c:64-68, primelist-lcg64.h:7) = 3037000493 for stream 0
 *       next : lcg64.c:220, :268  state = state*multiplier + prime ; return state * 2^-64
 *     The reference creates stream 0 of 1 (tools.cpp:3326), so no warm-up draws
 *     (lcg64.c:209 loops 127*stream_number times).
 *     Pinned against sprng/ compiled where it lies: oracle/_ref/sprng_ref.
 * (2) addition-order generator = PLL randum() (pllrepo/src/utils.c:335-358).
 */
#ifndef ORACLE_RNG_H
#define ORACLE_RNG_H
#include <stdint.h>

typedef struct { uint64_t state, mult; uint32_t prime; } orc_lcg64;

static inline void orc_lcg64_init(orc_lcg64 *g, int seed)
{
  g->mult = ((uint64_t)0x27bb2ee6u << 32) | 0x87b0b0fdu;
  g->prime = 3037000493u;
  g->state = (((uint64_t)0x2bc6ffffu << 32) | 0x8cfe166du) ^ ((uint64_t)(int64_t)seed << 33);
}

static inline double orc_lcg64_next(orc_lcg64 *g)
{
  g->state = g->state * g->mult + g->prime;
  return (double)g->state * 5.4210108624275222e-20; /* 2^-64 */
}

static inline double orc_randum(long *seed)
{
  long sum, mult0 = 1549, mult1 = 406, s0, s1, s2, n0, n1, n2;
  s0 = *seed & 4095;
  sum = mult0 * s0;
  n0 = sum & 4095;
  sum >>= 12;
  s1 = (*seed >> 12) & 4095;
  sum += mult0 * s1 + mult1 * s0;
  n1 = sum & 4095;
  sum >>= 12;
  s2 = (*seed >> 24) & 255;
  sum += mult0 * s2 + mult1 * s1;
  n2 = sum & 255;
  *seed = n2 << 24 | n1 << 12 | n0;
  return 0.00390625 * (n2 + 0.000244140625 * (n1 + 0.000244140625 * n0));
}
#endif
