/*
 * sprng_ref.c -- TEST INFRASTRUCTURE ONLY.
 * Prints the first K doubles of the reference's tie-break generator
 * random_double() (tools.cpp:3363-3368 -> sprng(randstream), stream created by
 * init_sprng(0, 1, seed, SPRNG_DEFAULT), tools.cpp:3326) so that our restatement
 * of SPRNG's 64-bit LCG (oracle/rng.h) can be pinned against the vendored
 * sprng/ sources compiled where they lie.
 */
#include <stdio.h>
#include <stdlib.h>
#include "sprng/sprng.h"

int main(int argc, char **argv)
{
  int seed = argc > 1 ? atoi(argv[1]) : 1;
  int k = argc > 2 ? atoi(argv[2]) : 16, i;
  int *stream = init_sprng(0, 1, seed, SPRNG_DEFAULT);
  for (i = 0; i < k; i++) printf("%.17g\n", sprng(stream));
  return 0;
}
