"""mpboot_amd/host/lcg_block.hpp: eight draws of the tie stream in one vector step must be, bit for bit, the eight draws the scalar
generator (host/rng.hpp: SPRNG's lcg64, pinned by tests/test_abi.py) produces one after the other -- state and value."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r'''
#include <cstdio>
#include <cstring>
#include "lcg_block.hpp"
__attribute__((target("avx512f,avx512dq,avx512vl"))) static int run()
{
  mpf::Lcg64Jump8 j;
  uint64_t st = 0x2bc6ffff8cfe166dULL ^ ((uint64_t)41 << 33), sv = st;
  long bad = 0;
  for (int it = 0; it < 200000; it++) {
    __m512d v;
    double out[8];
    sv = mpf::lcg64_draw8(sv, j, &v);
    _mm512_storeu_pd(out, v);
    for (int k = 0; k < 8; k++) {
      st = st * mpf::kLcg64A + mpf::kLcg64C;
      const double r = (double)st * 5.4210108624275222e-20;
      if (memcmp(&r, &out[k], 8)) bad++;
    }
    if (st != sv) bad++;
    if (it % 500 == 0) {                       /* other corners of the state space, high bit set and clear */
      st ^= (uint64_t)it * 0x9E3779B97F4A7C15ULL;
      st = (it & 1) ? (st | 0x8000000000000000ULL) : (st & 0x7FFFFFFFFFFFFFFFULL);
      sv = st;
    }
  }
  printf("%s %ld\n", bad ? "MISMATCH" : "ok", bad);
  return bad != 0;
}
int main() { if (!mpf::lcg64_have_avx512()) { printf("skip\n"); return 0; } return run(); }
'''


def test_eight_draws_at_once_equal_eight_scalar_draws(tmp_path):
    src = tmp_path / "t.cpp"
    src.write_text(SRC)
    exe = tmp_path / "t"
    cc = subprocess.run(["g++", "-O2", "-I", os.path.join(ROOT, "mpboot_amd", "host"), str(src), "-o", str(exe)], capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr[-2000:]
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    if out.stdout.strip() == "skip":
        pytest.skip("no AVX-512 F + DQ + VL on this host: the engine takes the scalar loop here as well")
    assert out.stdout.split()[0] == "ok"
