/*
 * multi_device_driver.cpp -- TEST INFRASTRUCTURE ONLY: runs integration/multi_device.hpp (G engines on G host threads inside one
 * process, replicate b on device b % G) on the devices given and prints every replicate's score and final tree.
 *
 * stdin: n P datatype(0|1) maxtrav base_seed G dev_0 .. dev_{G-1} R ; n rows of P tip codes ; P base weights ;
 *        then per replicate: P weights, 3(2n-1) record links.   "map" as first argument prints only the unit -> device schedule.
 */
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../integration/multi_device.hpp"

int main(int argc, char **argv)
{
  if (argc > 1 && !std::strcmp(argv[1], "map")) {       // CPU-only: the schedule
    const int units = std::atoi(argv[2]), G = std::atoi(argv[3]);
    for (int d = 0; d < G; d++) {
      std::printf("device %d:", d);
      for (int b : mpf_md::units_of_device(units, d, G)) std::printf(" %d(seed %d)", b, mpf_md::unit_seed(7, b));
      std::printf("\n");
    }
    for (int b = 0; b < units; b++)
      if (mpf_md::device_of_unit(b, G) != b % G) return 1;
    return 0;
  }
  int n, P, dt, maxtrav, seed, G, R;
  if (std::scanf("%d %d %d %d %d %d", &n, &P, &dt, &maxtrav, &seed, &G) != 6) return 2;
  std::vector<int> devs((size_t)G);
  for (int &d : devs) if (std::scanf("%d", &d) != 1) return 2;
  if (std::scanf("%d", &R) != 1) return 2;
  std::vector<uint8_t> codes((size_t)n * (size_t)P);
  for (auto &c : codes) { int v; if (std::scanf("%d", &v) != 1) return 2; c = (uint8_t)v; }
  std::vector<int32_t> w0((size_t)P);
  for (auto &w : w0) if (std::scanf("%d", &w) != 1) return 2;
  const size_t nrec = 3 * (size_t)(2 * n - 1);
  std::vector<std::vector<int32_t>> ws((size_t)R, std::vector<int32_t>((size_t)P)), st((size_t)R, std::vector<int32_t>(nrec)),
      fin((size_t)R, std::vector<int32_t>(nrec));
  std::vector<mpf_md::Replicate> reps((size_t)R);
  for (int b = 0; b < R; b++) {
    for (auto &w : ws[(size_t)b]) if (std::scanf("%d", &w) != 1) return 2;
    for (auto &r : st[(size_t)b]) if (std::scanf("%d", &r) != 1) return 2;
    reps[(size_t)b] = mpf_md::Replicate{ws[(size_t)b].data(), st[(size_t)b].data(), fin[(size_t)b].data(), 0u};
  }
  mpf_config cfg;
  std::memset(&cfg, 0, sizeof cfg);
  cfg.n_taxa = n; cfg.n_patterns = P; cfg.datatype = dt;
  const std::string err = mpf_md::refine_replicates(devs, cfg, codes.data(), w0.data(), reps, seed, maxtrav);
  if (!err.empty()) { std::fprintf(stderr, "%s\n", err.c_str()); return 1; }
  for (int b = 0; b < R; b++) {
    std::printf("replicate %d device %d score %u tree", b, devs[(size_t)mpf_md::device_of_unit(b, G)], reps[(size_t)b].score);
    for (int32_t r : fin[(size_t)b]) std::printf(" %d", r);
    std::printf("\n");
  }
  return 0;
}
