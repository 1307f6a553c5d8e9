// multi_device.hpp -- several GPUs inside ONE host process (mpboot is one C++ process): G engines on G host threads, the
// independent units of the reference's loops dealt out by index.  No collective, no second process.
//
//   units: bootstrap-refinement replicates of IQTree::optimizeBootTrees (iqtree.cpp:2515-2866: re-weight with
//          boot_samples_pars[b], climb from boot_trees[b]) or the start trees of initCandidateTreeSet
//          (phyloanalysis.cpp:1270-1317)
//   map:   unit b -> device b % G, visited in increasing b on that device; tie draws of unit b come from its own stream
//          seeded ran_seed + b * 12345 (as the reference seeds start trees, phyloanalysis.cpp:1273), so a unit's result does
//          not depend on G or on the thread schedule
#pragma once
#include <cstdint>
#include <string>
#include <thread>
#include <vector>

#include "../include/mpfitch.h"

namespace mpf_md {

inline int device_of_unit(int unit, int n_devices) { return unit % n_devices; }
inline int unit_seed(int base_seed, int unit) { return base_seed + unit * 12345; }
// the units device d works through, in order
inline std::vector<int> units_of_device(int n_units, int d, int n_devices)
{
  std::vector<int> u;
  for (int b = d; b < n_units; b += n_devices) u.push_back(b);
  return u;
}

struct Replicate {
  const int32_t *weights;      // boot_samples_pars[b] widened to int32, [P]
  const int32_t *start_back;   // boot_trees[b] as record links, [3 (2n - 1)]
  int32_t *final_back;         // out
  uint32_t score;              // out: boot_logl[b] = -score
};

// IQTree::optimizeBootTrees' default branch over `reps`, replicate b on device devices[b % G], one host thread per device.
// Returns "" or the first error text.  `codes` / `base_weights` as mpf_engine_create takes them.
inline std::string refine_replicates(const std::vector<int> &devices, const mpf_config &cfg_in, const uint8_t *codes,
                                     const int32_t *base_weights, std::vector<Replicate> &reps, int base_seed, int maxtrav)
{
  const int G = (int)devices.size();
  std::vector<std::string> err((size_t)G);
  std::vector<std::thread> th;
  for (int d = 0; d < G; d++)
    th.emplace_back([&, d]() {
      mpf_config cfg = cfg_in;
      cfg.device = devices[(size_t)d];
      mpf_engine *e = nullptr;
      auto fail = [&](const char *what) { err[(size_t)d] = std::string(what) + ": " + mpf_last_error(); if (e) mpf_engine_destroy(e); };
      if (mpf_engine_create(&e, &cfg, codes, base_weights)) return fail("mpf_engine_create");
      for (int b : units_of_device((int)reps.size(), d, G)) {
        Replicate &r = reps[(size_t)b];
        if (mpf_set_weights(e, r.weights)) return fail("mpf_set_weights");                  // modifyPatternFreq, :2520
        if (mpf_seed_ties(e, MPF_TIE_RANDOM, unit_seed(base_seed, b))) return fail("mpf_seed_ties");
        if (mpf_reset_node_order(e)) return fail("mpf_reset_node_order");
        if (mpf_set_tree(e, r.start_back)) return fail("mpf_set_tree");                     // readTreeString(boot_trees[b]), :2826
        if (mpf_optimize_spr(e, 1, maxtrav, &r.score)) return fail("mpf_optimize_spr");     // :2837
        if (mpf_get_tree(e, r.final_back)) return fail("mpf_get_tree");
      }
      mpf_engine_destroy(e);
    });
  for (std::thread &t : th) t.join();
  for (const std::string &s : err)
    if (!s.empty()) return s;
  return "";
}

}  // namespace mpf_md
