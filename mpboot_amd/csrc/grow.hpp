// grow.hpp -- launch interface of the device-resident randomized stepwise addition (grow.hip).
//
// _pllMakeParsimonyTreeFast's addition loop (reference sprparsimony.cpp:3107-3181) inserts the taxa of a random permutation one
// after the other: stepwiseAddition (:2977-3019) tests the new tip on every branch of the tree built so far -- a depth-first walk
// from the start tip with the descent cut `parsimonyScore[q] > 0` (:3014) and the random tie rule (:3004) -- and the tip goes
// where the tree gets shortest (:3158-3171).  Every insertion depends on the one before: driven from the host that is an upload,
// a refresh launch, a scan launch and a round trip per taxon (~130 us each, 0.13 s per 1000-taxon tree).  k_grow keeps the
// whole loop on the GPU, one launch per tree:
//   * one persistent workgroup per TILE of sites, as in k_climb; every workgroup holds the ROOTED tree (hung from the start tip:
//     parent, children in the reference's visiting order, pre-order position, subtree size, depth per node) in LDS and runs the
//     same deterministic control code;
//   * only the "down" vectors D(c) -- the subtree below c, away from the start tip -- are kept in HBM.  Inserting a tip changes
//     them along ONE root path (<= tree height operations, a chain in registers); the "up" vectors U(c) it would make stale
//     everywhere else are never stored: U(child) = fitch(U(parent), D(sibling)) is recomputed top-down while the candidates are
//     scored -- one vector load per branch, cost(c) = |sites where fitch(U(c), D(c)) and the new tip have no common state|;
//   * the tree is cut into a small SKELETON (nodes with more than S descendants; one wave walks it and leaves U of every
//     cut-off subtree's root in HBM) and parts of at most S nodes that the waves then walk independently;
//   * per step the workgroups exchange one row each (their share of every candidate's cost, two 16-bit sums per word, plus
//     one bit per node of the last root path: "this join costs something here", for the descent cut) and read all rows.
// The host replays the insertions the kernel reports onto its topology mirror.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.hpp"

namespace mpf {

enum GrowReason : uint32_t { GROW_RUNNING = 0, GROW_DONE = 1, GROW_ABORT = 4, GROW_ERROR = 5 };

struct GrowHeader {
  unsigned long long rng;          // TieRng::state, in and out
  unsigned long long draws;
  uint32_t steps_done, reason, err, len;     // len: length of the tree after the last insertion
  uint32_t arrive, abort, start_gate, xarrive;
  uint32_t pad[4];
  unsigned long long tph[8];       // 100 MHz ticks workgroup 0 spent per phase: plan, skeleton, parts, exchange, decide, insert, path
};

struct GrowParams {
  uint32_t *vec;                   // the engine's vector store [nslots][S][Wp]
  uint32_t n, nslots, Wp, tiles;
  uint32_t m0;                     // nodes below the start tip in the start tree (2 * tips - 3)
  uint32_t steps;                  // taxa to add
  uint32_t tie_mode;
  uint32_t len0;                   // length of the start tree
  uint32_t root_cid;               // the start tip's vector: U of the root's child
  uint32_t root_node;              // node id of that child
  // start tree, node ids: tips 0 .. n-1 (= their vector ids), inner node number v -> n + (v - n - 1).  Seven arrays of 2n entries
  // (parent, first child, second child -- in the reference's visiting order --, pre-order position, subtree size, depth, vector id
  // of the node's down vector), then the pre-order array itself (2n entries)
  const uint16_t *init;
  const uint16_t *tips;            // [steps] tip added at step s
  // the inner node that step s puts into the tree (tr->nodep[nextnode++], whichever node and record that is after earlier
  // nodeRectifierPars calls): its node id, and the vector id of the record that becomes its down record (q->next->next)
  const uint16_t *xnode, *xdcid;
  uint32_t *xrow;                  // [2][tiles][xstride] exchange rows
  uint32_t xstride;
  uint32_t *park;                  // [tiles][waves * kGrowParkPart + kGrowParkSkel][R][64] parked up-vectors
  uint32_t *ucp;                   // [tiles][kGrowMaxParts][R][64] U of the parts' roots
  GrowHeader *hdr;
  uint32_t *out;                   // [steps][2]: vector id of the insertion branch's down record, tree length
  uint32_t fault;                  // tests: 0xFFFFFFFF = the start barrier decides "abort"
};

constexpr uint32_t kGrowNone = 0xFFFFu;
constexpr uint32_t kGrowParkPart = 64;     // levels a part's walk can park = the largest part (nodes): one ballot gives its walk
constexpr uint32_t kGrowParkSkel = 512;    // levels of the skeleton's walk (deeper: GROW_ERROR 3, the host's loop takes over)
constexpr uint32_t kGrowMaxParts = 1024;

bool grow_supported(const Geometry &g, int n_taxa);
int grow_tiles(const Geometry &g, int vw);
int grow_waves(const Geometry &g, int vw);
size_t grow_lds_bytes(const Geometry &g, int n_taxa, int vw);
int grow_blocks_per_cu(const Geometry &g, int n_taxa, int vw);      // workgroups of the launch that fit one CU together (registers, LDS)
size_t grow_vec_words(const Geometry &g, int vw);      // 32-bit words of one tile of a vector (R * 64)
hipError_t launch_grow(hipStream_t st, const Geometry &g, int vw, const GrowParams &p);

}  // namespace mpf
