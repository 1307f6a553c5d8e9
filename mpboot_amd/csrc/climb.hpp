// climb.hpp -- launch interface of the device-resident SPR hill climb (climb.hip).
//
// pllOptimizeSprParsimony's sweep (reference sprparsimony.cpp:3295-3316) visits one prune node after the other and
// applies a move as soon as one is accepted, so every accepted move depends on the one before it: driven from the host
// that is one launch chain + one synchronisation per move (~75 us).  k_climb keeps the whole loop on the GPU:
//   * one persistent workgroup per TILE of sites (16 * VW words of every state row); a tile's directional vectors are
//     read and written by that workgroup only, for the whole launch;
//   * every workgroup holds the topology (back links in compact vector ids), the validity flags and the search state
//     (best score, tie counters, the lcg64 tie stream) in LDS and runs the SAME deterministic control code: candidate
//     enumeration in the reference's order, the closure of stale vectors, the accept / tie rules, the topology edit and
//     its invalidation -- nothing of that is ever communicated;
//   * the only exchange per step is the sum over tiles of the candidates' partial lengths (one 64-bit atomic add per
//     candidate and tile carrying value + arrival count; polled until every tile has arrived).
// The host launches one segment per sweep (nodeRectifierPars runs between sweeps, sprparsimony.cpp:3297) and replays
// the moves the kernel reports onto its own topology mirror.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.hpp"

namespace mpf {

enum ClimbReason : uint32_t {
  CLIMB_RUNNING = 0,
  CLIMB_SWEEP_END = 1,     // every prune node of the sweep visited
  CLIMB_IDLE = 2,          // no move for idle_limit prune nodes: the host's whole-chip batches are the better tool
  CLIMB_MOVES_FULL = 3,    // the move list is full
  CLIMB_ABORT = 4,         // the workgroups did not all become resident in time: nothing was changed
  CLIMB_ERROR = 5,         // an internal bound was hit (hdr.err says which); state is not to be trusted
  CLIMB_CUTOFF = 6         // stop_len: the prune node at hdr.pos has an insertion test no longer than that -- not visited, no draw taken for it
};

// search state handed over in both directions + counters + the launch's synchronisation words
struct ClimbHeader {
  unsigned long long rng;          // TieRng::state
  unsigned long long hits;         // bestTreeScoreHits
  uint32_t best, randomMP, iter_hits, pos;   // bestParsimony, randomMP, bestIterationScoreHits, next prune index (1-based)
  int32_t insert_cid, remove_cid;  // insertNode / removeNode as vector ids (-1: none)
  uint32_t n_moves, reason, err, steps;
  unsigned long long n_tests, n_ops, draws, n_scanned_nodes;
  uint32_t arrive, abort, since_move, batch;
  uint32_t pad[4];
  uint32_t start_gate, pad2[3];    // 0 = undecided, 1 = every workgroup has arrived: go, 2 = somebody timed out: nobody starts (ONE compare-and-swap decides)
  unsigned long long tph[16];      // 100 MHz ticks workgroup 0 spent per phase: set-up, enumerate, closure, refresh, scan, exchange, decide
  // ClimbParams::sweeps_inside: startMP of the sweep under way (in: the host's; out: of the sweep the launch ended in) and the sweeps
  // that were started inside the launch
  uint32_t start_mp, sweeps, pad3[2];
};

struct ClimbParams {
  uint32_t *vec;                   // the engine's vector store [nslots][S][Wp]
  uint32_t n, nslots, Wp, tiles;
  uint32_t total;                  // prune nodes per sweep (2n - 2)
  uint32_t maxtrav;                // min(maxtrav, ntips - 3), 1..6
  uint32_t tie_mode;               // MPF_TIE_RANDOM | MPF_TIE_FIRST
  uint32_t idle_limit;             // 0 = never leave for idleness
  uint32_t max_moves;
  uint32_t batch_min, batch_max;   // prune nodes per step (speculative; doubles after a step without a move)
  uint32_t near_q;                 // the batch behind a near move: this many quarters of the recent average distance between moves (3)
  uint16_t *order;                 // [total] vector ids of nodep[1..total] (in; out with sweeps_inside)
  uint16_t *bk;                    // [nslots] back links as vector ids (in: current tree, out: after the moves)
  uint32_t *sct;                   // [tiles][nslots][16] per-tile, per-word-group subtree scores (scratch of the launch; [64] per-word ones in the word-major shape)
  unsigned long long *gsum;        // [3][kClimbCap] exchange ring (zeroed by the host before the launch)
  // two-level exchange: every eighth workgroup forms a group that first sums into words of its own, the group's last arrival
  // forwards the total -- same-address atomics are served one after the other: 12-13 + 8 deep instead of 98
  unsigned long long *xsum;        // [8][3][kClimbCap], zeroed like gsum
  uint32_t *xcnt;                  // (unused: group sizes follow from the workgroup count)
  ClimbHeader *hdr;
  uint32_t *moves;                 // [max_moves][3] = remove cid, insert cid, score
  uint32_t *trace;                 // optional: 8 words per visited prune node
  uint32_t trace_cap;              // in records
  uint32_t *beat;                  // optional, pinned host memory: progress marks of workgroup 0 (16 words)
  // tests only (engine option "climb_fault"): 0 = none; 0xFFFFFFFF = the start barrier decides "abort"; k = in the exchange of
  // step k the last workgroup withholds its sums, so that the others run into their time-out (recovery paths of climb_host.cpp)
  uint32_t fault;
  // 0 = none.  Otherwise the launch ends IN FRONT of the first prune node one of whose insertion tests gives a tree of at most
  // this length (CLIMB_CUTOFF): with -bb and a logl_cutoff in force such a tree is the first one IQTree::saveCurrentTree would
  // book (iqtree.cpp:3343) -- up to there the climb is the plain one, from there on the host's tracked path takes over
  uint32_t stop_len;
  // Workgroups of the launch (1 .. tiles).  groups == tiles: one tile of sites per workgroup, the form that gets a single climb through
  // its chain of dependent steps fastest (98 workgroups at C3).  Fewer: workgroup g works through the tiles g, g + groups, ... one
  // after the other inside every step (refresh + scan per tile, the candidates' sums accumulating in its LDS) and only `groups`
  // sums meet in the exchange -- with groups == 1 nothing crosses between workgroups at all: a climb is ONE resident workgroup
  // and a chip holds hundreds of them (engine option "climb_groups")
  uint32_t groups;
  // 1 = a sweep that ends with the climb not yet at its optimum (randomMP < startMP, reference :3316) is followed by the next one
  // INSIDE the launch: nodeRectifierPars (:2046-2101) runs on the topology in LDS, the vectors stay valid (a new launch would have
  // to recompute every one of them in its first step).  The launch then ends at the optimum, or with a full move list.  `order`
  // comes back as the last rectification left it.  (k_climb_many: a climb is one launch; off for k_climb, whose 98 workgroups would
  // each walk the tree for it)
  uint32_t sweeps_inside;
  uint16_t *snap_r;                // [groups][nslots] the ready list of such a step
  uint32_t *snap;                  // [groups][nslots + nslots / 4 + 1] words: the claim words and validity flags of a step whose closure takes the plain dataflow path, restored per tile (groups < tiles only)
};

constexpr uint32_t kClimbCap = 1024;      // candidates per step

// states per lane group: DNA 1 (four lanes = the four states of a word), protein 5
// (batch_max: the launch's ClimbParams::batch_max -- the control state of a step's scan programs grows with it: 1000 taxa fit with
//  sixteen prune nodes per step, about 1600 with the plain climb's eight)
bool climb_supported(const Geometry &g, int n_taxa, int maxtrav, int batch_max = 16);
int climb_tiles(const Geometry &g, int vw);
// LDS of a workgroup: batch_max = ClimbParams::batch_max (a step's scan programs are sized by it), many = k_climb_many's wave count,
// word_major = the word-major shape on 64-word tiles (vw 4)
size_t climb_lds_bytes(const Geometry &g, int n_taxa, int vw, int batch_max = 16, bool many = false, bool word_major = false);
hipError_t launch_climb(hipStream_t st, const Geometry &g, int vw, const ClimbParams &p, bool word_major = false);
// n_climbs independent climbs, one workgroup each (k_climb_many): d_params[n_climbs] on the device, every entry with groups == 1
// word_major (four-state data on 64-word tiles, vw == 4): a lane holds the four states of a word (quadtile.hpp, kWordMajor); sct then
// has 64 score words per vector and tile instead of 16
hipError_t launch_climb_many(hipStream_t st, const Geometry &g, int vw, const ClimbParams *d_params, int n_climbs, size_t lds_bytes, bool word_major);

}  // namespace mpf
