// kernels.hpp -- launch interface of the gfx950 Fitch kernels (kernels.hip).
//
// Data layout in HBM: one "directional vector" per node record,
// stored at a compact slot index; a vector is S rows (one per state) of Wp 32-bit
// words, bit j of word i = "state k possible at site 32*i+j" -- the reference's
// parsVect row layout (sprparsimony.cpp:732-734, :2926-2927), site-major so that a
// wavefront reading one row touches 64*VW consecutive words (256 B .. 1 KiB).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mpf {

// recompute dst = fitch(a, b); cnt[dst] += #sites with empty intersection   (K1, newview)
struct NvOp { uint32_t dst, a, b, pad; };

// popcount(~OR_k(a_k & b_k)) -> out[out]                                      (K2, evaluate)
struct EvOp { uint32_t a, b, out, pad; };

// one step of an SPR-scan program
//   kind 0  CHAIN : U[d] = fitch(U[d-1], vec[sib]); if test: out[out] += cost(fitch(U[d], vec[own]), S)
//   kind 1  ROOT  : U[0] = vec[own]
//   kind 2  JOIN  : out[out] += cost(fitch(vec[own], vec[sib]), S)            (stepwise addition)
struct ScanOp { uint32_t own, sib, meta, out; };   // meta = depth | test<<8 | kind<<16
struct ScanHdr { uint32_t op_begin, op_end, s_slot, pad /* bit 0: stepwise-addition program (the subtree s is the root side of every test) */; };

enum { SCAN_CHAIN = 0, SCAN_ROOT = 1, SCAN_JOIN = 2, SCAN_EVAL = 3 };   // EVAL (weighted kernels only): min_x(vec[sib][x] + m(S)[x]), the evaluate at a tip's edge

// device-walked scan: the kernel enumerates the neighbourhood of prune record x itself from the
// topology (back links) resident in HBM; candidate i of the scan lands in out[out_base + i] in the
// reference's DFS order, ncand[scan] = number of candidates
// All ids are compact vector slots ("cid"): tips 0..n-1, inner record 3v+s -> n + 3(v-n-1) + s.
// kids[cid] = cids of the two records behind an inner record (back[next], back[next next]).
struct WalkDesc { uint32_t s_cid, xa_cid, xb_cid, trav /* mintrav | maxtrav<<8 | side_mask<<16 | child_mask<<18 */,
                  out_base, pad0, pad1, pad2; };
constexpr int kWalkMaxDepth = 8;    // deepest device-walked scan (k_scan_walk); longer radii use the host-planned k_scan
constexpr int kMaxDepth = 12;   // deepest chain the register-resident scan kernel supports
// ... for 4- and 20-row tiles.  32-row tiles: 6 (thirteen of them do not fit a lane's 512 registers: k_scan<32, 1, 12> spilled);
// longer radii there take the HBM-scratch kernel like everything above kMaxDepth
constexpr int scan_reg_depth(int S, bool sankoff) { return sankoff ? (S != 4 ? 6 : kMaxDepth) : (S >= 32 ? 6 : kMaxDepth); }

struct Geometry {
  int S;        // states (4 | 20)
  int Wp;       // elements per row, multiple of 32 (Fitch: 32-site words; Sankoff: patterns)
  int vw;       // words per lane (1 | 2 | 4)
  int reduce;   // 0 = DPP wave reduction, 1 = ds_bpermute (__shfl) reduction
  int map;      // 0 = scan-major wave mapping, 1 = tiles pinned to XCD classes
  int big = 0;  // the vector store is 2 GiB or more: the scan kernel uses 64-bit addressing instead of one raw buffer
  int nv_pipe = 1;   // level-synchronous refresh, one word per lane: 1 = k_newview_wgq (operands requested a round ahead), 0 = k_newview_wgh
  // DNA, below 2 GiB: a second copy of every vector in WORD-major order (the four state words of a 32-site word side by side, 16 B)
  // `shoff` words behind the row-major store.  A CU loads 1 KB of it with ONE buffer_load_dwordx4 per wave at 124-146 GB/s, against
  // 75 GB/s for the four 256-byte row loads (tools/ubench/l1_rate: the load path inside the CU is what bounds the planned scan).
  // Written by k_pack_tips and k_newview_wgq, read by k_scan_prog while Engine::shadow_ok_ holds.
  size_t shoff = 0;
  // radii above kMaxDepth (k_scan_deep): the scans' per-level up-vectors live in this scratch area (allocated on first use)
  uint32_t *deep_scratch = nullptr;
  size_t deep_scratch_words = 0;
  int nv_tile = 0;   // ... on tiles of 32 | 16 | 8 | 4 words (Wp / tile workgroups); 0 = chosen from Wp (newview_tile)
  // Sankoff (weighted parsimony) mode: vectors hold one 32-bit cost per state and pattern
  int sankoff = 0;
  const uint32_t *cost = nullptr;   // device, [S][S]
  const uint32_t *costT = nullptr;  // device, the transposed matrix; nullptr: the matrix is symmetric (see k_snk_scan)
  const uint32_t *pwgt = nullptr;   // device, [Wp] pattern weights (0 on padding)
  uint32_t highest_cost = 0;
  int snk16 = 0;                    // weighted mode: two 16-bit costs per lane (v_pk_add_u16 / v_pk_min_u16)
  size_t moff = 0;                  // words from a vector to its min-plus transform m(v) (second half of the vector array)
};

hipError_t launch_pack_tips(hipStream_t st, const Geometry &g, uint32_t *vec, const uint8_t *codes, int n_taxa,
                            int n_patterns, const int32_t *site2ptn, int n_sites, int datatype,
                            const uint32_t *tip_slots);
// cntp[tile][slot]: per-tile mutation counts of the recomputed vectors, folded by launch_cntsum into cnt[slot]
hipError_t launch_newview(hipStream_t st, const Geometry &g, uint32_t *vec, const NvOp *ops, int n_ops, uint32_t *cntp,
                          uint32_t nslots);
// chores a refresh launch does for the scan launch behind it on the stream: kids[kid_upd[3i]] = (kid_upd[3i+1], kid_upd[3i+2])
// (chained kernel only) and zero_ptr[0..zero_words) = 0
struct RefreshExtra { const uint32_t *kid_upd = nullptr; int n_kid_upd = 0; uint2 *kids = nullptr; uint32_t *zero_ptr = nullptr; uint32_t zero_words = 0;
                      uint32_t *cnt_host = nullptr; /* pinned host mirror of cnt[]: written by the in-kernel fold */
                      const int32_t *n_lev_ptr = nullptr; /* the schedule was made on the device (launch_sched): level count read from here */
                      // a device-planned sweep (launch_sched's descriptors): the DFS programs of its scan parts are written by EXTRA
                      // workgroups of the refresh launch (launch_newview_levels adds them; they also clear the scan's outputs), on CUs
                      // the refresh leaves idle -- no plan kernel on the critical path
                      const uint2 *wp_kids = nullptr; uint32_t wp_n = 0; const WalkDesc *wp_desc = nullptr; const uint32_t *wp_hdr = nullptr /* {parts, candidates} */;
                      void *wp_prog = nullptr; uint32_t *wp_out = nullptr; uint32_t wp_max_parts = 0;
                      uint32_t *shadow = nullptr; /* k_newview_wgq: the word-major copy of every vector written (Geometry::shoff) */
                      int waves_hint = 0; /* launch_newview_levels: waves per workgroup (0 = sixteen); narrow levels need fewer */ };
// Refresh schedule of a COMPLETE tree made on the device from the topology array alone (kids[cid], cids n .. n + n_ops - 1 are
// the inner records): ops in level order, lev_off[0 .. n_lev] as launch_newview_levels reads them, *n_lev.  One workgroup;
// trees of up to kSchedMaxSlots vectors.  Order inside a level is not defined (ops of a level are independent).
constexpr uint32_t kSchedMaxSlots = 16384;
// ... and, by a second workgroup of the same launch, the scan descriptors of a whole sweep (what Engine::plan_walk lays out on the
// host: per prune node the parts of its two neighbourhoods, their candidate counts N(q, maxtrav) and output offsets), radius <= 6
struct SweepDescArgs {
  const uint32_t *nodep = nullptr;   // cid of every prune record in sweep order; nullptr: no descriptors wanted
  uint32_t n_prune = 0, maxtrav = 0, split_cands = 0;
  WalkDesc *desc = nullptr;          // out: one per scan part, as launch_walk_plan / launch_scan_prog read them
  uint2 *parts = nullptr;            // out: (output offset, candidates) per part, for launch_part_min
  uint32_t *part_node = nullptr;     // out, pinned host memory: index of the prune node behind every part
  uint32_t *hdr_host = nullptr;      // out, pinned host memory: {parts, candidates, 0, flag raised behind everything}
  uint32_t *hdr_dev = nullptr;       // out, device memory: {parts, candidates} for the kernels that follow
};
// kids (and sw.nodep) may be PINNED HOST memory: the two workgroups read the 8 bytes per vector over the bus themselves (no copy
// dispatch in front of the launch); kids_copy != nullptr: the schedule workgroup leaves a copy in device memory for the kernels
// that follow
hipError_t launch_sched(hipStream_t st, const uint2 *kids, uint32_t n_taxa, uint32_t n_ops, NvOp *ops, int32_t *lev_off, int32_t *n_lev,
                        const SweepDescArgs &sw = SweepDescArgs(), uint2 *kids_copy = nullptr);
// every level in ONE launch (one 16-wave workgroup per tile, workgroup barrier between levels)
// Fitch mode also folds the per-tile counts into cnt[dst] (last workgroup; `done` = a zeroed device word, left zeroed);
// weighted mode leaves that to launch_cntsum
hipError_t launch_newview_levels(hipStream_t st, const Geometry &g, uint32_t *vec, const NvOp *ops, const int32_t *lev_off,
                                 int n_lev, uint32_t *cntp, uint32_t nslots, uint32_t *cnt, uint32_t *done,
                                 const RefreshExtra &x = RefreshExtra());
// the same refresh cut into chains (Fitch mode): ops laid out per (level, wave), wl_off[16 * n_lev + 1]; NvOp::a = 0xFFFFFFFF
// takes the previous op's result from registers
hipError_t launch_newview_chains(hipStream_t st, const Geometry &g, uint32_t *vec, const NvOp *ops, const int32_t *wl_off,
                                 int n_lev, int n_ops, uint32_t *cntp, uint32_t nslots, uint32_t *cnt, uint32_t *done,
                                 const RefreshExtra &x);
// pmin[i] = min(out[parts[i].x .. parts[i].x + parts[i].y)); pmin may be pinned host memory.  cnt_host != nullptr: cnt[0 .. n_cnt)
// is copied there as well; done != nullptr (a zeroed device word, left zeroed): pmin[n_parts] = 1 once everything is written
hipError_t launch_part_min(hipStream_t st, const uint32_t *out, const uint2 *parts, int n_parts, uint32_t *pmin,
                           const uint32_t *cnt = nullptr, uint32_t *cnt_host = nullptr, uint32_t n_cnt = 0, uint32_t *done = nullptr);
hipError_t launch_cntsum(hipStream_t st, const Geometry &g, const NvOp *ops, int n_ops, const uint32_t *cntp,
                         uint32_t nslots, uint32_t *cnt, int tiles = 0 /* 0 = tiles_for(g) */,
                         uint32_t *cnt_host = nullptr /* pinned host mirror of cnt[] */);
int tiles_for(const Geometry &g);
int newview_tile(const Geometry &g);
int tiles_for_levels(const Geometry &g);      // tiles (rows of cntp) launch_newview_levels uses: 32-word tiles for one word per lane
hipError_t launch_evaluate(hipStream_t st, const Geometry &g, const uint32_t *vec, const EvOp *ops, int n_ops,
                           uint32_t *out);
hipError_t launch_scan(hipStream_t st, const Geometry &g, const uint32_t *vec, const ScanHdr *hdr, int n_scans,
                       const ScanOp *ops, uint32_t *out, int max_depth,
                       uint32_t *host_out = nullptr, uint32_t n_out = 0, uint32_t *done = nullptr,   // as launch_scan_walk
                       // weighted mode, online UFBoot: vals[out index][npat] = per-pattern lengths of every tentative tree
                       // (16 bits each), *vmax = the largest of them (atomic max)
                       uint16_t *vals = nullptr, uint32_t npat = 0, uint32_t *vmax = nullptr);
hipError_t launch_scan_walk(hipStream_t st, const Geometry &g, const uint32_t *vec, const uint2 *kids, int n_taxa,
                            const WalkDesc *desc, int n_scans, uint32_t *out, uint32_t *ncand, int max_depth,
                            uint32_t *masks = nullptr, uint2 *info = nullptr,   // masks != nullptr: UFBoot variant (ufboot.hip)
                            // host_out != nullptr: the last workgroup copies out[0..n_out) to pinned host memory (done: zeroed word)
                            uint32_t *host_out = nullptr, uint32_t n_out = 0, uint32_t *done = nullptr,
                            bool word_major = false /* read the vectors from the word-major copy (Geometry::shoff): the caller knows it is current */);
// planned-program scan (radius <= 6, DNA): launch_walk_plan turns the descriptors into one DFS program per (scan part, gap end)
// -- WalkDesc::pad1 must hold the number of candidates behind the FIRST gap end (xa) of the part --, launch_scan_prog runs
// them with the children's vectors requested one expansion ahead.  Same outputs as launch_scan_walk.
hipError_t launch_walk_plan(hipStream_t st, const uint2 *kids, int n_taxa, const WalkDesc *desc, int n_scans, void *prog,
                            uint32_t *zero_ptr = nullptr, uint32_t zero_words = 0);   // zero_ptr: cleared by the same launch (the scan's outputs)
size_t scan_prog_bytes(int n_scans);
bool scan_prog_supported(const Geometry &g, int max_depth);
hipError_t launch_scan_prog(hipStream_t st, const Geometry &g, const uint32_t *vec, const WalkDesc *desc, int n_scans,
                            const void *prog, uint32_t *out, uint32_t *ncand, uint32_t *host_out = nullptr, uint32_t n_out = 0,
                            uint32_t *done = nullptr,
                            unsigned long long *trace = nullptr /* diagnostic: 4 words per workgroup (begin, end on the 100 MHz clock, where, what) */,
                            bool word_major = false /* read the vectors from the word-major copy (Geometry::shoff; the caller knows it is current) */);
size_t scan_prog_blocks(const Geometry &g, int n_scans);
// per-pattern Fitch lengths: ops = the (a, b) joins of a rooted traversal of the current tree; `planes` is
// scratch of site_planes_words() words; ptn_out[p] = length of pattern p (0 where first_site[p] < 0)
hipError_t launch_site_counts(hipStream_t st, const Geometry &g, const uint32_t *vec, const EvOp *ops, int n_ops,
                              uint32_t *planes, const int32_t *ptn_first_site, int n_ptn, uint16_t *ptn_out);
size_t site_planes_words(const Geometry &g, int n_ops);

// Sankoff: per-pattern cost of the branch (a, b): ptn[j] = min_x(A[x] + min_y(cost[x][y] + B[y]))
hipError_t launch_sankoff_pattern(hipStream_t st, const Geometry &g, const uint32_t *vec, uint32_t a, uint32_t b,
                                  uint16_t *ptn_out, uint32_t *vmax = nullptr);
hipError_t launch_pack_tips_sankoff(hipStream_t st, const Geometry &g, uint32_t *vec, const uint8_t *codes, int n_taxa,
                                    int n_patterns, const int32_t *inf_index, int n_inf, int datatype);

}  // namespace mpf
