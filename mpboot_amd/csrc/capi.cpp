// capi.cpp -- the extern "C" boundary declared in include/mpfitch.h.
#include <cstring>
#include <new>

#include "engine.hpp"

namespace mpf { const std::string &last_error(); }

struct mpf_engine {
  mpf::Engine eng;
};

using mpf::set_error;

#define NEED(e)                                                                                                        \
  do {                                                                                                                 \
    if (!(e)) { set_error("null engine handle"); return MPF_E_INVALID; }                                               \
    if ((e)->eng.broken()) { set_error("engine unusable: a device launch did not come back (destroy it)"); return MPF_E_STATE; } \
    (e)->eng.activate();                                                                                               \
  } while (0)

extern "C" {

const char *mpf_last_error(void) { return mpf::last_error().c_str(); }
int mpf_abi_version(void) { return MPF_ABI_VERSION; }

int mpf_engine_create(mpf_engine **out, const mpf_config *cfg, const uint8_t *codes, const int32_t *weights)
{
  if (!out || !cfg) { set_error("mpf_engine_create: null argument"); return MPF_E_INVALID; }
  *out = nullptr;
  mpf_engine *e = new (std::nothrow) mpf_engine();
  if (!e) { set_error("out of host memory"); return MPF_E_NOMEM; }
  int rc = e->eng.init(*cfg, codes, weights);
  if (rc != MPF_OK) { delete e; return rc; }
  *out = e;
  return MPF_OK;
}

int mpf_engine_create_sankoff(mpf_engine **out, const mpf_config *cfg, const uint8_t *codes, const int32_t *weights,
                              const uint32_t *cost)
{
  if (!out || !cfg || !cost) { set_error("mpf_engine_create_sankoff: null argument"); return MPF_E_INVALID; }
  *out = nullptr;
  mpf_engine *e = new (std::nothrow) mpf_engine();
  if (!e) { set_error("out of host memory"); return MPF_E_NOMEM; }
  int rc = e->eng.init(*cfg, codes, weights, cost);
  if (rc != MPF_OK) { delete e; return rc; }
  *out = e;
  return MPF_OK;
}

void mpf_engine_destroy(mpf_engine *e)
{
  if (e) e->eng.activate();
  delete e;
}

int mpf_set_weights(mpf_engine *e, const int32_t *weights)
{
  NEED(e);
  if (!weights) { set_error("null weights"); return MPF_E_INVALID; }
  return e->eng.set_weights(weights);
}

int mpf_get_geometry(const mpf_engine *e, int32_t *states, int32_t *words_per_row, int32_t *n_informative, int32_t *words_padded)
{
  NEED(e);
  if (states) *states = e->eng.S();
  if (words_per_row) *words_per_row = e->eng.Wref();
  if (n_informative) *n_informative = e->eng.n_informative();
  if (words_padded) *words_padded = e->eng.Wp();
  return MPF_OK;
}

int mpf_get_informative(const mpf_engine *e, int32_t *flags)
{
  NEED(e);
  const auto &v = e->eng.informative();
  std::memcpy(flags, v.data(), v.size() * sizeof(int32_t));
  return MPF_OK;
}

int mpf_get_tip_vector(mpf_engine *e, int32_t tip, uint32_t *out) { NEED(e); return e->eng.tip_vector(tip, out); }

int mpf_set_tree(mpf_engine *e, const int32_t *back)
{
  NEED(e);
  if (!back) { set_error("null topology"); return MPF_E_INVALID; }
  return e->eng.set_tree(back);
}

int mpf_get_tree(const mpf_engine *e, int32_t *back) { NEED(e); e->eng.get_tree(back); return MPF_OK; }
int mpf_reset_node_order(mpf_engine *e) { NEED(e); e->eng.reset_node_order(); return MPF_OK; }
int mpf_score_tree(mpf_engine *e, uint32_t *score) { NEED(e); return e->eng.score_tree(score); }

int mpf_score_trees(mpf_engine *e, int32_t n_trees, const int32_t *backs, uint32_t *scores)
{
  NEED(e);
  const size_t len = 3 * (size_t)(2 * e->eng.n() - 1);
  for (int t = 0; t < n_trees; t++) {
    int rc = e->eng.set_tree(backs + (size_t)t * len);
    if (rc) return rc;
    rc = e->eng.score_tree(scores + t);
    if (rc) return rc;
  }
  return MPF_OK;
}

int mpf_pattern_scores(mpf_engine *e, uint16_t *ptn_pars, int32_t *total) { NEED(e); return e->eng.pattern_scores(ptn_pars, total); }

int mpf_site_scores(mpf_engine *e, int32_t *site_pars, int32_t n_sites, int32_t *total)
{
  NEED(e);
  if (!site_pars || n_sites < 0) { set_error("mpf_site_scores: bad argument"); return MPF_E_INVALID; }
  return e->eng.site_scores(site_pars, n_sites, total);
}

int mpf_compute_parsimony(mpf_engine *e, const int32_t *back, uint32_t *score, uint16_t *pattern_pars)
{
  return mpf_compute_parsimony_at(e, back, 0, score, pattern_pars);
}

int mpf_compute_parsimony_at(mpf_engine *e, const int32_t *back, int32_t root_taxon, uint32_t *score, uint16_t *pattern_pars)
{
  NEED(e);
  if (back) { int rc = e->eng.set_tree(back); if (rc) return rc; }
  if (root_taxon < 0 || root_taxon > e->eng.n()) { set_error("mpf_compute_parsimony_at: root taxon out of range"); return MPF_E_INVALID; }
  // (the tree is evaluated at the edge of the leaf in start_; 0 keeps the engine's own -- tr->start = taxon 1)
  uint32_t s = 0;
  int rc = e->eng.score_tree(&s);            // (nodeRectifierPars puts start_ back on taxon 1: the guard goes behind it)
  if (rc) return rc;
  mpf::Engine::StartGuard guard(e->eng, root_taxon);
  if (root_taxon > 1) rc = e->eng.tree_length_at_start(&s);
  if (rc) return rc;
  if (score) *score = s;
  if (pattern_pars) {
    int32_t total = 0;
    rc = e->eng.pattern_scores(pattern_pars, &total);
    if (rc) return rc;
    if ((uint32_t)total != s) { set_error("per-pattern lengths do not add up to the tree length"); return MPF_E_STATE; }  // iqtree.cpp:3366-3367
  }
  return MPF_OK;
}

int mpf_encode_iqtree_states(int32_t datatype, const int8_t *states, int64_t count, uint8_t *codes)
{
  if (!states || !codes || count < 0) { set_error("mpf_encode_iqtree_states: null argument"); return MPF_E_INVALID; }
  for (int64_t i = 0; i < count; i++) {
    const int st = states[i];
    int code = -1;
    if (datatype == MPF_DNA) {
      if (st >= 0 && st < 4) code = 1 << st;
      else if (st == 18) code = 15;                       // STATE_UNKNOWN
      else if (st >= 4 && st <= 17) code = st - 3;        // ambiguity: mask + 3 (phylotree.cpp:1135-1137)
      if (code == 15 && st != 18) code = -1;              // 1+2+4+8+3 is not produced by convertState
    } else if (datatype == MPF_AA) {
      if (st >= 0 && st <= 22) code = st;                 // same numbering, B = 20, Z = 21, unknown = 22
    } else if (datatype == MPF_BIN) {
      if (st == 0 || st == 1) code = 1 << st;             // SEQ_BINARY: '0' -> 0, '1' -> 1 (alignment.cpp:846-851)
      else if (st == 2) code = 3;                         // STATE_UNKNOWN = num_states
    } else if (datatype == MPF_GENERIC) {
      if (st >= 0 && st <= 32) code = st;                 // SEQ_MORPH: symbol index, STATE_UNKNOWN = num_states <= 32 handed over as 32
    } else {
      set_error("mpf_encode_iqtree_states: unsupported data type");
      return MPF_E_UNSUPPORTED;
    }
    if (code <= 0 && !((datatype == MPF_AA || datatype == MPF_GENERIC) && code == 0)) { set_error("state outside the alphabet (STATE_INVALID)"); return MPF_E_INVALID; }
    codes[i] = (uint8_t)code;
  }
  return MPF_OK;
}

int mpf_seed_ties(mpf_engine *e, int32_t tie_mode, int32_t seed)
{
  NEED(e);
  if (tie_mode != MPF_TIE_FIRST && tie_mode != MPF_TIE_RANDOM) { set_error("bad tie mode"); return MPF_E_INVALID; }
  e->eng.seed_ties(tie_mode, seed);
  return MPF_OK;
}

int mpf_set_rand_callback(mpf_engine *e, double (*fn)(void *), void *arg) { NEED(e); e->eng.set_rand(fn, arg); return MPF_OK; }

int mpf_set_tie_state(mpf_engine *e, uint64_t state)
{
  NEED(e);
  e->eng.set_rand(nullptr, nullptr);
  e->eng.set_tie_state(state);
  return MPF_OK;
}

uint64_t mpf_tie_state_after(uint64_t state, uint64_t n_draws) { return mpf::lcg64_skip(state, n_draws); }

int mpf_get_tie_state(const mpf_engine *e, uint64_t *state)
{
  NEED(e);
  if (!state) { set_error("null argument"); return MPF_E_INVALID; }
  *state = e->eng.tie_state();
  return MPF_OK;
}

int mpf_spr_scan(mpf_engine *e, int32_t rec, int32_t mintrav, int32_t maxtrav, int32_t cap, int32_t *q_recs,
                 uint32_t *mp, int32_t *n_p, int32_t *n_total)
{
  NEED(e);
  std::vector<int32_t> q;
  std::vector<uint32_t> m;
  int np = 0;
  int rc = e->eng.spr_scan(rec, mintrav, maxtrav, q, m, np);
  if (rc) return rc;
  if (n_total) *n_total = (int32_t)q.size();
  if (n_p) *n_p = np;
  if ((int)q.size() > cap) { set_error("mpf_spr_scan: output capacity too small"); return MPF_E_INVALID; }
  if (!q.empty()) {
    std::memcpy(q_recs, q.data(), q.size() * sizeof(int32_t));
    std::memcpy(mp, m.data(), m.size() * sizeof(uint32_t));
  }
  return MPF_OK;
}

int mpf_spr_sweep_scan(mpf_engine *e, int32_t mintrav, int32_t maxtrav, uint64_t *n_tests, uint32_t *min_mp)
{
  NEED(e);
  return e->eng.sweep_scan(mintrav, maxtrav, n_tests, min_mp);
}

int mpf_spr_sweep_costs(mpf_engine *e, int32_t mintrav, int32_t maxtrav, uint64_t cap, uint32_t *mp, uint64_t *offsets, uint64_t *n_tests)
{
  NEED(e);
  if (!n_tests || (cap && !mp)) { set_error("mpf_spr_sweep_costs: null output"); return MPF_E_INVALID; }
  return e->eng.sweep_costs(mintrav, maxtrav, cap, mp, offsets, n_tests);
}

int mpf_get_node_order(mpf_engine *e, int32_t *recs)
{
  NEED(e);
  if (!recs) { set_error("null output"); return MPF_E_INVALID; }
  return e->eng.node_order(recs);
}

int mpf_optimize_spr(mpf_engine *e, int32_t mintrav, int32_t maxtrav, uint32_t *score) { NEED(e); return e->eng.optimize_spr(mintrav, maxtrav, score); }

int mpf_optimize_spr_many(mpf_engine **engines, int32_t n_engines, int32_t mintrav, int32_t maxtrav, uint32_t *final_scores)
{
  if (n_engines < 0 || (n_engines && (!engines || !final_scores))) { set_error("mpf_optimize_spr_many: bad argument"); return MPF_E_INVALID; }
  std::vector<mpf::Engine *> es((size_t)n_engines);
  for (int k = 0; k < n_engines; k++) {
    if (!engines[k]) { set_error("mpf_optimize_spr_many: null engine"); return MPF_E_INVALID; }
    for (int j = 0; j < k; j++) if (engines[j] == engines[k]) { set_error("mpf_optimize_spr_many: an engine is listed twice"); return MPF_E_INVALID; }
    es[(size_t)k] = &engines[k]->eng;
  }
  return mpf::Engine::climb_many(es.data(), n_engines, mintrav, maxtrav, final_scores);
}

int mpf_optimize_spr_many_round(mpf_engine **engines, int32_t n_engines, int32_t mintrav, int32_t maxtrav, uint8_t *state, uint32_t *final_scores)
{
  if (n_engines < 0 || (n_engines && (!engines || !final_scores || !state))) { set_error("mpf_optimize_spr_many_round: bad argument"); return MPF_E_INVALID; }
  std::vector<mpf::Engine *> es((size_t)n_engines);
  for (int k = 0; k < n_engines; k++) {
    if (!engines[k]) { set_error("mpf_optimize_spr_many_round: null engine"); return MPF_E_INVALID; }
    es[(size_t)k] = &engines[k]->eng;
  }
  return mpf::Engine::climb_many_round(es.data(), n_engines, mintrav, maxtrav, state, final_scores);
}

int mpf_make_parsimony_tree(mpf_engine *e, int64_t seed, int32_t spr_dist, uint32_t *score)
{
  NEED(e);
  return e->eng.make_parsimony_tree(seed, spr_dist, score);
}

int mpf_stepwise_addition(mpf_engine *e, int64_t seed, uint32_t *best_per_step, int32_t *insert_per_step, uint32_t *score)
{
  NEED(e);
  return e->eng.stepwise_addition(seed, best_per_step, insert_per_step, score);
}

int mpf_get_moves(const mpf_engine *e, int32_t cap, int32_t *remove_rec, int32_t *insert_rec, uint32_t *score, int32_t *n_moves)
{
  NEED(e);
  const auto &mv = e->eng.moves();
  if (n_moves) *n_moves = (int32_t)mv.size();
  const size_t k = std::min((size_t)std::max(cap, 0), mv.size());
  for (size_t i = 0; i < k; i++) {
    if (remove_rec) remove_rec[i] = mv[i].remove_rec;
    if (insert_rec) insert_rec[i] = mv[i].insert_rec;
    if (score) score[i] = mv[i].score;
  }
  return MPF_OK;
}

int mpf_ufboot_attach(mpf_engine *e, int32_t n_samples, const uint16_t *samples, double epsilon)
{
  NEED(e);
  return e->eng.ufboot_attach(n_samples, samples, epsilon);
}
int mpf_ufboot_attach_sharded(mpf_engine *e, int32_t n_samples, int32_t n_local, const int32_t *sample_ids,
                              const uint16_t *samples_local, double epsilon, mpf_ufb_exchange_fn exchange, void *arg)
{
  NEED(e);
  if (n_local < 1 || n_local > n_samples || !sample_ids || !exchange) { set_error("mpf_ufboot_attach_sharded: bad argument"); return MPF_E_INVALID; }
  return e->eng.ufboot_attach(n_samples, samples_local, epsilon, n_local, sample_ids, exchange, arg);
}
int mpf_ufboot_detach(mpf_engine *e) { NEED(e); e->eng.ufboot_detach(); return MPF_OK; }
int mpf_ufboot_set_cutoff(mpf_engine *e, double logl_cutoff) { NEED(e); return e->eng.ufboot_set_cutoff(logl_cutoff); }
int mpf_ufboot_set_ratchet_booking(mpf_engine *e, int32_t on) { NEED(e); return e->eng.ufboot_set_ratchet_booking(on); }
int mpf_ufboot_set_mulhits(mpf_engine *e, int32_t on) { NEED(e); return e->eng.ufboot_set_mulhits(on); }
int mpf_ufboot_set_cutoff_from_btrees(mpf_engine *e, int32_t on) { NEED(e); return e->eng.ufboot_set_cutoff_from_btrees(on); }
int mpf_ufboot_get_orig_logl(const mpf_engine *e, int32_t *out)
{
  if (!e || !out) { set_error("mpf_ufboot_get_orig_logl: bad argument"); return MPF_E_INVALID; }
  e->eng.activate();
  return e->eng.ufboot_orig_logl(out);
}
int mpf_ufboot_set_store_trees(mpf_engine *e, int32_t on) { NEED(e); return e->eng.ufboot_set_store_trees(on); }
int mpf_ufboot_get_duplicates(const mpf_engine *e, uint64_t *n) { NEED(e); return e->eng.ufboot_duplicates(n); }
int mpf_ufboot_set_topboot(mpf_engine *e, int32_t n_top) { NEED(e); return e->eng.ufboot_set_topboot(n_top); }
int mpf_ufboot_get_sample_top(const mpf_engine *e, int32_t sample, int64_t *trees, int32_t *rell, int32_t cap, int32_t *n, int32_t *threshold)
{
  NEED(e);
  int k = 0;
  int rc = e->eng.ufboot_sample_top(sample, trees, rell, cap, &k, threshold);
  if (!rc && n) *n = k;
  return rc;
}
int mpf_ufboot_set_distinct_iter(mpf_engine *e, int32_t k) { NEED(e); return e->eng.ufboot_set_distinct_iter(k); }
int mpf_ufboot_set_iteration(mpf_engine *e, int32_t cur_it) { NEED(e); return e->eng.ufboot_set_iteration(cur_it); }
int mpf_ufboot_get_sample_iters(const mpf_engine *e, int32_t sample, int32_t *iters, int32_t cap, int32_t *n)
{
  NEED(e);
  int k = 0;
  int rc = e->eng.ufboot_sample_iters(sample, iters, cap, &k);
  if (!rc && n) *n = k;
  return rc;
}
int mpf_ufboot_get_sample_trees(const mpf_engine *e, int32_t sample, int64_t *out, int32_t cap, int32_t *n)
{
  NEED(e);
  int k = 0;
  int rc = e->eng.ufboot_sample_trees(sample, out, cap, &k);
  if (!rc && n) *n = k;
  return rc;
}
int mpf_ufboot_next_cutoff(const mpf_engine *e, int32_t percent, double *logl_cutoff)
{
  NEED(e);
  if (!logl_cutoff || percent < 0 || percent > 100) { set_error("mpf_ufboot_next_cutoff: bad argument"); return MPF_E_INVALID; }
  if (!e->eng.ufboot_attached()) { set_error("no UFBoot tracker attached"); return MPF_E_STATE; }
  *logl_cutoff = e->eng.ufboot_next_cutoff(percent);
  return MPF_OK;
}
int mpf_ufboot_num_trees(const mpf_engine *e, int64_t *n_trees)
{
  NEED(e);
  if (!n_trees) { set_error("null output"); return MPF_E_INVALID; }
  if (!e->eng.ufboot_attached()) { set_error("no UFBoot tracker attached"); return MPF_E_STATE; }
  *n_trees = e->eng.ufboot_n_trees();
  return MPF_OK;
}
int mpf_ufboot_tree_logl(const mpf_engine *e, double *out) { NEED(e); if (!out) { set_error("null output"); return MPF_E_INVALID; } return e->eng.ufboot_tree_logl(out); }
int mpf_ufboot_get_state(const mpf_engine *e, double *boot_logl, int32_t *boot_counts, int32_t *boot_trees)
{
  NEED(e);
  return e->eng.ufboot_state(boot_logl, boot_counts, boot_trees);
}
int mpf_ufboot_get_tree(const mpf_engine *e, int64_t tree_index, int32_t *back)
{
  NEED(e);
  if (!back) { set_error("null output"); return MPF_E_INVALID; }
  return e->eng.ufboot_tree(tree_index, back);
}
int mpf_ufboot_adopt(mpf_engine *e, int32_t n_updates, const int32_t *sample, const uint32_t *score, const int32_t *tree_of, int32_t n_trees,
                     const int32_t *backs, const uint32_t *lengths, int32_t *n_taken)
{
  NEED(e);
  return e->eng.ufboot_adopt(n_updates, sample, score, tree_of, n_trees, backs, lengths, n_taken);
}
int mpf_ufboot_refine_sweep(mpf_engine *e, int32_t maxtrav, const int32_t *tie_seeds, uint32_t *scores, uint8_t *stable, int32_t *first_move_visit)
{
  NEED(e);
  return e->eng.ufboot_refine_sweep(maxtrav, tie_seeds, scores, stable, first_move_visit);
}

int mpf_ufboot_get_counters(const mpf_engine *e, uint64_t *tie_draws, uint64_t *events, uint64_t *reps_rows, double *reps_kernel_ms)
{
  NEED(e);
  return e->eng.ufboot_counters(tie_draws, events, reps_rows, reps_kernel_ms);
}

int mpf_get_stats(const mpf_engine *e, mpf_stats *out) { NEED(e); *out = e->eng.stats; return MPF_OK; }
int mpf_reset_stats(mpf_engine *e) { NEED(e); e->eng.stats = mpf_stats{}; return MPF_OK; }

int mpf_set_option(mpf_engine *e, const char *key, int64_t value)
{
  NEED(e);
  if (!key) { set_error("null option key"); return MPF_E_INVALID; }
  return e->eng.set_option(key, value);
}
int mpf_get_scan_trace(mpf_engine *e, uint64_t *out, uint64_t cap, uint64_t *n_words)
{
  NEED(e);
  if (!n_words || (cap && !out)) { set_error("null output"); return MPF_E_INVALID; }
  return e->eng.scan_trace(out, cap, n_words);
}

int mpf_get_option(const mpf_engine *e, const char *key, int64_t *value)
{
  NEED(e);
  if (!key || !value) { set_error("null argument"); return MPF_E_INVALID; }
  return e->eng.get_option(key, value);
}

}  // extern "C"
