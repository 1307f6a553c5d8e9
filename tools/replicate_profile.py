import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from mpboot_amd import engine, synth, bootstrap
cfg = synth.WORKLOADS["C3"]; letters, names = synth.workload("C3"); codes = synth.letters_to_codes(letters, cfg["alphabet"])
e = engine.FitchEngine(codes); n, P = codes.shape
e.seed_ties(engine.TIE_RANDOM, 1); e.make_parsimony_tree(12345, 0); back = e.get_tree()
B = 64
samples = np.random.default_rng(4242).multinomial(P, np.ones(P) / P, size=B).astype(np.uint16)
e.ufboot_attach(samples); e.set_tree(back); e.reset_node_order(); e.seed_ties(engine.TIE_RANDOM, 1); e.optimize_spr(1, 6)
_l, _c, bt = e.ufboot_state(); trees = [e.ufboot_tree(int(t)) for t in bt]; e.ufboot_detach()
e.set_option("timing", 2)
for b in range(3):
    e.set_weights(samples[b].astype(np.int32)); e.seed_ties(1, 5 + b); e.reset_node_order(); e.set_tree(trees[b]); e.optimize_spr(1, 6)
tw = tc = 0.0
mv = []
e.reset_stats()
for b in range(3, 23):
    t0 = time.perf_counter(); e.set_weights(samples[b].astype(np.int32)); t1 = time.perf_counter()
    e.seed_ties(1, 5 + b); e.reset_node_order(); e.set_tree(trees[b]); s = e.optimize_spr(1, 6); t2 = time.perf_counter()
    tw += t1 - t0; tc += t2 - t1; mv.append(len(e.moves()[0]))
st = e.stats()
print(f"per replicate: set_weights {tw/20*1e3:.2f} ms, climb {tc/20*1e3:.2f} ms; moves {st['moves_applied']/20:.1f} tests {st['insertion_tests']/20:.0f} scan launches {st['scan_launches']/20:.1f} view launches {st['view_launches']/20:.1f}")
print('moves per replicate', mv)
print({k: round(v/20, 3) for k, v in st.items() if k.endswith('ms_total')})
