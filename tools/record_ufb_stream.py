#!/usr/bin/env python3
"""Records what pipelined tracked climbs hand their log worker (MPF_UFB_RECORD, host/ufboot.cpp) -- the fixture
tests/test_ufb_books.py replays on the CPU under the thread sanitizer.   python tools/record_ufb_stream.py <out.bin> [taxa] [patterns] [samples] [climbs]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out = os.path.abspath(sys.argv[1])
n, P, B, climbs = (int(x) for x in (sys.argv[2:6] + ["36", "700", "20", "6"][len(sys.argv) - 2:]))
if os.path.exists(out):
    os.remove(out)
os.environ["MPF_UFB_RECORD"] = out
from mpboot_amd import engine, synth, trees
letters, _ = synth.synth_alignment(n, P, "DNA", 0.08, seed=5)
codes = synth.letters_to_codes(letters, "DNA")
e = engine.FitchEngine(codes)
rng = np.random.default_rng(3)
samples = rng.multinomial(codes.shape[1], np.ones(codes.shape[1]) / codes.shape[1], size=B).astype(np.uint16)
e.ufboot_attach(samples, 0.5)
e.seed_ties(engine.TIE_RANDOM, 11)
for c in range(climbs):
    if c % 2 == 0 and c:
        # (a tracker that has seen a few climbs books little: most of a run's deferred work is in its first climbs)
        e.ufboot_detach()
        samples = rng.multinomial(codes.shape[1], np.ones(codes.shape[1]) / codes.shape[1], size=B).astype(np.uint16)
        e.ufboot_attach(samples, 0.5)
    e.set_option("scan_batch", [3, 64, 1, 8][c % 4])     # (small batches: many jobs, hand-overs in quick succession)
    e.set_tree(trees.random_topology(n, np.random.default_rng(100 + c)))
    e.reset_node_order()
    s = e.optimize_spr(1, 6)
    print("climb", c, "->", s, "saved trees", len(e.ufboot_tree_logl()), flush=True)
print(out, os.path.getsize(out), "bytes")
