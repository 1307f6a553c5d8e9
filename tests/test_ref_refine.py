"""oracle/_ref/pll_ref_driver `refine` (the reference's PLL parsimony code timed on one bootstrap-refinement replicate,
bench.py's CPU side of the bootstrap metric) against the oracle: same re-weighted start length, same first-best climb.
Only where the reference build exists (this container); the GPU box gets the prebuilt driver with the snapshot."""
import os
import subprocess

import numpy as np
import pytest

from helpers import ROOT, load_fixture

DRV = os.path.join(ROOT, "oracle", "_ref", "pll_ref_driver")


@pytest.mark.skipif(not os.access(DRV, os.X_OK), reason="reference build (oracle/_ref) not present")
@pytest.mark.parametrize("name,seed", [("dna_48", 3), ("dna_ambig", 5), ("aa", 7)])
def test_refine_matches_oracle(tmp_path, name, seed):
    from mpboot_amd import synth, trees
    from oracle import pyoracle as po
    fx = load_fixture(name)
    codes = fx["codes_np"]
    n, P = codes.shape
    rng = np.random.default_rng(seed)
    w = rng.multinomial(P, np.ones(P) / P).astype(np.int32)
    back = np.array(fx["trees"][2]["back"], dtype=np.int32)
    names = [f"t{i + 1}" for i in range(n)]
    aln = tmp_path / "a.phy"
    table, skip = (synth._DNA_CODE, "U?NOX") if fx["datatype"] == 0 else (synth._AA_CODE, "?*X")
    inv = {v: k for k, v in table.items() if k not in skip}
    inv[15 if fx["datatype"] == 0 else 22] = "-"
    rows = ["".join(inv[int(c)] for c in row) for row in codes]
    synth.write_phylip(str(aln), rows, names)
    tf = tmp_path / "t.nwk"
    tf.write_text(trees.back_to_newick(back, names) + "\n")
    wf = tmp_path / "w.txt"
    wf.write_text(" ".join(map(str, w.tolist())) + "\n")
    out = subprocess.run([DRV, "refine", str(aln), "DNA" if fx["datatype"] == 0 else "WAG", "0", str(tf), "6", str(wf)],
                         capture_output=True, text=True, check=True, timeout=300).stdout
    tok = [l.split() for l in out.splitlines() if l.startswith("refined")][0]
    start, final, moves = int(tok[2]), int(tok[4]), int(tok[6])
    assert int([l.split() for l in out.splitlines() if l.startswith("final_check")][0][1]) == final
    o = po.Oracle(codes, w, datatype=fx["datatype"])
    assert o.score_tree(back) == start
    o.seed_ties(po.TIE_FIRST)
    o.set_pre_evaluate(0)                  # the PLL original, literally
    o.trace(True)
    assert o.optimize_spr(1, 6) == final
    assert len(o.get_moves()[0]) == moves
