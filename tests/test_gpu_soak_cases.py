"""Cases the randomised soak (tools/soak.py) found, kept as regressions: the later iterations of a -bb run (cut-offs, perturbed
trees, ratchet climbs) on the engine -- with and without the round-5 shortcuts -- against the oracle, snapshot by snapshot."""
import glob
import os
import sys

import numpy as np
import pytest

from helpers import GOLDEN, ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))

CASES = sorted(glob.glob(os.path.join(GOLDEN, "soak", "*.npz")))


@pytest.mark.gpu
@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[:-4] for p in CASES])
@pytest.mark.parametrize("opts", [{}, {"ufb_quiet": 0, "ufb_moot": 0, "ufb_memo": 0, "ufb_pipe": 0}], ids=["default", "plain"])
def test_soak_case_equals_the_oracle(path, opts):
    import soak_lib
    from mpboot_amd import engine, trees
    from oracle import pyoracle as po
    z = np.load(path)
    case = {k: z[k] for k in z.files}
    for k in ("tie", "seed", "radius", "iters", "it_seed"):
        case[k] = int(case[k])
    case["btrees"] = bool(case["btrees"])
    alpha = str(case["alpha"])
    codes = case["codes"]
    w = case["w0"] if bool(case["weighted"]) else None
    dt_e, dt_o = (engine.DNA, po.DNA) if alpha == "DNA" else (engine.AA, po.AA)
    e = engine.FitchEngine(codes, w, datatype=dt_e)
    for k, v in opts.items():
        e.set_option(k, v)
    got = soak_lib.later_iterations(e, engine.FitchEngine(codes, w, datatype=dt_e), trees, case, case["tie"])
    o = po.Oracle(codes, w, datatype=dt_o)
    want = soak_lib.later_iterations(o, engine.FitchEngine(codes, w, datatype=dt_e), trees, case, case["tie"])
    assert o.ufboot_bad() == 0
    assert soak_lib.first_difference(want, got) is None


def test_soak_cases_are_present():
    assert CASES
