"""CPU-side checks of the drop-in boundary: the library loads and exports what include/mpfitch.h declares."""
import ctypes
import os
import re

import pytest

from helpers import ROOT


def _declared():
    with open(os.path.join(ROOT, "include", "mpfitch.h")) as f:
        src = f.read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mpf_[a-z_]+)\s*\(", src)))


def test_header_declares_entry_points():
    names = _declared()
    for must in ("mpf_engine_create", "mpf_optimize_spr", "mpf_make_parsimony_tree", "mpf_score_tree", "mpf_spr_scan"):
        assert must in names


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    lib_path = os.path.join(ROOT, "mpboot_amd", "libmpfitch.so")
    if not os.path.exists(lib_path):
        g.build()
    lib = ctypes.CDLL(lib_path)
    for name in _declared():
        assert hasattr(lib, name), name
    lib.mpf_abi_version.restype = ctypes.c_int
    assert lib.mpf_abi_version() == 8


def test_python_binding_lists_the_same_symbols():
    from mpboot_amd import engine
    assert sorted(engine.EXPORTS) == _declared()


def test_no_gpu_means_loud_failure():
    """Without a device the product must refuse to run (no CPU fallback)."""
    import numpy as np
    from mpboot_amd import engine
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        pytest.skip("GPU present")
    codes = np.ones((4, 8), dtype=np.uint8)
    with pytest.raises(engine.MpfError) as ei:
        engine.FitchEngine(codes)
    assert ei.value.code == -1


def test_product_never_imports_the_oracle():
    for dirpath, _dirs, files in os.walk(os.path.join(ROOT, "mpboot_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")):
                with open(os.path.join(dirpath, f)) as fh:
                    txt = fh.read()
                assert "pyoracle" not in txt and "fitch_oracle" not in txt and "liboracle" not in txt, f


def test_iqtree_state_encoding_matches_reference_tip_codes():
    """Alignment::convertState codes -> PLL tip codes, against the codes the reference's own PLL parser produced
    (fixture 'codes', dumped from tr->yVector by oracle/_ref/pll_ref_driver)."""
    import numpy as np
    from helpers import load_fixture
    from mpboot_amd import engine
    from oracle import iqtree_fitch
    for name, alpha, dt in (("dna_clean", "DNA", engine.DNA), ("dna_ambig", "DNA", engine.DNA), ("aa", "AA", engine.AA)):
        fx = load_fixture(name)
        states = iqtree_fitch.convert_states(fx["rows"], alpha)
        codes = engine.encode_iqtree_states(states, dt)
        assert (codes == fx["codes_np"]).all(), name
    with pytest.raises(engine.MpfError):
        engine.encode_iqtree_states(np.array([[19]], dtype=np.int8), engine.DNA)


def test_slow_iqtree_fitch_agrees_with_pinned_oracle():
    import numpy as np
    from helpers import load_fixture
    from oracle import iqtree_fitch, pyoracle as po
    for name, alpha, ns in (("dna_ambig", "DNA", 4), ("aa", "AA", 20)):
        fx = load_fixture(name)
        states = iqtree_fitch.convert_states(fx["rows"], alpha)
        o = po.Oracle(fx["codes_np"], fx["weights_np"], datatype=fx["datatype"])
        o.enable_persite(True)
        for t in fx["trees"][:3]:
            back = np.array(t["back"], dtype=np.int32)
            score, ptn = iqtree_fitch.compute_parsimony(states, fx["weights"], back, ns)
            assert score == t["score"] == o.score_tree(back)
            optn, _ = o.pattern_scores()
            assert (ptn == optn).all()


def test_vectorised_lcg64_matches_sprng_fixture_and_scalar():
    import json
    import numpy as np
    from helpers import GOLDEN
    from mpboot_amd.rng import Lcg64
    from oracle import pyoracle as po
    with open(os.path.join(GOLDEN, "sprng_lcg64.json")) as f:
        ref = json.load(f)
    for seed, vals in ref.items():
        got = Lcg64(int(seed)).doubles(len(vals))
        assert [float.fromhex(v) for v in vals] == got.tolist()
    g = Lcg64(77)
    a = np.concatenate([g.doubles(5), g.doubles(1000), g.doubles(3)])
    assert a.tolist() == po.lcg64_doubles(77, 1008)


def test_bootstrap_weights_resample_sites():
    import numpy as np
    from mpboot_amd.bootstrap import bootstrap_weights
    from mpboot_amd.rng import Lcg64
    w = np.array([1, 0, 3, 2, 1], dtype=np.int32)
    b = bootstrap_weights(w, Lcg64(5))
    assert b.sum() == w.sum() and b[1] == 0
    # the same draws, one at a time (alignment.cpp:1981-1990)
    g = Lcg64(5)
    site_pattern = np.repeat(np.arange(5), w)
    exp = np.zeros(5, dtype=np.int64)
    for d in g.doubles(int(w.sum())):
        exp[site_pattern[int(np.floor(d * w.sum()))]] += 1
    assert b.tolist() == exp.tolist()


def test_tie_stream_skip_ahead_equals_drawing_one_by_one():
    """mpf_tie_state_after (pure host arithmetic in libmpfitch.so: the k-step jump of the lcg64 tie stream the batched refinement
    uses) against the oracle's restated generator stepped draw by draw, which is pinned against SPRNG's own output"""
    import ctypes
    import numpy as np
    from mpboot_amd import engine
    from oracle import pyoracle as po
    lib = engine.load_library()
    o = po.Oracle(np.array([[1, 2], [2, 1], [1, 1], [2, 2]], dtype=np.uint8))
    for seed in (0, 1, 77, 2 ** 31 - 1):
        o.seed_ties(po.TIE_RANDOM, seed)
        s0 = o.tie_state()
        state, drawn = s0, 0
        for k in (0, 1, 2, 3, 63, 64, 1000, 12345):
            # the restated stream, k more draws: state = state * A + c, k times (oracle/rng.h)
            for _ in range(k):
                state = (state * 0x27bb2ee687b0b0fd + 3037000493) & (2 ** 64 - 1)
            drawn += k
            assert lib.mpf_tie_state_after(ctypes.c_uint64(s0), ctypes.c_uint64(drawn)) == state
        assert lib.mpf_tie_state_after(ctypes.c_uint64(s0), ctypes.c_uint64(2 ** 40)) == \
            lib.mpf_tie_state_after(ctypes.c_uint64(lib.mpf_tie_state_after(ctypes.c_uint64(s0), ctypes.c_uint64(2 ** 39))), ctypes.c_uint64(2 ** 39))
