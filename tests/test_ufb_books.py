"""The deferred half of the online UFBoot bookkeeping -- boot_trees, reference counts, stored topologies, the topology map -- and
the second host thread that works it off during a pipelined climb (mpboot_amd/host/ufb_books.hpp, the code libmpfitch.so runs),
WITHOUT a GPU and under the thread sanitizer: tests/cpu/ufb_books_test.cpp replays the stream a GPU run recorded
(tests/golden/ufb_stream.bin.gz: what eight pipelined climbs of four freshly attached trackers handed their worker, 1142 batches,
tools/record_ufb_stream.py) inline, through the
worker, and through a worker that is joined and restarted in mid-climb -- all must end in the state the GPU run recorded --,
then random streams the same three ways."""
import os
import subprocess

import pytest

from helpers import GOLDEN, ROOT

SRC = os.path.join(ROOT, "tests", "cpu", "ufb_books_test.cpp")


@pytest.fixture(scope="module")
def prog(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("ufb_books") / "ufb_books_test")
    base = ["g++", "-std=c++17", "-O1", "-g", "-pthread", SRC, "-o", out]
    r = subprocess.run(base[:5] + ["-fsanitize=thread"] + base[5:], capture_output=True, text=True)
    tsan = r.returncode == 0
    if not tsan:                                   # (a toolchain without the sanitizer's runtime: the comparison still runs)
        r = subprocess.run(base, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return out, tsan


def _run(prog, *args):
    out, _tsan = prog
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 exitcode=66")
    r = subprocess.run([out, *args], capture_output=True, text=True, env=env, timeout=600)
    assert "ThreadSanitizer" not in r.stderr, r.stderr[-3000:]
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    return r.stdout


def test_sanitizer_is_in_the_build(prog):
    assert prog[1], "g++ -fsanitize=thread did not build here"


def test_recorded_stream_replays_to_the_recorded_state(prog):
    import gzip
    import shutil
    path = os.path.join(os.path.dirname(prog[0]), "ufb_stream.bin")
    with gzip.open(os.path.join(GOLDEN, "ufb_stream.bin.gz"), "rb") as src, open(path, "wb") as dst:
        shutil.copyfileobj(src, dst)
    out = _run(prog, "replay", path)
    assert "equal" in out and "DIFFERENT" not in out
    n_batches = int(out.split(" batches")[0].split()[-1])
    assert n_batches >= 1000, out


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5])
def test_random_streams_inline_equal_worker(prog, seed):
    out = _run(prog, "random", str(seed), "300")
    assert "equal" in out and "DIFFERENT" not in out


def test_canonical_form_names_topologies_not_numberings(prog):
    """books::canonical_topology (the key of the topology map, treels' tree "string" of iqtree.cpp:3689-3707): equal under any
    numbering of the inner nodes and rotation of their records, different for different sets of bipartitions"""
    out = _run(prog, "canon", "11", "3000")
    assert "3000 relabelled pairs equal" in out
