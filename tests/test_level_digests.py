"""CPU: the oracle pinned to the REFERENCE cell by cell.  tests/golden/level_digests.json holds, for every diploid golden case, the
digest of dp_cur after every roll (approximator.cpp:706) and the sink's two weighted-edge lists (:757-764, :781-782), printed by
the reference itself: an instrumented copy built by tests/golden/make_level_digests.py with oracle/Makefile's rules (the
instrumentation is ours; no reference text is kept here).  The oracle, run on the levelized graph that OUR host pipeline builds
from the same files, must give the same digest on every level -- value and winning predecessor pair (the :657-659 tie-break) of
every reachable cell, also of the cells off the answer path -- and the same edge lists.  The GPU parity tests require HIP ==
oracle on these digests, so they stand on the reference's own cells."""
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

import oracle_py as orc
from dipgenie_amd import capi

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
FIX = json.load(open(os.path.join(HERE, "golden", "level_digests.json")))
SMALL = [n for n, c in FIX.items() if "digests" in c]


def compare(g, fx):
    ref = orc.dp_solve(g, want_digest=True)
    assert g.n_levels == fx["n_levels"]
    assert (ref["value"], ref["s_het"]) == (fx["dp_value"], fx["s_het"])
    assert ref["p1"] == [tuple(e) for e in fx["p1"]] and ref["p2"] == [tuple(e) for e in fx["p2"]]
    d = np.asarray(ref["digest"], np.uint64)[1:]
    if "digests" in fx:
        want = np.array([int(x, 16) for x in fx["digests"]], np.uint64)
        bad = np.flatnonzero(d != want)
        assert bad.size == 0, f"first differing level: {int(bad[0]) + 1}"
    else:
        assert [f"{int(x):016x}" for x in d[:8]] == fx["first8"] and [f"{int(x):016x}" for x in d[-8:]] == fx["last8"]
        assert hashlib.sha256(d.astype("<u8").tobytes()).hexdigest() == fx["digests_sha256"]


@pytest.mark.parametrize("name", SMALL)
def test_oracle_cells_equal_reference_cells(name, built_cpu, tmp_path):
    fx = FIX[name]
    subprocess.run([built_cpu, "-q", "-t2", *fx["args"], "-X", "-D", str(tmp_path / "g"), "-g", os.path.join(ROOT, fx["gfa"]), "-r", os.path.join(ROOT, fx["reads"]),
                    "-o", str(tmp_path / "o.fa")], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    compare(capi.DpGraphArrays.load(str(tmp_path / "g.dpg")), fx)


def test_oracle_cells_equal_reference_cells_mhc4():
    """full-size MHC_4 (BASELINE configs[1] graph with the CHM13 reads): 120,362 levels, 421 M cells; the levelized graph is the
    committed dump tests/data/mhc4.dpg (written by the host pipeline; tests/test_gpu_parity.py re-derives it from the GFA on the
    GPU box and compares every digest of the HIP path with the oracle's)"""
    compare(capi.DpGraphArrays.load(os.path.join(HERE, "data", "mhc4.dpg")), FIX["mhc4_p2"])
