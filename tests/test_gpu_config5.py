"""-m gpu: BASELINE configs[4] (chr22-style linear panel, 100 walks, R = 32) at rehearsal scale.  No reference answer exists at
the full 50 Mbp (the reference would need days), so the path is pinned by properties at a size the oracle still finishes:
the drop-in CLI on a 100-walk panel, its levelized graph solved again through the C ABI resident / in forced segments
(checkpoint + recompute, the mode of the full size) / with plain launches -- one answer, equal to the oracle's, with every
level digest equal."""
import json
import os
import subprocess

import numpy as np
import pytest

import oracle_py as orc
from dipgenie_amd import capi, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def c5_panel(tmp_path_factory, built_hip):
    d = tmp_path_factory.mktemp("c5")
    segs, links, walks, reads = synth.linear_panel(22, backbone_bp=8_000, n_haps=100)
    synth.write_gfa(str(d / "c5.gfa"), segs, links, walks)
    synth.write_fasta(str(d / "c5.fa"), reads)
    out = {}
    for rep in range(2):                                   # two CLI runs: byte-identical output
        p = subprocess.run([built_hip, "-t8", "-p2", "-R32", "-g", str(d / "c5.gfa"), "-r", str(d / "c5.fa"), "-o", str(d / f"o{rep}.fa"),
                            "-J", str(d / f"o{rep}.json"), "-D", str(d / "c5")], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert p.returncode == 0, p.stderr.decode()[-2000:]
        out[rep] = (open(d / f"o{rep}.fa", "rb").read(), json.load(open(d / f"o{rep}.json")))
    assert out[0][0] == out[1][0]
    return capi.DpGraphArrays.load(str(d / "c5.dpg")), out[0][1]


def test_config5_panel_one_answer_in_every_execution_mode(c5_panel, gpu_ctx):
    g, summ = c5_panel
    assert g.R == 32 and g.n_levels > 200
    ref = orc.dp_solve(g, want_digest=True)
    assert ref["value"] == summ["dp_value"]
    modes = [{}, {"segment_cells": max(1, int(ref["cells"]) // 7)}, {"segment_cells": max(1, int(ref["cells"]) // 23), "plane_limit": 0},
             {"segment_cells": max(1, int(ref["cells"]) // 23), "graph_batch": 0}, {"graph_batch": 0}, {"lattice_chunk_cells": max(2, int(ref["cells"]) // 5)}, {"fast": 0}]
    try:
        gpu_ctx.dp_set_option("digest", 1)
        for m in modes:
            for k, v in m.items():
                gpu_ctx.dp_set_option(k, v)
            out = gpu_ctx.dp_solve(g)
            assert (out.value, out.s_het, out.p1, out.p2) == (ref["value"], ref["s_het"], ref["p1"], ref["p2"]), m
            assert np.array_equal(gpu_ctx.dp_level_digest(g.n_levels)[1:], ref["digest"][1:]), m
            for k in m:
                gpu_ctx.dp_set_option(k, {"segment_cells": 0, "graph_batch": -1, "lattice_chunk_cells": 1 << 32, "fast": 1, "plane_limit": 1}[k])
    finally:
        for k, v in {"digest": 0, "segment_cells": 0, "graph_batch": -1, "lattice_chunk_cells": 1 << 32, "fast": 1, "plane_limit": 1}.items():
            gpu_ctx.dp_set_option(k, v)


def test_config5_5mbp_tier_naturally_segmented(built_hip, tmp_path_factory):
    """BASELINE configs[4] at a tenth of its size: 5 Mbp backbone x 100 walks, R = 32 -- 7.1 x 10^11 cells, a 1.4 TB back-pointer
    lattice that does NOT fit HBM, so checkpoint + recompute, delta windows and ~5.5 x 10^5 launches run as they do at full size
    (nothing forced).  No reference answer exists (the reference would need a day), so size-independent properties: the lattice is
    segmented; a different segmentation (smaller chunks) gives the same FASTA and value; the walked path re-scores to the DP value
    in every run (dg_dp_run fails otherwise); the value is monotone in the recombination budget."""
    d = tmp_path_factory.mktemp("c5_5m")
    segs, links, walks, reads = synth.linear_panel(22, backbone_bp=5_000_000, n_haps=100)
    synth.write_gfa(str(d / "c5.gfa"), segs, links, walks)
    synth.write_fasta(str(d / "c5.fa"), reads)
    del segs, links, walks, reads

    def run(tag, R, options=None):
        env = dict(os.environ)
        if options:
            env["DG_DP_OPTIONS"] = options
        p = subprocess.run([built_hip, "-t16", "-p2", f"-R{R}", "-g", str(d / "c5.gfa"), "-r", str(d / "c5.fa"), "-o", str(d / f"{tag}.fa"), "-J", str(d / f"{tag}.json")],
                           env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        assert p.returncode == 0, p.stderr.decode()[-2000:]
        return open(d / f"{tag}.fa", "rb").read(), json.load(open(d / f"{tag}.json"))

    fa, s32 = run("r32", 32)
    assert s32["dp_segments"] > 1 and s32["dp_chunks"] > s32["dp_segments"], s32          # beyond HBM: checkpoint + recompute
    assert s32["cells"] > 5e11 and s32["n_levels"] > 2e5 and s32["dp_value"] > 0 and s32["r1"] <= 32 and s32["r2"] <= 32
    assert s32["len1"] > 4_500_000 and s32["len2"] > 4_500_000
    fb, s32b = run("r32b", 32, "lattice_chunk_cells=%d,plane_limit=0" % (3 << 30))         # 6 GB chunks: other chunk and segment boundaries; every plane re-swept
    assert fb == fa and s32b["dp_value"] == s32["dp_value"] and (s32b["dp_chunks"], s32b["dp_segments"]) != (s32["dp_chunks"], s32["dp_segments"])
    assert s32["dp_traceback_ms"] < 0.85 * s32b["dp_traceback_ms"], (s32["dp_traceback_ms"], s32b["dp_traceback_ms"])   # the plane-limited second pass is the shorter one
    _, s28 = run("r28", 28)
    _, s36 = run("r36", 36)
    assert s28["dp_value"] <= s32["dp_value"] <= s36["dp_value"], (s28["dp_value"], s32["dp_value"], s36["dp_value"])
    assert s28["cells"] * 33 == s32["cells"] * 29                                          # same graph, (R + 1) planes
