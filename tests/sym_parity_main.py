"""Parity of the symmetric sweep form against the oracle (run by tests/test_gpu_parity.py::test_dp_symmetric_form_measurement_build in a
process of its own with DG_LIB = the measurement build bin/libdipgenie_hip_sym.so: capi binds one library per process)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)
import graphgen
import oracle_py as orc
from dipgenie_amd import capi

MODES = {"rc4": {"sym_rc": 4}, "rc1": {"sym_rc": 1}, "rc2": {"sym_rc": 2}, "rc3": {"sym_rc": 3}, "rc6": {"sym_rc": 6}, "rc8": {"sym_rc": 8},
         "rows_inline": {"coop": 0}, "plain_launches": {"graph_batch": 0}, "host_tables": {"host_tables": 1}, "no_rowx": {"rowx": 0},
         "folded_grid": {"sym_fold": 1}, "folded_rc2": {"sym_fold": 1, "sym_rc": 2}, "folded_rc8_inline": {"sym_fold": 1, "sym_rc": 8, "coop": 0}}
SHAPES = [dict(), dict(max_width=30, n_levels=40, R=6), dict(max_width=3, n_levels=200, R=2), dict(R=0), dict(p_w1=0.9, R=18),
          dict(p_colour=0.0), dict(p_colour=1.0, max_list=9, n_colours=10), dict(max_width=70, n_levels=10, R=4, extra_edges=3.0),
          dict(min_width=1, max_width=1, n_levels=30, R=3), dict(max_width=12, n_levels=300, R=5, p_colour=0.1),
          dict(max_width=40, n_levels=25, R=33, p_w1=0.5), dict(n_levels=2, R=2),
          dict(max_width=30, n_levels=60, R=18, p_w1=0.3, p_colour=0.5), dict(max_width=60, n_levels=30, R=32, p_w1=0.6),
          dict(max_width=64, n_levels=30, R=3, p_w1=0.5, p_colour=0.8), dict(max_width=3, n_levels=8, R=2, extra_edges=100.0),
          dict(min_width=20, max_width=24, n_levels=8, R=2, extra_edges=70.0), dict(min_width=15, max_width=18, n_levels=40, R=7, p_colour=0.6),
          dict(min_width=63, max_width=66, n_levels=12, R=9, p_w1=0.4, p_colour=0.3), dict(min_width=120, max_width=200, n_levels=6, R=5, p_colour=0.3, extra_edges=0.5),
          dict(min_width=40, max_width=90, n_levels=10, R=4, p_colour=0.3, extra_edges=0.05), dict(min_width=100, max_width=140, n_levels=5, R=3, extra_edges=0.02)]


def fan_in_graph(k):
    """source -> k vertices -> ONE vertex (in-degree k, alternating weights) -> sink, with colours (giant column AND fan-in row)"""
    level_off = np.array([0, 1, 1 + k, 2 + k, 3 + k], np.int32)
    out, w, out_off = [], [], [0]
    for j in range(k):
        out.append(1 + j); w.append(0)
    out_off.append(len(out))
    for j in range(k):
        out.append(1 + k); w.append(j & 1)
        out_off.append(len(out))
    out.append(2 + k); w.append(0); out_off.append(len(out))
    out_off.append(len(out))
    nV = 3 + k
    hom = [[] for _ in range(nV)]; het = [[] for _ in range(nV)]
    for j in range(k):
        het[1 + j] = [j % 7]; hom[1 + j] = [10 + j % 5]
    hom[1 + k] = [10, 12]; het[1 + k] = [3]

    def csr(ls):
        off = np.zeros(nV + 1, np.int64); off[1:] = np.cumsum([len(x) for x in ls])
        return off, np.array([c for x in ls for c in x], np.int32)
    ho, hc = csr(hom); to, tc = csr(het)
    return capi.DpGraphArrays(2, level_off=level_off, out_off=np.array(out_off, np.int64), out_dst=np.array(out, np.int32), out_w=np.array(w, np.uint8),
                              hom_off=ho, hom_col=hc, het_off=to, het_col=tc)


def both(ctx, g, tag):
    out = ctx.dp_solve(g)
    ref = orc.dp_solve(g, want_digest=True)
    assert (out.value, out.s_het, out.p1, out.p2) == (ref["value"], ref["s_het"], ref["p1"], ref["p2"]), tag
    assert np.array_equal(ctx.dp_level_digest(g.n_levels)[1:], ref["digest"][1:]), tag


ctx = capi.Context(0)
ctx.dp_set_option("sym", 2)
ctx.dp_set_option("digest", 1)
n = 0
for mode, opts in sorted(MODES.items()):
    for k, v in opts.items():
        ctx.dp_set_option(k, v)
    for q, kw in enumerate(SHAPES):
        both(ctx, graphgen.random_levelized(9700 + q, **kw), (mode, q))
        assert any("sym" in k for k in ctx.dp_launch_profile()), (mode, q)
        n += 1
    for k in (90, 200):
        both(ctx, fan_in_graph(k), (mode, "fan-in", k))
        n += 1
    for k, v in {"sym_rc": 4, "coop": 1, "graph_batch": -1, "host_tables": 0, "rowx": 1, "sym_fold": 0}.items():
        ctx.dp_set_option(k, v)
print("symmetric form parity ok", n, "graph x mode cases")
