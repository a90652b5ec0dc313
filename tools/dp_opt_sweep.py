#!/usr/bin/env python3
"""Sweep one dg_dp_set_option over values on a .dpg: python tools/dp_opt_sweep.py graph.dpg option v1 v2 ..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dipgenie_amd import capi
ctx = capi.Context(0)
g = capi.DpGraphArrays.load(sys.argv[1])
ctx.dp_load_graph(g)
ref = None
for v in sys.argv[3:]:
    ctx.dp_set_option(sys.argv[2], int(v))
    best = None
    for it in range(3):
        out = ctx.dp_run(); tm = ctx.dp_timing()
        if ref is None: ref = out.key()
        assert out.key() == ref
        if best is None or tm.forward_ms < best[0]: best = (tm.forward_ms, tm.traceback_ms, tm.delta_ms)
    print(f"{sys.argv[2]}={v}: forward {best[0]:.1f} ms traceback {best[1]:.1f} ms delta {best[2]:.1f} ms ({1e3*best[0]/(g.n_levels-1):.2f} us/level)", flush=True)
