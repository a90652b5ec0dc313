#!/usr/bin/env python3
"""Sweep a dg_dp_set_option over values: python tools/dp_tune.py key v1,v2,... graph.dpg ..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dipgenie_amd import capi
key, vals, paths = sys.argv[1], [int(v) for v in sys.argv[2].split(",")], sys.argv[3:]
ctx = capi.Context(0)
for p in paths:
    g = capi.DpGraphArrays.load(p); ctx.dp_load_graph(g)
    ref = None
    for v in vals:
        ctx.dp_set_option(key, v)
        best = 1e9
        for _ in range(3):
            out = ctx.dp_run(); tm = ctx.dp_timing(); best = min(best, tm.forward_ms)
        if ref is None: ref = out.key()
        assert out.key() == ref
        print(f"{os.path.basename(p)} {key}={v}: fwd {best:.1f} ms ({1e3*best/(g.n_levels-1):.2f} us/level)", flush=True)
