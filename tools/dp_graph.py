#!/usr/bin/env python3
"""hipGraph replay of the level chain: python tools/dp_graph.py graph.dpg batch1 batch2 ...  (first pass captures, later passes replay)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dipgenie_amd import capi
ctx = capi.Context(0)
g = capi.DpGraphArrays.load(sys.argv[1])
ctx.dp_load_graph(g)
ref = None
for v in sys.argv[2:]:
    ctx.dp_set_option("graph_batch", int(v))
    row = []
    for it in range(4):
        t0 = time.perf_counter(); out = ctx.dp_run(); wall = 1e3 * (time.perf_counter() - t0); tm = ctx.dp_timing()
        if ref is None: ref = out.key()
        assert out.key() == ref
        row.append(f"{tm.forward_ms:.1f}/{wall:.0f}")
    print(f"graph_batch={v}: forward/wall ms per pass: {'  '.join(row)}  ({1e3*tm.forward_ms/(g.n_levels-1):.2f} us/level steady)", flush=True)
