#!/usr/bin/env python3
"""Try option settings (k=v,k=v;...) on several .dpg files: python tools/dp_opt_grid.py "a=1,b=2;a=3" g1.dpg g2.dpg"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dipgenie_amd import capi
ctx = capi.Context(0)
settings = [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in st.split(",") if kv) for st in sys.argv[1].split(";")]
for path in sys.argv[2:]:
    g = capi.DpGraphArrays.load(path)
    ctx.dp_load_graph(g)
    ref = None
    for st in settings:
        for k, v in st.items(): ctx.dp_set_option(k, v)
        best, best_tb, best_tot = 1e30, 1e30, 1e30
        for it in range(3):
            out = ctx.dp_run(); tm = ctx.dp_timing()
            if ref is None: ref = out.key()
            assert out.key() == ref
            best = min(best, tm.forward_ms); best_tb = min(best_tb, tm.traceback_ms); best_tot = min(best_tot, tm.total_ms)
        print(f"{os.path.basename(path):16s} {st}: forward {best:.1f} ms ({1e3*best/(g.n_levels-1):.2f} us/level), traceback {best_tb:.1f} ms, pass {best_tot:.1f} ms", flush=True)
