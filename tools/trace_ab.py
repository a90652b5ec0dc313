#!/usr/bin/env python3
"""Traceback walk A/B on a dumped levelized graph: general walk / lean walk / two-ahead walk (lean_chain 0 / 1 / 2), a few passes each,
answers compared.   usage: python tools/trace_ab.py graph.dpg [passes]      (bash tools/mhc24_dpg.sh writes /tmp/c/mhc24.dpg)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dipgenie_amd import capi
g = capi.DpGraphArrays.load(sys.argv[1])
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ctx = capi.Context(0)
ref = None
for mode, name in ((1, "lean walk"), (2, "two-ahead walk"), (0, "general walk"), (2, "two-ahead walk"), (1, "lean walk")):
    ctx.dp_set_option("lean_chain", mode)
    ctx.dp_load_graph(g)
    tb, fw = [], []
    for _ in range(passes):
        out = ctx.dp_run()
        t = ctx.dp_timing()
        tb.append(t.traceback_ms); fw.append(t.forward_ms)
    if ref is None: ref = out.key()
    print(f"lean_chain={mode} ({name}): traceback ms {' '.join(f'{x:.2f}' for x in tb)} | forward {min(fw):.1f} | answer {'same' if out.key() == ref else 'DIFFERENT'}", flush=True)
