#!/usr/bin/env python3
"""Run the HIP DP once on a .dpg (for rocprofv3). usage: dp_once.py graph.dpg [graph_mode] [n_runs]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dipgenie_amd import capi
ctx = capi.Context(0)
g = capi.DpGraphArrays.load(sys.argv[1])
ctx.dp_set_option("fast", int(sys.argv[2]) if len(sys.argv) > 2 else 1)
if os.environ.get("DG_SYNC_EVERY"): ctx.dp_set_option("sync_every", int(os.environ["DG_SYNC_EVERY"]))
if os.environ.get("DG_OPTS"):
    for kv in os.environ["DG_OPTS"].split(","):
        k_, v_ = kv.split("="); ctx.dp_set_option(k_, int(v_))
if os.environ.get("DG_GRAPH_BATCH"): ctx.dp_set_option("graph_batch", int(os.environ["DG_GRAPH_BATCH"]))
ctx.dp_load_graph(g)
if os.environ.get("DG_DUMP_MAPS"):                      # tools/pmc_abort_probe.sh: the memory map every later crash address is resolved against
    open(os.environ["DG_DUMP_MAPS"], "w").write(open("/proc/self/maps").read())
for it in range(int(sys.argv[3]) if len(sys.argv) > 3 else 1):
    out = ctx.dp_run()
    tm = ctx.dp_timing()
    print("pass", it, "value", out.value, "delta_ms", round(tm.delta_ms, 2), "fwd_ms", round(tm.forward_ms, 1), "tb_ms", round(tm.traceback_ms, 1), "total_ms", round(tm.total_ms, 1), flush=True)
