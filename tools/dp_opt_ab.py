#!/usr/bin/env python3
"""A/B of dg_dp_set_option sets on a dumped levelized graph: per option set, DP timings of a few passes, the launch profile, and -- with
every level's digest collected -- equality of value, s_het, edge lists and all digests with the first set (the reference).
usage: python tools/dp_opt_ab.py graph.dpg "k=v,k=v" ["k=v,k=v" ...]      (e.g. "symmetric=0" "symmetric=1"; options named in any set
are put back to the value they have in the FIRST set before the next set runs; load-time options reload the graph)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from dipgenie_amd import capi

g = capi.DpGraphArrays.load(sys.argv[1])
sets = [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in spec.split(",") if kv) for spec in sys.argv[2:]] or [{}]
base = {}
for o in sets:
    for k in o:
        base.setdefault(k, sets[0].get(k))
passes = int(os.environ.get("AB_PASSES", "3"))
ctx = capi.Context(0)
ref = None
for spec, opts in zip(sys.argv[2:] or [""], sets):
    for k, v in base.items():
        if v is not None: ctx.dp_set_option(k, v)
    for k, v in opts.items():
        ctx.dp_set_option(k, v)
    ctx.dp_load_graph(g)
    ctx.dp_set_option("digest", 1)
    out = ctx.dp_run()
    dg = ctx.dp_level_digest(g.n_levels)
    ctx.dp_set_option("digest", 0)
    key = (out.key(), dg.tobytes())
    if ref is None:
        ref = key
    fw, tb = [], []
    for it in range(passes):
        out2 = ctx.dp_run()
        tm = ctx.dp_timing()
        fw.append(tm.forward_ms); tb.append(tm.traceback_ms)
        assert out2.key() == out.key()
    prof = ctx.dp_launch_profile()
    print(f"{spec:40s} parity {'OK' if key == ref else 'MISMATCH'}  value {out.value}  forward {min(fw):9.2f} ms  traceback {min(tb):7.2f} ms  launches {tm.n_forward_launches}  "
          f"{1e3 * min(fw) / max(tm.n_forward_launches, 1):.3f} us/launch  segments {tm.n_segments}", flush=True)
    print("    " + " ".join(f"{k}:{v}" for k, v in sorted(prof.items())), flush=True)
