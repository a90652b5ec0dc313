#!/usr/bin/env python3
"""A/B of the chained dispatches on a dumped levelized graph: per option set, DP timings of a few passes, the launch profile, and -- with
every level's digest collected -- equality of value, s_het, edge lists and all digests with the unchained run.
usage: python tools/chain_ab.py graph.dpg ["k=v,k=v" ...]      (first option set: chain=0 is always run as the reference)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from dipgenie_amd import capi

g = capi.DpGraphArrays.load(sys.argv[1])
sets = sys.argv[2:] or ["chain=0", "chain=1"]                  # the first set is the reference
ctx = capi.Context(0)
ctx.dp_load_graph(g)
ref = None
for spec in sets:
    opts = dict(kv.split("=") for kv in spec.split(","))
    for k, v in opts.items():
        ctx.dp_set_option(k, int(v))
    dbg = "chain_dbg" in opts                                    # (timing experiments with parts switched off: results are void)
    if not dbg:
        ctx.dp_set_option("digest", 1)
        out = ctx.dp_run()
        dg = ctx.dp_level_digest(g.n_levels)
        ctx.dp_set_option("digest", 0)
        key = (out.key(), dg.tobytes())
    else:
        key = ref
    if ref is None:
        ref = key
    same = key == ref or "chain_dbg" in opts
    best = None
    for it in range(3):
        try:
            t0 = time.time(); out2 = ctx.dp_run(); dt = time.time() - t0
        except capi.DgError:
            out2 = None
        tm = ctx.dp_timing()
        best = tm.forward_ms if best is None else min(best, tm.forward_ms)
        assert dbg or out2.key() == out.key()
    prof = ctx.dp_launch_profile()
    print(f"{spec:32s} parity {'OK' if same else 'MISMATCH'}  forward {best:8.2f} ms  traceback {tm.traceback_ms:6.2f} ms  launches {tm.n_forward_launches}  "
          f"{1e3 * best / (g.n_levels - 1):.3f} us/level  chain dispatches {sum(v for k, v in prof.items() if 'chain' in k)}", flush=True)
    for k, v in opts.items():                                    # back to the defaults for the next set
        ctx.dp_set_option(k, {"chain": 0, "chain_rc": 2, "chain_max": 15, "chain_dbg": 0}.get(k, int(v)))
