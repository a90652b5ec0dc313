#!/usr/bin/env python3
"""Perf probe: load .dpg graphs, run the HIP DP in its modes, print timings.
usage: python tools/dp_perf.py [--check] [--modes=fast,norowx,generic] graph.dpg ... """
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from dipgenie_amd import capi

check = "--check" in sys.argv
modes = "fast"
for a in sys.argv[1:]:
    if a.startswith("--modes="): modes = a.split("=")[1]
paths = [a for a in sys.argv[1:] if not a.startswith("--")]
ctx = capi.Context(0)
OPT = {"fast": (1, 1), "norowx": (0, 1), "generic": (1, 0)}     # (rowx, fast)
for p in paths:
    g = capi.DpGraphArrays.load(p)
    t0 = time.time(); ctx.dp_load_graph(g); t1 = time.time()
    print(f"== {os.path.basename(p)}: L={g.n_levels} V={g.n_vertices} load {t1-t0:.2f}s", flush=True)
    ref = None
    for mode in modes.split(","):
        ctx.dp_set_option("rowx", OPT[mode][0]); ctx.dp_set_option("fast", OPT[mode][1])
        ctx.dp_load_graph(g)                               # rowx takes effect at load
        for it in range(3):
            t0 = time.time(); out = ctx.dp_run(); dt = time.time() - t0
            tm = ctx.dp_timing()
            print(f"  {mode:7s} it={it}: value {out.value} s_het {out.s_het} r1 {len(out.p1)-1} r2 {len(out.p2)-1} | delta {tm.delta_ms:.1f} fwd {tm.forward_ms:.1f} "
                  f"tb {tm.traceback_ms:.1f} total {tm.total_ms:.1f} ms wall {dt*1e3:.1f} ms launches {tm.n_forward_launches} | {out.cells/ (tm.forward_ms/1e3)/1e9:.2f} Gcells/s "
                  f"{1e3*tm.forward_ms/(g.n_levels-1):.2f} us/level", flush=True)
            if ref is None: ref = out.key()
            assert out.key() == ref, "MODE MISMATCH"
    if check:
        import oracle_py as orc
        ctx.dp_set_option("digest", 1)
        out = ctx.dp_run(); dg = ctx.dp_level_digest(g.n_levels)
        ctx.dp_set_option("digest", 0)
        t0 = time.time(); r = orc.dp_solve(g, want_digest=True); print(f"  oracle {time.time()-t0:.1f}s")
        ok = (out.value, out.s_het, out.p1, out.p2) == (r["value"], r["s_het"], r["p1"], r["p2"]) and np.array_equal(dg[1:], r["digest"][1:]) and ref == out.key()
        print("  PARITY", "OK" if ok else "FAIL", flush=True)
