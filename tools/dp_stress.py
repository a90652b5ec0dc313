#!/usr/bin/env python3
"""Randomised GPU-vs-oracle parity sweep beyond the fixed test seeds: python tools/dp_stress.py [n_graphs] [seed0]
Varies widths, level counts, R, weight-1 density, colour density (hom-only / het-only / mixed / none), list lengths, and runs
every graph plain and with the alternative execution modes (graph batches, no look-ahead, segments, delta windows)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import graphgen, oracle_py as orc
from dipgenie_amd import capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 990000
ctx = capi.Context(0)
rng = np.random.default_rng(seed0)
modes = [{}, {"graph_batch": 5}, {"graph_batch": 0}, {"warm_ahead": 0}, {"segment_cells": 3000}, {"segment_cells": 900, "plane_limit": 0}, {"segment_cells": 700}, {"delta_cap_entries": 500}, {"adaptive_rc": 0}, {"coop": 2}, {"rowx": 0},
         {"lean_chain": 0}, {"l2_prefetch": 0}, {"delta_overlap": 2}, {"pf_far": 0}, {"host_tables": 1}]
defaults = {"plane_limit": 1, "graph_batch": -1, "warm_ahead": 128, "segment_cells": 0, "delta_cap_entries": 0, "adaptive_rc": 1, "coop": 1, "rowx": 1, "lean_chain": 1, "l2_prefetch": 6, "delta_overlap": 1, "pf_far": 128, "host_tables": 0}
bad = 0
for t in range(n):
    kw = dict(n_levels=int(rng.integers(2, 120)), max_width=int(rng.choice([3, 8, 20, 45, 70])), R=int(rng.choice([0, 1, 3, 6, 18, 33])),
              p_w1=float(rng.choice([0.0, 0.2, 0.6])), p_colour=float(rng.choice([0.0, 0.05, 0.3, 0.7, 1.0])), n_colours=int(rng.choice([2, 12, 200])),
              max_list=int(rng.choice([1, 4, 12])), extra_edges=float(rng.choice([0.3, 1.5, 4.0])), min_width=int(rng.choice([1, 1, 3])))
    g = graphgen.random_levelized(seed0 + t, **kw)
    kind = t % 4                                       # colour mix: 0 as generated, 1 hom only, 2 het only, 3 none
    if kind == 1: g.het_off = np.zeros_like(g.het_off); g.het_col = g.het_col[:0]
    if kind == 2: g.hom_off = np.zeros_like(g.hom_off); g.hom_col = g.hom_col[:0]
    if kind == 3: g.het_off = np.zeros_like(g.het_off); g.het_col = g.het_col[:0]; g.hom_off = np.zeros_like(g.hom_off); g.hom_col = g.hom_col[:0]
    ref = orc.dp_solve(g, want_digest=True)
    mode = modes[t % len(modes)]
    for k, v in mode.items(): ctx.dp_set_option(k, v)
    ctx.dp_set_option("digest", 1)
    out = ctx.dp_solve(g)
    dg = ctx.dp_level_digest(g.n_levels)
    ok = (out.value, out.s_het, out.p1, out.p2, out.cells, out.relaxations) == (ref["value"], ref["s_het"], ref["p1"], ref["p2"], ref["cells"], ref["relaxations"]) \
        and np.array_equal(dg[1:], ref["digest"][1:])
    again = ctx.dp_run()
    ok = ok and again.key() == out.key()
    ctx.dp_set_option("digest", 0)
    for k in mode: ctx.dp_set_option(k, defaults[k])
    if t % 100 == 99: print(f"... {t + 1} graphs, {bad} mismatches so far", flush=True)
    if not ok:
        bad += 1
        print("MISMATCH", t, kw, mode, kind, flush=True)
print(f"{n} graphs, {bad} mismatches", flush=True)
sys.exit(1 if bad else 0)
