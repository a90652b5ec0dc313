#!/usr/bin/env python3
"""Per-level geometry of a levelized DP graph, as the sweep sees it: python3 tools/level_geometry.py graph.dpg
Width classes of the destination level with launches, cells, in-edges per level, in-degree statistics, colour share."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from dipgenie_amd import capi
g = capi.DpGraphArrays.load(sys.argv[1])
lo = np.asarray(g.level_off, np.int64); L = g.n_levels; nV = g.n_vertices; RP = g.R + 1
indeg = np.bincount(np.asarray(g.out_dst), minlength=nV).astype(np.int64)
T = np.add.reduceat(indeg, lo[:-1])
dmax = np.maximum.reduceat(indeg, lo[:-1])
heavy = np.add.reduceat((indeg > 8).astype(np.int64), lo[:-1])
col = ((np.diff(np.asarray(g.hom_off)) + np.diff(np.asarray(g.het_off))) > 0).astype(np.int64)
c_lvl = np.add.reduceat(col, lo[:-1])
k = np.diff(lo)
cells = k[1:] ** 2 * RP; pairs = T[1:] ** 2
print(f"levels {L}, vertices {nV}, R {g.R}; cells {cells.sum() / 1e9:.2f} G, edge pairs x RP {(pairs * RP).sum() / 1e9:.2f} G; relaxations per cell {(pairs.sum()) / (k[1:] ** 2).sum():.3f}")
edges = [0, 16, 32, 64, 128, 192, 256, 384, 512, 1024, 1 << 20]
print("k2 class        levels    share of cells   mean k2   mean T   T/k2   mean dmax   heavy rows/level   coloured")
for a, b in zip(edges[:-1], edges[1:]):
    m = (k[1:] >= a) & (k[1:] < b)
    if not m.any(): continue
    coloured = ((c_lvl[1:] + c_lvl[:-1]) > 0)[m].mean()
    print(f"[{a:5d},{b:7d}) {int(m.sum()):9d}   {cells[m].sum() / cells.sum():10.4f}   {k[1:][m].mean():9.1f} {T[1:][m].mean():8.1f} {T[1:][m].sum() / k[1:][m].sum():6.2f} {dmax[1:][m].mean():9.1f} {heavy[1:][m].mean():14.2f} {coloured:14.3f}")
