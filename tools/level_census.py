#!/usr/bin/env python3
"""Census of the levels of a levelized DP graph: python3 tools/level_census.py graph.dpg
How many transitions are 'pure copies' (every destination vertex has exactly one in-edge, of weight 0, and neither level
carries a colour): their sweep is a permutation of the state and their back-pointers are all (0, 0)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from dipgenie_amd import capi
g = capi.DpGraphArrays.load(sys.argv[1])
lo = np.asarray(g.level_off); L = g.n_levels; nV = g.n_vertices
indeg = np.bincount(g.out_dst, minlength=nV)
src = np.repeat(np.arange(nV), np.diff(g.out_off))
w_in = np.zeros(nV, np.int64); np.add.at(w_in, g.out_dst, g.out_w.astype(np.int64))
col = (np.diff(g.hom_off) + np.diff(g.het_off)) > 0
lvl_of = np.repeat(np.arange(L), np.diff(lo))
max_in = np.zeros(L, np.int64); np.maximum.at(max_in, lvl_of, indeg)
min_in = np.full(L, 1 << 30, np.int64); np.minimum.at(min_in, lvl_of, indeg)
w_lvl = np.zeros(L, np.int64); np.add.at(w_lvl, lvl_of, w_in)
c_lvl = np.zeros(L, np.int64); np.add.at(c_lvl, lvl_of, col.astype(np.int64))
width = np.diff(lo)
pure = (max_in[1:] == 1) & (min_in[1:] == 1) & (w_lvl[1:] == 0) & (c_lvl[1:] == 0) & (c_lvl[:-1] == 0)
same_w = width[1:] == width[:-1]
print(f"levels {L}, transitions {L - 1}")
print(f"pure-copy transitions (in-degree 1 everywhere, weight 0, no colour on either level): {int(pure.sum())} = {100 * pure.mean():.1f} %  (of which width unchanged: {int((pure & same_w).sum())})")
runs = np.diff(np.flatnonzero(np.diff(np.concatenate([[0], pure.astype(np.int8), [0]]))))[::2]
if runs.size: print(f"runs of consecutive pure-copy transitions: {runs.size}, mean length {runs.mean():.2f}, max {runs.max()}; a collapsed chain would have {L - 1 - int(pure.sum()) + runs.size} launches")
nocol = (c_lvl[1:] == 0) & (c_lvl[:-1] == 0)
print(f"colourless transitions: {100 * nocol.mean():.1f} %; max in-degree 1: {100 * (max_in[1:] == 1).mean():.1f} %; weight-free: {100 * (w_lvl[1:] == 0).mean():.1f} %")
