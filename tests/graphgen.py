"""Random levelized DP graphs (dg_dp_graph layout) for kernel parity tests -- test infrastructure."""
import numpy as np

from dipgenie_amd.capi import DpGraphArrays


def random_levelized(seed, n_levels=12, max_width=9, R=3, p_w1=0.3, p_colour=0.4, n_colours=12, max_list=4,
                     extra_edges=1.5, dup_edges=True, min_width=1):
    """Level 0 = {source}, last level = {sink}. Every vertex gets >=1 out-edge (except sink) and edges only
    go to the next level. Parallel edges carry equal weights (the product's documented precondition)."""
    rng = np.random.default_rng(seed)
    widths = [1] + [int(rng.integers(min_width, max_width + 1)) for _ in range(n_levels - 2)] + [1]
    level_off = np.zeros(n_levels + 1, np.int32)
    level_off[1:] = np.cumsum(widths)
    nV = int(level_off[-1])
    out = [[] for _ in range(nV)]
    for l in range(n_levels - 1):
        a0, k, b0, k2 = level_off[l], widths[l], level_off[l + 1], widths[l + 1]
        wmap = {}
        for i in range(k):
            n_e = 1 + int(rng.poisson(extra_edges))
            for _ in range(n_e):
                j = int(rng.integers(0, k2))
                w = wmap.setdefault((i, j), int(rng.random() < p_w1))
                out[a0 + i].append((b0 + j, w))
                if dup_edges and rng.random() < 0.1:
                    out[a0 + i].append((b0 + j, w))
        # make sure every next-level vertex is reachable from someone, most of the time
        for j in range(k2):
            if rng.random() < 0.9 and not any(d == b0 + j for i in range(k) for (d, _) in out[a0 + i]):
                i = int(rng.integers(0, k))
                w = wmap.setdefault((i, j), int(rng.random() < p_w1))
                out[a0 + i].append((b0 + j, w))
    out_off = np.zeros(nV + 1, np.int64)
    out_off[1:] = np.cumsum([len(o) for o in out])
    out_dst = np.array([d for o in out for (d, _) in o], np.int32)
    out_w = np.array([w for o in out for (_, w) in o], np.uint8)
    hom, het = [], []
    for v in range(nV):
        def lst():
            if rng.random() < p_colour:
                n = int(rng.integers(1, max_list + 1))
                return sorted(set(int(x) for x in rng.integers(0, n_colours, n)))
            return []
        a, b = lst(), lst()
        b = [c + n_colours for c in b]   # HOM and HET colour ids are disjoint in the product
        hom.append(a)
        het.append(b)
    hom_off = np.zeros(nV + 1, np.int64)
    het_off = np.zeros(nV + 1, np.int64)
    hom_off[1:] = np.cumsum([len(x) for x in hom])
    het_off[1:] = np.cumsum([len(x) for x in het])
    return DpGraphArrays(R, level_off=level_off, out_off=out_off, out_dst=out_dst, out_w=out_w,
                         hom_off=hom_off, hom_col=np.array([c for x in hom for c in x], np.int32),
                         het_off=het_off, het_col=np.array([c for x in het for c in x], np.int32))


def random_topological(seed, n=60, R=4, avg_deg=2.0, p_w1=0.4, p_zero_colour=0.5, max_colours=4, span=8, dup=0.15):
    """A DAG whose vertex ids are a topological order (every edge u -> v has u < v), as ExpandedGraph::topologically_reorder
    leaves it for the haploid DP: edges reach up to `span` ids ahead (several depth levels), parallel edges may carry
    DIFFERENT weights (the scatter loop's arrival order then matters), many vertices carry no colour (ties at value 0).
    Returns (out_off, out_dst, out_w, n_colours)."""
    rng = np.random.default_rng(seed)
    out = [[] for _ in range(n)]
    for u in range(n - 1):
        for _ in range(1 + int(rng.poisson(avg_deg - 1))):
            v = int(min(n - 1, u + 1 + rng.integers(0, span)))
            w = int(rng.random() < p_w1)
            out[u].append((v, w))
            if rng.random() < dup:
                out[u].append((v, int(rng.random() < 0.5)))      # parallel edge, independent weight
    out_off = np.zeros(n + 1, np.int64)
    out_off[1:] = np.cumsum([len(o) for o in out])
    out_dst = np.array([v for o in out for (v, _) in o], np.int32)
    out_w = np.array([w for o in out for (_, w) in o], np.uint8)
    ncol = np.where(rng.random(n) < p_zero_colour, 0, rng.integers(1, max_colours + 1, n)).astype(np.int32)
    return out_off, out_dst, out_w, ncol
