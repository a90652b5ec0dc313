#!/usr/bin/env python3
"""Per-level view of a traced DP pass: joins rocprofv3's kernel trace (dispatch order = level order) with the level
geometry of the .dpg.
    rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 tools/dp_once.py graph.dpg
    python3 tools/level_profile.py graph.dpg OUT/**/*_kernel_trace.csv [summary.txt]
Prints launches / total / average microseconds grouped by kernel variant and by level class (largest in-degree,
cooperative or not, width bucket)."""
import csv, glob, re, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from dipgenie_amd.capi import DpGraphArrays

g = DpGraphArrays.load(sys.argv[1])
paths = [p for a in sys.argv[2:3] for p in glob.glob(a, recursive=True)]
out = open(sys.argv[3], "w") if len(sys.argv) > 3 else sys.stdout
rows = []
for p in paths:
    with open(p) as f:
        for r in csv.DictReader(f):
            if "dp_sweep" in r["Kernel_Name"]:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
L = g.n_levels
n_pass = len(rows) // (L - 1)
print(f"{len(rows)} sweep launches = {n_pass} pass(es) of {L - 1} levels", file=out)
rows = rows[-(L - 1):]                                   # last pass
dur = np.array([e - s for s, e, _ in rows]) / 1e3
gap = np.array([rows[i + 1][0] - rows[i][1] for i in range(len(rows) - 1)] + [0]) / 1e3
name = [re.sub(r"void dgi::|\(.*", "", n) for _, _, n in rows]
lo = g.level_off
nV = g.n_vertices
indeg = np.bincount(g.out_dst, minlength=nV)
lvl = np.repeat(np.arange(L), np.diff(lo))
maxin = np.zeros(L, int); np.maximum.at(maxin, lvl, indeg)
k = np.diff(lo)
T = np.zeros(L, int); np.add.at(T, lvl, indeg)
print(f"pass: {dur.sum() / 1e3:.1f} ms in kernels, {gap.sum() / 1e3:.1f} ms in gaps, {dur.mean():.3f} us per launch, {(dur.sum() + gap.sum()) / len(dur):.3f} us pitch", file=out)
def table(title, keys):
    print(f"\n== by {title}", file=out)
    agg = {}
    for i, kx in enumerate(keys):
        a = agg.setdefault(kx, [0, 0.0, 0.0]); a[0] += 1; a[1] += dur[i]; a[2] += gap[i]
    for kx in sorted(agg, key=lambda x: -agg[x][1]):
        n, d, gp = agg[kx]
        print(f"  {str(kx):60s} {n:8d} launches {d / 1e3:9.2f} ms {d / n:7.3f} us avg  gap {gp / n:6.3f}", file=out)
table("kernel variant", name)
def dcls(m): return "<=2" if m <= 2 else "3-8" if m <= 8 else "9-32" if m <= 32 else "33-64" if m <= 64 else ">64"
table("largest in-degree of the level", [dcls(maxin[l]) for l in range(1, L)])
def wcls(x): return "<32" if x < 32 else "<64" if x < 64 else "<128" if x < 128 else "<256" if x < 256 else ">=256"
table("level width k2", [wcls(k[l]) for l in range(1, L)])
table("in-degree class x variant", [dcls(maxin[l]) + " " + name[l - 1] for l in range(1, L)])
if len(sys.argv) > 4:                                   # per-level arrays for offline analysis
    variants = sorted(set(name))
    np.savez_compressed(sys.argv[4], dur_us=dur.astype(np.float32), gap_us=gap.astype(np.float32), variant=np.array([variants.index(n) for n in name], np.int16),
                        variants=np.array(variants), k=k.astype(np.int32), T=T.astype(np.int32), maxin=maxin.astype(np.int32))
