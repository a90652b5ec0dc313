#!/usr/bin/env python3
"""Phase timing of the level sweep from the measurement build (make -C dipgenie_amd/csrc probe):
    python3 tools/level_probe.py graph.dpg [out.npz]
One wave per destination row stamps the 100 MHz real-time counter (10 ns ticks) at its phase boundaries; per level the
build keeps the earliest task start, the latest task end and the full timeline of the middle row's wave.
Reports, per level class: period (start to next level's start), span (first start to last end), and the sampled wave's
phases: [0] entry -> [1] row/slot records in -> [2] values in + selects done -> [3] segmented max done -> [4] stores
issued -> [5] stores complete."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("DG_LIB") is None:
    lib = os.path.join(ROOT, "bin", "libdipgenie_hip_probe.so")
    if not os.path.exists(lib):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "dipgenie_amd", "csrc"), "probe"])
    os.environ["DG_LIB"] = lib
    os.environ["DG_PROBE_OUT"] = "/tmp/dg_probe.bin"
    sys.exit(subprocess.call([sys.executable] + sys.argv))      # child process with the measurement library (no exec on a GPU box)
sys.path.insert(0, ROOT)
import numpy as np
from dipgenie_amd import capi

g = capi.DpGraphArrays.load(sys.argv[1])
ctx = capi.Context(0)
ctx.dp_load_graph(g)
for _ in range(2):
    out = ctx.dp_run()
tm = ctx.dp_timing()
L = g.n_levels
print(f"value {out.value} fwd {tm.forward_ms:.1f} ms = {1e3 * tm.forward_ms / (L - 1):.3f} us/level (probe build)")
pr = np.fromfile(os.environ["DG_PROBE_OUT"], np.uint64).reshape(L, 8)[1:].astype(np.int64)
start, end, tl = pr[:, 0], pr[:, 1], pr[:, 2:8]
ok = (tl[:, 0] > 0) & (start > 0)
period = np.diff(start, append=start[-1]) * 0.01
span = (end - start) * 0.01
ph = np.diff(tl, axis=1) * 0.01                           # 5 phases of the sampled wave, us
lead = (tl[:, 0] - start) * 0.01                          # sampled wave's entry after the level's first entry
tail = (end - tl[:, 5]) * 0.01
lo = g.level_off
nV = g.n_vertices
indeg = np.bincount(g.out_dst, minlength=nV)
lvl = np.repeat(np.arange(L), np.diff(lo))
maxin = np.zeros(L, int); np.maximum.at(maxin, lvl, indeg)
k = np.diff(lo)
def row(title, m):
    m = m & ok
    m[-1] = False
    if m.sum() == 0: return
    p = ph[m].mean(axis=0)
    print(f"  {title:28s} n={m.sum():7d} period {period[m].mean():6.3f} span {span[m].mean():6.3f} | lead {lead[m].mean():5.2f} rec {p[0]:5.2f} gather {p[1]:5.2f} "
          f"reduce {p[2]:5.2f} issue {p[3]:5.2f} drain {p[4]:5.2f} tail {tail[m].mean():5.2f}")
mx = maxin[1:]; kk = k[1:]
print("all times in us; period - span = kernel boundary (drain, flush, dispatch of the next level)")
row("all levels", np.ones(L - 1, bool))
for name, m in (("maxin <= 2", mx <= 2), ("maxin 3-8", (mx > 2) & (mx <= 8)), ("maxin 9-32", (mx > 8) & (mx <= 32)), ("maxin > 32", mx > 32)):
    row(name, m)
for name, m in (("k2 < 32", kk < 32), ("k2 32-63", (kk >= 32) & (kk < 64)), ("k2 64-127", (kk >= 64) & (kk < 128)), ("k2 128-255", (kk >= 128) & (kk < 256)), ("k2 >= 256", kk >= 256)):
    row(name, m)
if len(sys.argv) > 2:
    np.savez_compressed(sys.argv[2], start=start, end=end, tl=tl, k=k, maxin=maxin)
