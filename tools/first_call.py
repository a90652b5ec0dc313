#!/usr/bin/env python3
"""Cost of the first call into each part of the library in a fresh process (code-object load, first allocations)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
t0 = time.perf_counter()
from dipgenie_amd import capi
t1 = time.perf_counter()
ctx = capi.Context(0)
t2 = time.perf_counter()
print(f"import capi {1e3*(t1-t0):.1f} ms, dg_create {1e3*(t2-t1):.1f} ms")
seq = b"ACGTTGCATGCATTGACCATGACGTTGCATGCAAGTCCATGACGATGACTAGCATGCATGCATTTAGCGAC" * 4
for i in range(3):
    t = time.perf_counter(); ctx.sketch_haplotype(seq, 31, 25); print(f"sketch_haplotype #{i}: {1e3*(time.perf_counter()-t):.2f} ms")
for i in range(3):
    t = time.perf_counter(); ctx.sketch_reads([seq[:150], seq[50:200]], 31, 25); print(f"sketch_reads #{i}: {1e3*(time.perf_counter()-t):.2f} ms")
g = capi.DpGraphArrays.load(os.path.join(ROOT, "tests", "golden", "toy2_R2.dpg"))
for i in range(3):
    t = time.perf_counter(); ctx.dp_solve(g); print(f"dp_solve(toy) #{i}: {1e3*(time.perf_counter()-t):.2f} ms")
