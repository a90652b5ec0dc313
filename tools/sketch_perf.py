#!/usr/bin/env python3
"""Sketch timing on the bench read set (and a 30x = 7.5-fold copy of it): python tools/sketch_perf.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from dipgenie_amd import capi, synth
gfa, reads_path, info = synth.ensure_mhc24("/tmp/mhc24")
reads = [r for r in open(reads_path, "rb").read().split(b"\n")[1::2] if r]
ctx = capi.Context(0)
for mult in (1, 8):
    rs = reads * mult
    off = np.zeros(len(rs) + 1, np.int64); np.cumsum([len(r) for r in rs], out=off[1:])
    bases = b"".join(rs)
    for it in range(3):
        t0 = time.time(); h, c = ctx.sketch_reads_flat(bases, off, 31, 25); dt = time.time() - t0
        tm = ctx.sketch_timing()
    print(f"{len(rs)} reads ({off[-1]/1e6:.1f} Mbp): kernel {tm.kernel_ms:.3f} ms sort {tm.sort_ms:.3f} ms total {tm.total_ms:.3f} ms (host call {dt*1e3:.1f} ms incl. H2D) "
          f"emitted {tm.n_emitted} distinct {len(h)} -> {len(rs)/tm.total_ms/1e3:.1f} M reads/s on device", flush=True)
