#!/usr/bin/env python3
"""Randomised GPU-vs-oracle parity sweep of the sketch kernels: python tools/sketch_stress.py [n_cases] [seed0]
Random (k, w) in 1..64 x 1..64, alphabets with N / lower case / IUPAC, low-complexity repeats, ragged read lengths."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_py as orc
from dipgenie_amd import capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 4242
ctx = capi.Context(0)
rng = np.random.default_rng(seed0)
alphas = [b"ACGT", b"ACGT", b"ACGTN", b"acgtACGT", b"ACGTRYKMSWn", b"AC", b"A", b"ACGTACGTACGTN"]
def rnd(m, a): return bytes(rng.choice(np.frombuffer(a, np.uint8), m).tobytes()) if m else b""
bad = 0
paths, hosted = [0, 0, 0], 0
for t in range(n):
    k = int(rng.integers(1, 65)); w = int(rng.integers(1, 65))
    a = alphas[int(rng.integers(0, len(alphas)))]
    reads = [rnd(int(rng.integers(0, 400)), a) for _ in range(int(rng.integers(0, 300)))]
    if rng.random() < 0.3: reads += [rnd(int(rng.integers(1, 30)), b"ACGT") * int(rng.integers(1, 40)) for _ in range(5)]   # tandem repeats
    if rng.random() < 0.3 and reads: reads += [reads[0], reads[-1]]                                                           # duplicates
    # the read spectrum's routes and bucket geometry at random (round 3): buckets filled by the tile kernel / exact placement, tiny
    # strides and residual lists (host-finished buckets, the give-up route)
    opts = {"spectrum_mode": int(rng.choice([0, 0, 0, 2])), "bucket_bits": int(rng.choice([0, 0, 1, 3, 6])), "bucket_stride": int(rng.choice([0, 0, 0, 64, 1024])),
            "residual_cap": int(rng.choice([0, 0, 0, 4, -1])), "host_buckets": int(rng.choice([0, 0, 3])), "spill_cap": int(rng.choice([0, 0, 0, 50, -1]))}
    for key, v in opts.items(): ctx.sketch_set_option(key, v)
    hg, cg = ctx.sketch_reads(reads, k, w)
    paths[ctx.sketch_stat("spectrum_path")] += 1
    hosted += ctx.sketch_stat("overflow_buckets") > 0
    ho, co = orc.sketch_reads(reads, k, w)
    ok = np.array_equal(hg, ho) and np.array_equal(cg, co)
    hap = rnd(int(rng.integers(0, 60000)), a) + rnd(int(rng.integers(0, 200)), b"ACGTN") + rnd(int(rng.integers(0, 20000)), b"ACGT")
    hg2, pg2 = ctx.sketch_haplotype(hap, k, w)
    ho2, po2 = orc.minimizers(hap, k, w)
    ok = ok and np.array_equal(hg2, ho2) and np.array_equal(pg2, po2)
    if t % 100 == 99: print(f"... {t + 1} cases, {bad} mismatches so far", flush=True)
    if not ok:
        bad += 1
        print("MISMATCH", t, k, w, a, len(reads), len(hap), opts, flush=True)
print(f"{n} cases, {bad} mismatches; spectrum routes: {paths[0]} buckets filled by the tile kernel, {paths[1]} exact placement, {paths[2]} generic; {hosted} with host-finished buckets", flush=True)
sys.exit(1 if bad else 0)
