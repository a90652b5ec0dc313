#!/usr/bin/env python3
"""Where a scoring pass (dist_sketch.ShardedSketch.score, one rank) spends its time on the config-4 read set:
wall time per device operation (synchronised after each, so the sum over-states the pipelined pass) and the pass itself.
    python3 tools/score_profile.py [n_reads|0 = all] [generic]      (generic: the rocPRIM sort path of the spectrum)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from dipgenie_amd import capi, synth
from dipgenie_amd.dist_sketch import HipOps, ShardedSketch

cache = os.path.join(os.environ.get("DG_BENCH_CACHE", "/tmp/dg_bench_cache"), "mhc24")
gfa, _, _ = synth.ensure_mhc24(cache)
arr = np.load(synth.ensure_mhc24_reads(cache), mmap_mode="r")
n = int(sys.argv[1]) if len(sys.argv) > 1 and int(sys.argv[1]) > 0 else arr.shape[0]
arr = np.array(arr[:n]); rl = arr.shape[1]
if "pairs" in sys.argv[2:]:                                  # 300-bp reads (two tiles each): pairs of reads joined
    arr = arr[: n // 2 * 2].reshape(-1, 2 * rl); n, rl = arr.shape
dev = torch.device("cuda", 0)
ctx = capi.Context(0)
for a in sys.argv[2:]:
    if "=" in a: ctx.sketch_set_option(a.split("=")[0], int(a.split("=")[1]))      # e.g. bucket_bits=11
if "exact" in sys.argv[2:]: ctx.sketch_set_option("spectrum_mode", 2)
if "generic" in sys.argv[2:]: ctx.sketch_set_option("spectrum_mode", 1)
_, seqs, _, walks = synth.parse_gfa(gfa)
D = np.unique(np.concatenate([ctx.sketch_haplotype(b"".join(seqs[v] for v in wv), 31, 25)[0] for (_, _, wv) in walks]))
dict_t = torch.from_numpy(D.view(np.int64).copy()).to(dev)
b = torch.from_numpy(arr.reshape(-1)).to(dev); o = (torch.arange(n + 1, dtype=torch.int64) * rl).to(dev)
ops = HipOps(ctx, dev)
sk = ShardedSketch(ops, dev)
acc = {}
def wrap(name):
    f = getattr(ops, name)
    def g(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = f(*a, **k)
        torch.cuda.synchronize(); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
        return r
    setattr(ops, name, g)
for _ in range(2): sk.score(b, o, dict_t, 31, 25)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): sk.score(b, o, dict_t, 31, 25)
torch.cuda.synchronize(); plain = (time.perf_counter() - t0) / 5
for m in ("sketch_reads", "count_dictionary", "rank_dictionary", "histogram"): wrap(m)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): sk.score(b, o, dict_t, 31, 25)
torch.cuda.synchronize(); tot = (time.perf_counter() - t0) / 5
tm = ctx.sketch_timing()
print(f"{n} reads: pass {1e3 * plain:.2f} ms ({n / plain / 1e6:.1f} M reads/s); with per-op synchronisation {1e3 * tot:.2f} ms")
for k, v in acc.items(): print(f"  {k:18s} {1e3 * v / 5:7.2f} ms")
print(f"  (sketch_reads device events: tile kernel + compaction {tm.kernel_ms:.2f} ms, sort + reduce {tm.sort_ms:.2f} ms)")
print(f"  spectrum path {ctx.sketch_stat('spectrum_path')} buckets {ctx.sketch_stat('buckets')} overflow {ctx.sketch_stat('overflow_buckets')}")
print(f"  rest (torch glue)  {1e3 * (tot - sum(acc.values()) / 5):7.2f} ms")
