#!/usr/bin/env python3
"""RCCL rehearsal on ONE GPU: torch.distributed with the "nccl" backend (= RCCL) in a world of one rank, the sharded scoring
class forced through its collective path (async all-reduce, all-gather of the send counts, all-to-all of the runs, fused
all-reduce) on the config-4 read set, checked against the collective-free pass.  Two ranks on one device are refused by RCCL
("Duplicate GPU detected"), so this is as close as a one-GPU box gets to the N > 1 bench: every RCCL entry point the class
uses is called with the dtypes, devices and streams of the product path.
usage: python3 tools/rccl_rehearsal.py [n_reads]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("DG_BENCH_CACHE", "/tmp/dg_bench_cache")
import bench
from dipgenie_amd import capi, synth
from dipgenie_amd.dist_sketch import HipOps, ShardedSketch

torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
device = "cuda:0"
K, W = 31, 25
cache = os.environ.get("DG_BENCH_CACHE", "/tmp/dg_bench_cache")
gfa, _reads, _info = synth.ensure_mhc24(os.path.join(cache, "mhc24"))
ctx = capi.Context(0)
_, seqs, _, walks = synth.parse_gfa(gfa)
D = np.unique(np.concatenate([ctx.sketch_haplotype(b"".join(seqs[v] for v in wv), K, W)[0] for (_, _, wv) in walks]))
dict_t = torch.from_numpy(D.view(np.int64).copy()).to(device)
(b4, o4), resident, n4, rl4 = bench.load_config4_shard(cache, device, 1, 0)
if len(sys.argv) > 1:
    b4, o4 = resident(0, int(sys.argv[1]))
ops = HipOps(ctx, device)
plain = ShardedSketch(ops, device).score(b4, o4, dict_t, K, W)
forced = ShardedSketch(ops, device, force_exchange=True)
out = forced.score(b4, o4, dict_t, K, W)
assert bench.same_score(plain, out) and torch.equal(plain.range_hash, out.range_hash) and torch.equal(plain.range_count, out.range_count), "forced-exchange scoring differs"
gh, gc = forced.gather_spectrum(out)
assert torch.equal(gh, plain.range_hash) and torch.equal(gc, plain.range_count)
for _ in range(6):                                              # (the first passes of the collective path still pay lazy set-up: 3.5, 2.6, 2.4 ms ...)
    out = forced.score(b4, o4, dict_t, K, W)
N_PASS = 20
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(N_PASS):
    out = forced.score(b4, o4, dict_t, K, W)
torch.cuda.synchronize(); t1 = time.perf_counter()
for _ in range(N_PASS):
    plain = ShardedSketch(ops, device).score(b4, o4, dict_t, K, W)
torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"RCCL world-1 rehearsal OK: {o4.numel() - 1} reads, {out.n_distinct} distinct hashes; collective path {1e3 * (t1 - t0) / N_PASS:.2f} ms per pass, "
      f"collective-free {1e3 * (t2 - t1) / N_PASS:.2f} ms (the difference is the fixed cost of 3 collectives per pass: hit-vector all-reduce, fixed-size all-to-all of [world, 1 + cap, 2] blocks (cap = {forced.cap}), fused all-reduce; "
      f"no size exchange and no host read in dist_sketch.py after the calibrating first pass; overflow flag {out.exchange_overflow})")
forced.laps = {}
for _ in range(5):
    forced.score(b4, o4, dict_t, K, W)
print("stages of the collective path (ms per pass, each followed by a device synchronisation):")
for k_, v in forced.laps.items():
    print(f"  {k_:34s} {1e3 * v / 5:.3f}")
dist.destroy_process_group()
