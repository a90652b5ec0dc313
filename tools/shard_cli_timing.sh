#!/bin/bash
# GPU box: wall of the one-process sharded CLI (bin/DipGenie --gpus 1, RCCL / host-staged) beside the plain CLI on the 30x read set, three runs each, 6 s apart
export DG_BENCH_CACHE=${DG_BENCH_CACHE:-/tmp/dg_bench_cache} HIP_FORCE_DEV_KERNARG=1
python3 - <<PY
import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from dipgenie_amd import synth
c = os.path.join(os.environ["DG_BENCH_CACHE"], "mhc24")
gfa, _, _ = synth.ensure_mhc24(c)
arr = np.load(synth.ensure_mhc24_reads(c), mmap_mode="r"); n, rl = arr.shape
b = np.empty((n, 3 + rl + 1), np.uint8); b[:, :3] = np.frombuffer(b">r\n", np.uint8); b[:, 3:3 + rl] = arr; b[:, -1] = 10
open("/tmp/reads30.fa", "wb").write(b.tobytes()); print(gfa)
PY
G=$DG_BENCH_CACHE/mhc24/mhc24.gfa
for MODE in "" "--gpus 1 --shard-transport rccl" "--gpus 2 --shard-transport host"; do
  for i in 1 2 3; do
    sleep 6
    T0=$(date +%s%N)
    DG_DEBUG=1 bin/DipGenie -t16 -p2 -R18 $MODE -g $G -r /tmp/reads30.fa -o /tmp/o.fa > /tmp/o.out 2> /tmp/o.err; RC=$?
    T1=$(date +%s%N)
    echo "wall $(( (T1 - T0) / 1000000 )) ms rc=$RC [$MODE]"; grep -E "shard" /tmp/o.err
  done
done
