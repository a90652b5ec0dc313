#!/bin/bash
# One-GPU box: BASELINE config 4 as ONE job at world 2 and 3 -- both ranks on device 0, gloo in place of RCCL (RCCL refuses two ranks
# on one device) -- the haplotype-sharded sketches, the hash-range exchange and the injected spectrum with the real HIP operations.
# The FASTA must be the reference's (md5 cd13930a... for the bench panel with its 4x reads).   bash tools/run_sharded_world2.sh
python3 - <<PY
import sys; sys.path.insert(0, ".")
from dipgenie_amd import synth
print(synth.ensure_mhc24("/tmp/dg_bench_cache/mhc24")[:2])
PY
D=/tmp/dg_bench_cache/mhc24
for W in 2 3; do
  S=$(date +%s.%N)
  timeout -k 10 300 python3 -m dipgenie_amd.run_sharded --gpus $W --backend gloo --device 0 -q -g $D/mhc24.gfa -r $D/mhc24_4x.fa -o /tmp/sh_$W.fa -J /tmp/sh_$W.json > /tmp/sh_$W.log 2>&1
  rc=$?; E=$(date +%s.%N)
  echo "world $W: rc=$rc wall $(python3 -c "print(round($E - $S, 2))") s  md5 $(md5sum < /tmp/sh_$W.fa | cut -c1-8)"
  tail -2 /tmp/sh_$W.log | cut -c1-200
done
