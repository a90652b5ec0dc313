#!/bin/bash
# End-to-end timeline of the drop-in CLI on the bench workload (GPU box): bash tools/e2e_timeline.sh [reps] [threads]
# Prints, per run: wall seconds as a parent process sees them, the CLI's own "Real time", every DG_DEBUG lap.
REPS=${1:-3}; THR=${2:-16}
REPO=$(pwd)
python3 - <<PY
import sys; sys.path.insert(0, "$REPO")
from dipgenie_amd import synth
print(synth.ensure_mhc24("/tmp/dg_bench_cache/mhc24")[:2])
PY
D=/tmp/dg_bench_cache/mhc24
for i in $(seq 1 $REPS); do
  S=$(date +%s.%N)
  DG_DEBUG=1 "$REPO/bin/DipGenie" -t $THR -p2 -R18 -g $D/mhc24.gfa -r $D/mhc24_4x.fa -o /tmp/e2e_$i.fa -J /tmp/e2e_$i.json > /tmp/e2e_$i.out 2> /tmp/e2e_$i.err
  E=$(date +%s.%N)
  echo "== run $i: rc=$? wall $(python3 -c "print(round($E - $S, 3))") s  md5 $(md5sum < /tmp/e2e_$i.fa | cut -c1-8)"
  grep -v '^[A-Za-z0-9_.#]* : ' /tmp/e2e_$i.err | grep 'dg::\|dipgenie_hip\]\|Real time'
done
