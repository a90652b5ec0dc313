#!/bin/bash
# HBM traffic counters of the bench workload's DP pass (run on the MI355X box from the repo root):
#   bash tools/pmc_mhc24.sh <out_dir> [iteration-range]
# Builds the synthetic MHC-24 .dpg with the drop-in CLI, then one rocprofv3 --pmc pass per counter
# (no trace domains), counters limited to a dispatch-iteration range of every kernel because the
# profiler crashes when it has to keep counters for all 140 k dispatches.
set -e
OUT=${1:-gpurun_out/pmc24}; RANGE=${2:-[1-3000]}
REPO=$(pwd); mkdir -p "$OUT" /tmp/c
python3 - <<PY
import sys; sys.path.insert(0, "$REPO")
from dipgenie_amd import synth
print(synth.ensure_mhc24("/tmp/mhc24")[:2])
PY
[ -f /tmp/c/mhc24.dpg ] || "$REPO/bin/DipGenie" -t16 -p2 -R18 -g /tmp/mhc24/mhc24.gfa -r /tmp/mhc24/mhc24_4x.fa -o /tmp/c/o.fa -D /tmp/c/mhc24 > /tmp/c/cli.log 2>&1
ls -la /tmp/c/mhc24.dpg
cd /tmp && export TMPDIR=/tmp DG_SYNC_EVERY=${DG_SYNC_EVERY:-512}
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C
  timeout -k 10 400 rocprofv3 --pmc $C --kernel-iteration-range "$RANGE" --output-format csv -d /tmp/pmc_$C -- python3 "$REPO/tools/dp_once.py" /tmp/c/mhc24.dpg 1 1 > "$REPO/$OUT/pmc24_$C.log" 2>&1 || echo "rocprofv3 $C failed: $?"
  python3 "$REPO/tools/pmc_sum.py" /tmp/pmc_$C "$REPO/$OUT/pmc24_$C.csv" > /dev/null || true
  tail -3 "$REPO/$OUT/pmc24_$C.log"; cat "$REPO/$OUT/pmc24_$C.csv" || true
done
