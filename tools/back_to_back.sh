#!/bin/bash
# Does the HBM of a CLI run that has just exited delay the next run?  (GPU box)  bash tools/back_to_back.sh
REPO=$(pwd); D=/tmp/dg_bench_cache/mhc24
python3 - <<PY
import sys; sys.path.insert(0, "$REPO")
from dipgenie_amd import synth
synth.ensure_mhc24("$D")
PY
run() {  # label, env...
  local label=$1; shift
  S=$(date +%s.%N)
  env DG_DEBUG=1 "$@" "$REPO/bin/DipGenie" -t 16 -p2 -R18 -g $D/mhc24.gfa -r $D/mhc24_4x.fa -o /tmp/x.fa > /tmp/x.out 2> /tmp/x.err
  E=$(date +%s.%N)
  echo "== $label: wall $(python3 -c "print(round($E - $S, 3))") s; $(grep -h 'dp_level_loop\|plan lattice\|uploads\|dg_destroy\|destroy:' /tmp/x.err | tr -s ' ' | tr '\n' ';')"
}
run first; run second_right_after; sleep 5
run clean DG_CLEAN_EXIT=1 DG_CLEAN_EXIT_FAST=1; run after_clean; sleep 5
run first; sleep 1; run after_1s; sleep 2; run after_2s
