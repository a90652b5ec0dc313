#!/bin/bash
# Where does the time between the CLI's last line and the parent's wait() go?  (GPU box)  bash tools/exit_cost.sh
REPO=$(pwd); D=/tmp/dg_bench_cache/mhc24
python3 - <<PY
import sys; sys.path.insert(0, "$REPO")
from dipgenie_amd import synth
synth.ensure_mhc24("$D")
PY
run() {  # label, env...
  local label=$1; shift
  sleep 4
  S=$(date +%s.%N)
  env DG_DEBUG=1 "$@" "$REPO/bin/DipGenie" -t 16 -p2 -R18 -g $D/mhc24.gfa -r $D/mhc24_4x.fa -o /tmp/x.fa > /tmp/x.out 2> /tmp/x.err
  E=$(date +%s.%N)
  echo "== $label: wall $(python3 -c "print(round($E - $S, 3))") s; $(grep -h 'Real time\|leaving\|dg_destroy\|destroy:' /tmp/x.err | tr '\n' ';')"
}
run default
run default
run clean_exit DG_CLEAN_EXIT=1
run device_freed_host_left DG_CLEAN_EXIT=1 DG_CLEAN_EXIT_FAST=1
run default
