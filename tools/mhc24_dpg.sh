#!/bin/bash
# Builds the bench workload's levelized graph on the GPU box: bash tools/mhc24_dpg.sh  ->  /tmp/c/mhc24.dpg
set -e
REPO=$(pwd); mkdir -p /tmp/c
python3 - <<PY
import sys; sys.path.insert(0, "$REPO")
from dipgenie_amd import synth
print(synth.ensure_mhc24("/tmp/mhc24")[:2])
PY
[ -f /tmp/c/mhc24.dpg ] || "$REPO/bin/DipGenie" -t16 -p2 -R18 -g /tmp/mhc24/mhc24.gfa -r /tmp/mhc24/mhc24_4x.fa -o /tmp/c/o.fa -D /tmp/c/mhc24 > /tmp/c/cli.log 2>&1 || { tail -20 /tmp/c/cli.log; exit 1; }
ls -la /tmp/c/mhc24.dpg
