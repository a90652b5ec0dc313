#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export HIP_FORCE_DEV_KERNARG=1
bash tools/mhc24_dpg.sh > gpurun_out/r2_dpg.log 2>&1 || { echo dpg failed; exit 1; }
for rep in 1 2 3; do
echo "-- plain"; DG_DEBUG=1 timeout -k 10 100 python tools/dp_once.py /tmp/c/mhc24.dpg 1 3 2>&1 | grep -E "^pass|host issued"
echo "-- graph 1000"; DG_GRAPH_BATCH=1000 DG_DEBUG=1 timeout -k 10 100 python tools/dp_once.py /tmp/c/mhc24.dpg 1 3 2>&1 | grep -E "^pass|host issued"
done
