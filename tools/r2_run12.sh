#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export HIP_FORCE_DEV_KERNARG=1
bash tools/mhc24_dpg.sh > gpurun_out/r2_dpg.log 2>&1 || { echo dpg failed; exit 1; }
timeout -k 10 600 python tools/dp_opt_grid.py "graph_batch=0;graph_batch=1000;graph_batch=4000;graph_batch=0;graph_batch=250" /tmp/c/mhc24.dpg 2>&1 | grep -v amdgpu | tee gpurun_out/r2_grid12.log
DG_DEBUG=1 timeout -k 10 100 python tools/dp_once.py /tmp/c/mhc24.dpg 1 2 2>&1 | grep -E "host issued|value"
