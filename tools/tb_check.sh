#!/bin/bash
# traceback check: DP parity tests, then three passes over the bench workload's graph
set -o pipefail
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export HIP_FORCE_DEV_KERNARG=1
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "dp" 2>&1 | grep -v amdgpu | tail -4 || exit 1
bash tools/mhc24_dpg.sh > gpurun_out/r2_dpg.log 2>&1 || { echo dpg failed; tail -5 gpurun_out/r2_dpg.log; exit 1; }
DG_DEBUG=1 timeout -k 10 100 python tools/dp_once.py /tmp/c/mhc24.dpg 1 3 2>&1 | grep -E "^pass|helpers"
