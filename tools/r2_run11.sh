#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export HIP_FORCE_DEV_KERNARG=1
bash tools/mhc24_dpg.sh > gpurun_out/r2_dpg.log 2>&1 || { echo dpg failed; exit 1; }
timeout -k 10 200 python3 tools/level_probe.py /tmp/c/mhc24.dpg gpurun_out/r2_probe2.npz 2>&1 | grep -v amdgpu | tee gpurun_out/r2_probe2.log
