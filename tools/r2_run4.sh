#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export HIP_FORCE_DEV_KERNARG=1 DG_BENCH_CACHE=/tmp/dg_bench_cache
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
echo "== score profile"
timeout -k 10 300 python tools/score_profile.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2_score_profile.log
timeout -k 10 300 python tools/score_profile.py 125927 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r2_score_profile.log
echo "== dp_two"
bash tools/mhc24_dpg.sh > gpurun_out/r2_dpg.log 2>&1 || { echo dpg failed; exit 1; }
timeout -k 10 300 python tools/dp_two.py /tmp/c/mhc24.dpg 3 3 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2_dp_two.log
DG_OPTS="rowx=0" timeout -k 10 300 python tools/dp_two.py /tmp/c/mhc24.dpg 3 3 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r2_dp_two.log
DG_OPTS="warm_ahead=0" timeout -k 10 300 python tools/dp_two.py /tmp/c/mhc24.dpg 3 3 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r2_dp_two.log
