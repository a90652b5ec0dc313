#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export HIP_FORCE_DEV_KERNARG=1 DG_BENCH_CACHE=/tmp/dg_bench_cache
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "dp_ or cli_e2e" 2>&1 | grep -v amdgpu | tail -6
bash tools/mhc24_dpg.sh > gpurun_out/r2_dpg.log 2>&1 || { echo dpg failed; exit 1; }
timeout -k 10 300 python tools/dp_perf.py --modes=fast /tmp/c/mhc24.dpg tests/data/mhc4.dpg 2>&1 | grep -E "it=2"
