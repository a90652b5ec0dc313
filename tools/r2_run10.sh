#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export HIP_FORCE_DEV_KERNARG=1
bash tools/mhc24_dpg.sh > gpurun_out/r2_dpg.log 2>&1 || { echo dpg failed; exit 1; }
for rep in 1 2; do
for L in "" bin/libdipgenie_hip_prev.so; do
  echo "== DG_LIB=$L"
  DG_LIB=$L timeout -k 10 300 python tools/dp_perf.py --modes=fast /tmp/c/mhc24.dpg tests/data/mhc4.dpg 2>&1 | grep -E "it=[12]|==" 
done; done
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "dp_" 2>&1 | tail -3
