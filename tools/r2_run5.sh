#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export HIP_FORCE_DEV_KERNARG=1 DG_BENCH_CACHE=/tmp/dg_bench_cache
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "anchor" 2>&1 | tail -5
echo "== mhc24 e2e dev vs host anchors"
python tools/run_mhc24.py 2 16 2>&1 | grep -v amdgpu.ids | grep -E "anchors|index|wall|md5|Real" | tail -40
DG_HOST_ANCHORS=1 python tools/run_mhc24.py 1 16 2>&1 | grep -v amdgpu.ids | grep -E "anchors|index|total|wall" | tail -12
