#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export HIP_FORCE_DEV_KERNARG=1 DG_BENCH_CACHE=/tmp/dg_bench_cache
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config4.py -m gpu -x -q -k "sketch or hash or config4 or anchor" 2>&1 | tail -5
timeout -k 10 600 python tools/sketch_stress.py 1200 2>&1 | tail -3
timeout -k 10 300 python tools/score_profile.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2_score_profile2.log
