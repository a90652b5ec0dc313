#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export HIP_FORCE_DEV_KERNARG=1 DG_BENCH_CACHE=/tmp/dg_bench_cache
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "launch_profile or alternative" 2>&1 | grep -v amdgpu | tail -4 || exit 1
timeout -k 10 900 python bench.py > gpurun_out/r2_bench_final.json 2> gpurun_out/r2_bench_final.err; echo "bench rc=$?"
for rep in 1 2; do
DG_DEBUG=1 ./bin/DipGenie -t16 -p2 -R18 -g /tmp/dg_bench_cache/mhc24/mhc24.gfa -r /tmp/dg_bench_cache/mhc24/mhc24_4x.fa -o /tmp/o.fa 2>&1 | grep -E "dg::stage|dipgenie_hip\]|\[dg\]" | tail -40
echo ----
done
