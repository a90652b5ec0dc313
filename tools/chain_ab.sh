#!/bin/bash
# GPU box: DP parity tests, then the chained-dispatch A/B on the bench workload's graph.  bash tools/chain_ab.sh ["k=v,..." ...]
set -o pipefail
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export HIP_FORCE_DEV_KERNARG=1 DG_LIB=$PWD/bin/libdipgenie_hip_chain.so   # the measurement build (make -C dipgenie_amd/csrc chain)
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "dp_" --timeout 60 > gpurun_out/chain_tests.log 2>&1; tail -3 gpurun_out/chain_tests.log
grep -q passed gpurun_out/chain_tests.log && ! grep -q failed gpurun_out/chain_tests.log || { grep -B30 Error gpurun_out/chain_tests.log | tail -60; exit 1; }
bash tools/mhc24_dpg.sh > gpurun_out/dpg.log 2>&1 || { echo dpg failed; tail -5 gpurun_out/dpg.log; exit 1; }
DG_DEBUG=1 timeout -k 10 280 python tools/chain_ab.py /tmp/c/mhc24.dpg "$@" 2>&1 | grep -v "^\[dipgenie_hip\] load\|lattice\|helpers\|prefetcher covered\|host issued"
