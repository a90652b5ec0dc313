#!/bin/bash
# GPU box: A/B without the test suite, then a kernel trace of one chained pass.  bash tools/chain_prof.sh ["k=v,..." ...]
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export HIP_FORCE_DEV_KERNARG=1 DG_LIB=$PWD/bin/libdipgenie_hip_chain.so   # the measurement build (make -C dipgenie_amd/csrc chain)
bash tools/mhc24_dpg.sh > gpurun_out/dpg.log 2>&1 || { echo dpg failed; tail -5 gpurun_out/dpg.log; exit 1; }
timeout -k 10 400 python tools/chain_ab.py /tmp/c/mhc24.dpg "$@" 2>&1 | grep -v "amdgpu.ids"
