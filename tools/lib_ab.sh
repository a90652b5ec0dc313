#!/bin/bash
# GPU box: the same three DP passes over the bench workload's graph with several builds of the library (bin/lib_*.so) and the tree's own.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export HIP_FORCE_DEV_KERNARG=1
bash tools/mhc24_dpg.sh > gpurun_out/dpg.log 2>&1 || { echo dpg failed; exit 1; }
for lib in "$@" dipgenie_amd/csrc/libdipgenie_hip.so; do
  for rep in 1 2; do
    echo "== $lib"; DG_LIB=$PWD/$lib timeout -k 10 100 python tools/dp_once.py /tmp/c/mhc24.dpg 1 4 2>&1 | grep "^pass" | tail -2
  done
done
