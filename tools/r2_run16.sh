#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export HIP_FORCE_DEV_KERNARG=1
bash tools/mhc24_dpg.sh > gpurun_out/r2_dpg.log 2>&1 || { echo dpg failed; exit 1; }
for rep in 1 2; do
for v in "" bin/libdipgenie_hip_valnt.so; do
echo "-- lib=${v:-default}"; DG_LIB=${v:+$PWD/$v} timeout -k 10 100 python tools/dp_once.py /tmp/c/mhc24.dpg 1 3 2>&1 | grep -E "^pass"
done; done
