#!/bin/bash
# round-2 GPU session 1: parity suite, A/B of the row in-edge matrices on the bench graph, per-level profile
set -o pipefail
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export HIP_FORCE_DEV_KERNARG=1
echo skip pytest

bash tools/mhc24_dpg.sh > gpurun_out/r2_dpg.log 2>&1 || { echo dpg failed; tail gpurun_out/r2_dpg.log; exit 1; }
echo skip perf

cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof1 -- python3 "$GRAFT_REPO_ROOT/tools/dp_once.py" /tmp/c/mhc24.dpg > "$GRAFT_REPO_ROOT/gpurun_out/r2_prof1.log" 2>&1; echo "rocprof rc=$?"
cd "$GRAFT_REPO_ROOT"
python3 tools/level_profile.py /tmp/c/mhc24.dpg "/tmp/prof1/**/*kernel_trace.csv" gpurun_out/r2_level_profile1.txt gpurun_out/r2_levels1.npz; echo "profile rc=$?"
head -12 gpurun_out/r2_level_profile1.txt
timeout -k 10 200 python3 tools/level_probe.py /tmp/c/mhc24.dpg gpurun_out/r2_probe1.npz > gpurun_out/r2_probe1.log 2>&1; echo "probe rc=$?"; cat gpurun_out/r2_probe1.log
