#!/bin/bash
# round-2 GPU session 1: parity suite, A/B of the row in-edge matrices on the bench graph, per-level profile
set -o pipefail
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export HIP_FORCE_DEV_KERNARG=1
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r2_pytest1.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r2_pytest1.log
tail -3 gpurun_out/r2_pytest1.log
bash tools/mhc24_dpg.sh > gpurun_out/r2_dpg.log 2>&1 || { echo dpg failed; tail gpurun_out/r2_dpg.log; exit 1; }
timeout -k 10 300 python tools/dp_perf.py --modes=fast,norowx,fast /tmp/c/mhc24.dpg > gpurun_out/r2_perf1.log 2>&1; echo "perf rc=$?"
cat gpurun_out/r2_perf1.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof1 -- python3 "$GRAFT_REPO_ROOT/tools/dp_once.py" /tmp/c/mhc24.dpg > "$GRAFT_REPO_ROOT/gpurun_out/r2_prof1.log" 2>&1; echo "rocprof rc=$?"
cd "$GRAFT_REPO_ROOT"
python3 tools/level_profile.py /tmp/c/mhc24.dpg "/tmp/prof1/**/*kernel_trace.csv" gpurun_out/r2_level_profile1.txt; echo "profile rc=$?"
head -60 gpurun_out/r2_level_profile1.txt
