#!/bin/bash
# Collects, on the MI355X box, the rocprofv3 evidence that tools/roofline_from_profiles.py turns into profiles/rNN_roofline.json:
#   bash tools/roofline_profile.sh <out_dir>          (run from the repo root; about 4 minutes)
# 1. kernel trace + stats of 3 DP passes over the bench workload's levelized graph (default launch path);
# 2. FETCH_SIZE and WRITE_SIZE of one DP pass, one --pmc pass each (no trace domains).  NOTE: these two passes run with
#    DG_SYNC_EVERY=512 -- the stream is drained every 512 level launches, without it rocprofv3 --pmc dies once ~10^5
#    dispatches are queued -- which also keeps the launches plain (no hipGraph batches): same kernels, same launch counts;
# 3. kernel trace + stats and one SQ/GRBM counter pass of the config-4 scoring pass (sketch kernels).
set -e
OUT=${1:-gpurun_out/roofline}; REPO=$(pwd); mkdir -p "$OUT"
export HIP_FORCE_DEV_KERNARG=1 DG_BENCH_CACHE=${DG_BENCH_CACHE:-/tmp/dg_bench_cache}
bash tools/mhc24_dpg.sh > "$OUT/dpg.log" 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rf_trace /tmp/rf_FETCH_SIZE /tmp/rf_WRITE_SIZE /tmp/rf_sk_trace /tmp/rf_sk_pmc
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rf_trace -- python3 "$REPO/tools/dp_once.py" /tmp/c/mhc24.dpg 1 3 > "$REPO/$OUT/trace.log" 2>&1
cp $(find /tmp/rf_trace -name "*kernel_stats.csv" | head -1) "$REPO/$OUT/dp_kernel_stats.csv"
export DG_SYNC_EVERY=512
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $C --output-format csv -d /tmp/rf_$C -- python3 "$REPO/tools/dp_once.py" /tmp/c/mhc24.dpg 1 1 > "$REPO/$OUT/pmc_$C.log" 2>&1 || echo "rocprofv3 $C failed: $?"
  python3 "$REPO/tools/pmc_sum.py" /tmp/rf_$C "$REPO/$OUT/dp_pmc_$C.csv" > /dev/null
done
unset DG_SYNC_EVERY
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rf_sk_trace -- python3 "$REPO/tools/score_profile.py" > "$REPO/$OUT/sk_trace.log" 2>&1
cp $(find /tmp/rf_sk_trace -name "*kernel_stats.csv" | head -1) "$REPO/$OUT/sketch_kernel_stats.csv"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d /tmp/rf_sk_pmc -- python3 "$REPO/tools/score_profile.py" > "$REPO/$OUT/sk_pmc.log" 2>&1 || echo "rocprofv3 sketch pmc failed: $?"
python3 "$REPO/tools/pmc_sum.py" /tmp/rf_sk_pmc "$REPO/$OUT/sketch_pmc_sq.csv" > /dev/null || true
ls -la "$REPO/$OUT"
