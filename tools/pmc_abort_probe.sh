#!/bin/bash
# ONE diagnostic run of the rocprofv3 --pmc abort (profiles/README.md: "dies with ~10^5 queued dispatches"): the DP of the bench workload under
# --pmc FETCH_SIZE WITHOUT the stream drains (sync_every off), with the process's memory map saved right before dg_dp_run, so that the
# addresses of the crash report (glog: PC + return addresses, no symbols) can be resolved to (library, offset, nearest exported symbol).
#   bash tools/pmc_abort_probe.sh <out_dir>            (run once; the result is profiles/r04_pmc_abort_symbolised.txt)
OUT=${1:-gpurun_out/pmc_abort}; REPO=$(pwd); mkdir -p "$OUT"
export HIP_FORCE_DEV_KERNARG=1 DG_BENCH_CACHE=${DG_BENCH_CACHE:-/tmp/dg_bench_cache}
bash tools/mhc24_dpg.sh > "$OUT/dpg.log" 2>&1 || { tail -5 "$OUT/dpg.log"; exit 1; }
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pa_FETCH
DG_DUMP_MAPS="$REPO/$OUT/maps.txt" DG_OPTS="graph_batch=0" timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pa_FETCH -- python3 "$REPO/tools/dp_once.py" /tmp/c/mhc24.dpg 1 1 > "$REPO/$OUT/run.log" 2>&1
echo "exit code $?" >> "$REPO/$OUT/run.log"
cd "$REPO"
python3 tools/pmc_abort_symbolise.py "$OUT/run.log" "$OUT/maps.txt" > "$OUT/symbolised.txt" 2>&1
cat "$OUT/symbolised.txt"
