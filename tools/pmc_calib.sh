#!/bin/bash
# bash tools/pmc_calib.sh <out_dir>   (on the MI355X box, repo root)
OUT=${1:-gpurun_out/calib}; REPO=$(pwd); mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/cal_$C
  timeout -k 10 120 rocprofv3 --pmc $C --output-format csv -d /tmp/cal_$C -- "$REPO/bin/pmc_calib" > "$REPO/$OUT/calib_$C.log" 2>&1 || echo "failed $?"
  python3 "$REPO/tools/pmc_sum.py" /tmp/cal_$C "$REPO/$OUT/calib_$C.csv" > /dev/null
  cat "$REPO/$OUT/calib_$C.csv"
done
