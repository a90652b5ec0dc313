#!/bin/bash
# rocprofv3 evidence for BASELINE configs[4] (chr22-style panel, 100 walks, R = 32) at the 5 Mbp tier, on the MI355X box:
#   bash tools/c5_profile.sh <out_dir> [backbone_bp]          (run from the repo root; about 6 minutes at 5 Mbp)
# 1. the drop-in CLI alone (DG_DEBUG stage lines, forward / traceback split, segments);
# 2. rocprofv3 --kernel-trace --stats of the same command (the program itself after --);
# 3. FETCH_SIZE, WRITE_SIZE, then SQ counters, one --pmc pass each (no trace domains), DG_DP_OPTIONS=sync_every=512
#    (plain launches, stream drained every 512 levels: rocprofv3 --pmc dies with ~10^5 queued dispatches);
# 4. level geometry of the panel's levelized graph.
set -e -x
OUT=${1:-gpurun_out/c5prof}; BP=${2:-5000000}; REPO=$(pwd); mkdir -p "$OUT"
D=/tmp/c5p; mkdir -p $D
python3 tools/c5_gen.py $BP $D > "$OUT/gen.log" 2>&1
CMD="$REPO/bin/DipGenie -t16 -p2 -R32 -g $D/c5.gfa -r $D/c5.fa"
export HIP_FORCE_DEV_KERNARG=1
DG_DEBUG=1 $CMD -o $D/plain.fa -J "$REPO/$OUT/plain.json" > "$OUT/plain.out" 2> "$OUT/plain.err"
grep -E "stage\]|lattice|dg::dp|Real time" "$OUT/plain.err" > "$OUT/plain_stages.txt" || true
cd /tmp && export TMPDIR=/tmp
export DG_CLEAN_EXIT=1        # the CLI normally leaves through _exit(): rocprofv3 writes its files from an exit handler
rm -rf /tmp/c5_trace /tmp/c5_FETCH_SIZE /tmp/c5_WRITE_SIZE /tmp/c5_SQ
sleep 5
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c5_trace -- $CMD -o $D/trace.fa -J "$REPO/$OUT/trace.json" > "$REPO/$OUT/trace.log" 2>&1
cp $(find /tmp/c5_trace -name "*kernel_stats.csv" | head -1) "$REPO/$OUT/c5_kernel_stats.csv"
export DG_DP_OPTIONS=sync_every=512
for C in FETCH_SIZE WRITE_SIZE; do
  sleep 5
  timeout -k 10 500 rocprofv3 --pmc $C --output-format csv -d /tmp/c5_$C -- $CMD -o $D/pmc.fa -J "$REPO/$OUT/pmc_$C.json" > "$REPO/$OUT/pmc_$C.log" 2>&1 || echo "rocprofv3 $C failed: $?"
  python3 "$REPO/tools/pmc_sum.py" /tmp/c5_$C "$REPO/$OUT/c5_pmc_$C.csv" > /dev/null || true
  rm -rf /tmp/c5_$C
done
sleep 5
timeout -k 10 500 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d /tmp/c5_SQ -- $CMD -o $D/pmc.fa > "$REPO/$OUT/pmc_SQ.log" 2>&1 || echo "rocprofv3 SQ failed: $?"
python3 "$REPO/tools/pmc_sum.py" /tmp/c5_SQ "$REPO/$OUT/c5_pmc_SQ.csv" > /dev/null || true
unset DG_DP_OPTIONS
cmp $D/plain.fa $D/trace.fa && echo "FASTA identical under the tracer" > "$REPO/$OUT/identical.txt"
cd "$REPO"
if true; then
  $CMD -o $D/dump.fa -D $D/c5 -X > /dev/null 2>&1 || true
  python3 tools/level_geometry.py $D/c5.dpg > "$OUT/geometry.txt" 2>&1 || true
fi
ls -la "$OUT"
