#!/bin/bash
# GPU box: per-kernel durations (rocprofv3 --kernel-trace --stats) of the DP on a chr22-style panel for several option sets.
#   bash tools/sym_prof.sh <backbone_bp> "k=v,..." ["k=v,..." ...]        -> gpurun_out/sym_prof_<n>.txt
set -o pipefail
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export HIP_FORCE_DEV_KERNARG=1 DG_LIB=$PWD/bin/libdipgenie_hip_sym.so   # the measurement build (make -C dipgenie_amd/csrc sym)
BP=${1:-1000000}; shift
D=/tmp/c5ab; mkdir -p $D
[ -f $D/c5.dpg ] || { python3 tools/c5_gen.py $BP $D > gpurun_out/sym_gen.log 2>&1; bin/DipGenie -t16 -p2 -R32 -g $D/c5.gfa -r $D/c5.fa -o $D/dump.fa -D $D/c5 -X > /dev/null 2>&1; }
REPO=$(pwd); cd /tmp; export TMPDIR=/tmp
n=0
for SET in "$@"; do
  n=$((n+1)); rm -rf /tmp/sp_$n
  DG_OPTS="$SET" timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sp_$n -- python3 "$REPO/tools/dp_once.py" $D/c5.dpg 1 2 > "$REPO/gpurun_out/sym_prof_$n.log" 2>&1
  python3 - "$SET" $(find /tmp/sp_$n -name "*kernel_stats.csv" | head -1) <<'PY' | tee "$REPO/gpurun_out/sym_prof_$n.txt"
import csv, sys
print("==", sys.argv[1])
tot = 0; n = 0
for r in csv.DictReader(open(sys.argv[2])):
    if "dp_sweep" in r["Name"]:
        nm = r["Name"].split("(")[0].replace("void dgi::", "")
        print(f"{nm:46s} calls {int(r['Calls']):7d} total {float(r['TotalDurationNs']) / 1e6:8.1f} ms avg {float(r['AverageNs']) / 1e3:7.2f} us min {float(r['MinNs']) / 1e3:6.2f} max {float(r['MaxNs']) / 1e3:7.2f}")
        tot += float(r["TotalDurationNs"]); n += int(r["Calls"])
print(f"all sweep launches: {n}, {tot / 1e6:.1f} ms, {tot / n / 1e3:.2f} us each")
PY
  grep "^pass" "$REPO/gpurun_out/sym_prof_$n.log" | tail -1
done
