#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export HIP_FORCE_DEV_KERNARG=1 DG_BENCH_CACHE=/tmp/dg_bench_cache
timeout -k 10 1100 python -m pytest tests -m gpu -x -q 2>&1 | grep -v amdgpu | tail -4 || exit 1
timeout -k 10 500 python tools/dp_stress.py 200 881000 2>&1 | tail -1
timeout -k 10 900 python bench.py > gpurun_out/r2_bench_final.json 2> gpurun_out/r2_bench_final.err; echo "bench rc=$?"
python - <<'PY'
import json
l=json.loads([x for x in open("gpurun_out/r2_bench_final.json") if x.startswith("{")][-1])
print({k:l[k] for k in ("value","ms_per_step","end_to_end_s")}, l["dp_ms"], l["roofline"]["frac"], l["sketch_config4"]["reads_per_s"], l["concurrent_instances"]["value"])
print(l["roofline"].get("traffic_note"))
PY
bash tools/roofline_profile.sh gpurun_out/r2_roofline 2>&1 | tail -3
