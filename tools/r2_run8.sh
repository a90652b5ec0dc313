#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export HIP_FORCE_DEV_KERNARG=1 DG_BENCH_CACHE=/tmp/dg_bench_cache
echo "== bench N=1"
timeout -k 10 900 python bench.py > gpurun_out/r2_bench2.json 2> gpurun_out/r2_bench2.err; echo "bench rc=$?"; tail -2 gpurun_out/r2_bench2.err
python - <<'PY'
import json
l=json.loads([x for x in open("gpurun_out/r2_bench2.json") if x.startswith("{")][-1])
print({k:l[k] for k in ("value","ms_per_step","end_to_end_s")}, l["dp_ms"], l["roofline"]["frac"], l["sketch_config4"]["reads_per_s"], l["concurrent_instances"]["value"], l["cpu_baseline"]["value"])
PY
echo "== roofline profile"
bash tools/roofline_profile.sh gpurun_out/r2_roofline > gpurun_out/r2_roofline.log 2>&1; echo "rc=$?"; tail -15 gpurun_out/r2_roofline.log
python3 tools/roofline_from_profiles.py gpurun_out/r2_bench2.json gpurun_out/r2_roofline gpurun_out/r2_roofline.json
echo "== 2 ranks nccl on one device (expected to be refused by RCCL)"
DG_BENCH_DEVICE=0 timeout -k 10 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 2 --steps 3 --warmup 1 > gpurun_out/r2_bench_2rank_nccl.json 2> gpurun_out/r2_bench_2rank_nccl.err; echo "rc=$?"; grep -v amdgpu.ids gpurun_out/r2_bench_2rank_nccl.err | tail -6; cut -c1-400 gpurun_out/r2_bench_2rank_nccl.json
