#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export HIP_FORCE_DEV_KERNARG=1 DG_BENCH_CACHE=/tmp/dg_bench_cache
timeout -k 10 600 python -m pytest tests/test_gpu_config4.py -m gpu -x -q 2>&1 | tail -15
echo "== bench N=1"
timeout -k 10 900 python bench.py > gpurun_out/r2_bench1.json 2> gpurun_out/r2_bench1.err; echo "bench rc=$?"; tail -3 gpurun_out/r2_bench1.err; cat gpurun_out/r2_bench1.json
echo "== bench 2 ranks, gloo rehearsal on one GPU"
DG_BENCH_BACKEND=gloo DG_BENCH_DEVICE=0 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 1 > gpurun_out/r2_bench_2rank_gloo.json 2> gpurun_out/r2_bench_2rank_gloo.err; echo "rc=$?"; tail -5 gpurun_out/r2_bench_2rank_gloo.err; cat gpurun_out/r2_bench_2rank_gloo.json
