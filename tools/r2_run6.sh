#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export HIP_FORCE_DEV_KERNARG=1 DG_BENCH_CACHE=/tmp/dg_bench_cache
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "haploid" 2>&1 | tail -5
echo "== mhc4 -p1 dev vs host"
for e in "" "DG_HOST_HAPLOID=1"; do
  env $e DG_DEBUG=1 bin/DipGenie -t16 -p1 -g tests/data/MHC_4.gfa.gz -r tests/data/CHM13_reads.fq.gz -o /tmp/p1.fa 2>&1 | grep -E "haploid|Real time|Recombination" ; md5sum /tmp/p1.fa
done
