#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export HIP_FORCE_DEV_KERNARG=1
bash tools/mhc24_dpg.sh > gpurun_out/r2_dpg.log 2>&1 || { echo dpg failed; tail gpurun_out/r2_dpg.log; exit 1; }
timeout -k 10 500 python tools/dp_opt_grid.py "val_wt=0,coop_first=0,fit_wg=0;val_wt=0,coop_first=1,fit_wg=0;val_wt=0,coop_first=0,fit_wg=1;val_wt=1,coop_first=0,fit_wg=0;val_wt=1,coop_first=1,fit_wg=1;val_wt=0,coop_first=1,fit_wg=1;val_wt=0,coop_first=0,fit_wg=0" /tmp/c/mhc24.dpg tests/data/mhc4.dpg 2>&1 | tee gpurun_out/r2_grid2.log
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "dp_" 2>&1 | tail -3
