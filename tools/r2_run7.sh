#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export HIP_FORCE_DEV_KERNARG=1
bash tools/mhc24_dpg.sh > gpurun_out/r2_dpg.log 2>&1 || { echo dpg failed; exit 1; }
timeout -k 10 900 python tools/dp_opt_grid.py "rc_t0_ns=3000,rc_tg_ps=24000,rc_tw_ps=50,rc_cap=65536;rc_tw_ps=100;rc_tw_ps=200;rc_tw_ps=400;rc_tw_ps=50,rc_tg_ps=12000;rc_tg_ps=48000;rc_tg_ps=96000;rc_tg_ps=24000,rc_t0_ns=1500;rc_t0_ns=6000;rc_t0_ns=3000,rc_cap=16384;rc_cap=8192;rc_cap=65536,rc_tw_ps=150,rc_tg_ps=36000;rc_tw_ps=50,rc_tg_ps=24000,bp_nt_min_cells=1;bp_nt_min_cells=1000000000;bp_nt_min_cells=16384,warm_ahead=64;warm_ahead=256;warm_ahead=128" /tmp/c/mhc24.dpg 2>&1 | grep -v amdgpu | tee gpurun_out/r2_grid7.log
