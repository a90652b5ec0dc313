#!/bin/bash
# Where the minimizer tile kernel spends its time: a measurement build (-DDG_TILE_DEBUG, never shipped) whose kernel returns after phase N
# (1 staging + 2-bit stream, 3 k-mer codes, 4 window minima, 5 runs + hashing, 0 everything), timed on the config-4 read set.
#   bash tools/tile_phases.sh          (GPU box, from the repo root; the numbers include ~0.1 ms of tile-descriptor prelude)
set -e
cd dipgenie_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DDG_TILE_DEBUG -mllvm -amdgpu-kernarg-preload-count=16 -shared \
  dg_api.hip dg_dp_tables.hip dg_dp_build.hip dg_dp_delta.hip dg_dp_sweep.hip dg_dp_trace.hip dg_dp_run.hip dg_sketch.hip dg_sketch_spectrum.hip dg_anchor.hip dg_hap.hip \
  -o ../../bin/lib_tiledbg.so
cd ../..
for v in 2 3 4 5 0; do echo "exit after phase $v"; DG_LIB=bin/lib_tiledbg.so DG_TILE_DEBUG=$v python3 tools/score_profile.py 0 2>&1 | grep -E "device events"; done
