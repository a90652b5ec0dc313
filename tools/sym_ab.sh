#!/bin/bash
# GPU box: parity of the symmetric sweep form, then its A/B on a chr22-style panel (BASELINE configs[4] geometry, 1 Mbp: the lattice is
# resident) and on the bench workload's graph (synthetic MHC-24).   bash tools/sym_ab.sh [c5_backbone_bp] ["k=v,..." ...]
set -o pipefail
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export HIP_FORCE_DEV_KERNARG=1 DG_LIB=$PWD/bin/libdipgenie_hip_sym.so   # the measurement build (make -C dipgenie_amd/csrc sym)
BP=${1:-1000000}; shift || true
timeout -k 10 900 env -u DG_LIB python -m pytest tests/test_gpu_parity.py -x -q -k "symmetric" --timeout 300 > gpurun_out/sym_tests.log 2>&1; tail -3 gpurun_out/sym_tests.log
grep -q passed gpurun_out/sym_tests.log && ! grep -q failed gpurun_out/sym_tests.log || { grep -B40 Error gpurun_out/sym_tests.log | tail -80; exit 1; }
D=/tmp/c5ab; mkdir -p $D
python3 tools/c5_gen.py $BP $D > gpurun_out/sym_gen.log 2>&1
bin/DipGenie -t16 -p2 -R32 -g $D/c5.gfa -r $D/c5.fa -o $D/dump.fa -D $D/c5 -X > /dev/null 2>&1
SETS=("$@"); [ ${#SETS[@]} -eq 0 ] && SETS=("sym=0" "sym=1,sym_rc=4" "sym=1,sym_rc=2" "sym=1,sym_rc=6" "sym=1,sym_rc=8" "sym=1,sym_rc=3")
timeout -k 10 500 python tools/dp_opt_ab.py $D/c5.dpg "${SETS[@]}" 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/sym_ab_c5.txt
bash tools/mhc24_dpg.sh > gpurun_out/dpg.log 2>&1 || { echo dpg failed; tail -5 gpurun_out/dpg.log; exit 1; }
timeout -k 10 400 python tools/dp_opt_ab.py /tmp/c/mhc24.dpg "${SETS[@]}" 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/sym_ab_mhc24.txt
