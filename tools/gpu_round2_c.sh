#!/bin/bash
# GPU session C: new preprocess tests; decode3d under saturated top-k (3200 objects/batch) with / without the 3D decode
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_preprocess.py -m gpu -q -x > gpurun_out/r2_t4.log 2>&1; rc=$?
tail -4 gpurun_out/r2_t4.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
B="--steps 10 --warmup 3 --no-cpu-baseline --no-parity"
export RTM3D_CONV_RING=0
timeout -k 10 200 python bench.py $B > gpurun_out/r2_c_norm.json 2> gpurun_out/r2_c_norm.err || exit 1
timeout -k 10 200 python bench.py $B --diag-no-decode3d > gpurun_out/r2_c_norm_nod3.json 2>> gpurun_out/r2_c_norm.err || exit 1
timeout -k 10 200 python bench.py $B --heat-bias 2 > gpurun_out/r2_c_sat.json 2> gpurun_out/r2_c_sat.err || exit 1
timeout -k 10 200 python bench.py $B --heat-bias 2 --diag-no-decode3d > gpurun_out/r2_c_sat_nod3.json 2>> gpurun_out/r2_c_sat.err || exit 1
timeout -k 10 200 python bench.py $B --heat-bias 2 --serial > gpurun_out/r2_c_sat_serial.json 2>> gpurun_out/r2_c_sat.err || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2_c_prof_sat -o sat -- python3 $GRAFT_REPO_ROOT/bench.py $B --heat-bias 2 > $GRAFT_REPO_ROOT/gpurun_out/r2_c_prof_sat.log 2>&1 || exit 1
echo done
