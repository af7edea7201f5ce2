#!/bin/bash
# GPU session D: decode3d with 16 objects per workgroup + wave-aggregated NMS append: tests, normal and saturated benches
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q -x > gpurun_out/r2_t5.log 2>&1; rc=$?
tail -4 gpurun_out/r2_t5.log
if [ $rc -ne 0 ]; then exit 1; fi
B="--steps 10 --warmup 3 --no-cpu-baseline --no-parity"
export RTM3D_CONV_RING=0
timeout -k 10 200 python bench.py $B > gpurun_out/r2_d_norm.json 2> gpurun_out/r2_d_norm.err || exit 1
timeout -k 10 200 python bench.py $B --diag-no-decode3d > gpurun_out/r2_d_norm_nod3.json 2>> gpurun_out/r2_d_norm.err || exit 1
timeout -k 10 200 python bench.py $B --heat-bias 2 > gpurun_out/r2_d_sat.json 2> gpurun_out/r2_d_sat.err || exit 1
timeout -k 10 200 python bench.py $B --heat-bias 2 --diag-no-decode3d > gpurun_out/r2_d_sat_nod3.json 2>> gpurun_out/r2_d_sat.err || exit 1
timeout -k 10 200 python bench.py $B --heat-bias 2 --serial > gpurun_out/r2_d_sat_serial.json 2>> gpurun_out/r2_d_sat.err || exit 1
echo done
