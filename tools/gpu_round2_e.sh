#!/bin/bash
# GPU session E: decode3d objects per workgroup A/B (8 / 12 / 16), normal and saturated
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "decode3d or planted or pipeline" > gpurun_out/r2_t6.log 2>&1; rc=$?
tail -3 gpurun_out/r2_t6.log
if [ $rc -ne 0 ]; then exit 1; fi
B="--steps 10 --warmup 3 --no-cpu-baseline --no-parity"
export RTM3D_CONV_RING=0
for w in 8 12 16; do
  RTM3D_D3_WPB=$w timeout -k 10 200 python bench.py $B > gpurun_out/r2_e_norm_w$w.json 2> gpurun_out/r2_e.err || exit 1
  RTM3D_D3_WPB=$w timeout -k 10 200 python bench.py $B --heat-bias 2 > gpurun_out/r2_e_sat_w$w.json 2>> gpurun_out/r2_e.err || exit 1
done
timeout -k 10 200 python bench.py $B --diag-no-decode3d > gpurun_out/r2_e_norm_nod3.json 2>> gpurun_out/r2_e.err || exit 1
timeout -k 10 200 python bench.py $B --heat-bias 2 --diag-no-decode3d > gpurun_out/r2_e_sat_nod3.json 2>> gpurun_out/r2_e.err || exit 1
python tools/gpu_d3.py > gpurun_out/r2_e_d3_alone.txt 2>&1 || true
echo done
