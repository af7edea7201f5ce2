#!/bin/bash
# GPU session F: 256x128 ring kernel for the backbone; level4 on the persistent 256x256 kernel; A/B per-op tables
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q -x > gpurun_out/r2_t7.log 2>&1; rc=$?
tail -4 gpurun_out/r2_t7.log
if [ $rc -ne 0 ]; then exit 1; fi
B="--steps 10 --warmup 3 --per-op --no-cpu-baseline --no-parity"
for r in 0 1 3 7; do
  RTM3D_CONV_RING=$r timeout -k 10 200 python bench.py $B > gpurun_out/r2_f_ring$r.json 2> gpurun_out/r2_f_ring$r.txt || exit 1
done
RTM3D_CONV_RING=3 RTM3D_V2_MIN_TILES=200 timeout -k 10 200 python bench.py $B > gpurun_out/r2_f_ring3_v2.json 2> gpurun_out/r2_f_ring3_v2.txt || exit 1
RTM3D_CONV_RING=3 timeout -k 10 200 python bench.py --batch 1 --steps 50 --warmup 10 --per-op --no-cpu-baseline --no-parity > gpurun_out/r2_f_bs1.json 2> gpurun_out/r2_f_bs1.txt || exit 1
echo done
