#!/bin/bash
# GPU session B of round 2: ring kernel + band order A/B (per-op tables), tests
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q -x > gpurun_out/r2_t3.log 2>&1; rc=$?
tail -4 gpurun_out/r2_t3.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
B="--steps 10 --warmup 3 --per-op --no-cpu-baseline --no-parity"
RTM3D_CONV_RING=0 RTM3D_CONV256_BAND=0 timeout -k 10 200 python bench.py $B > gpurun_out/r2_b3_ring0_band0.json 2> gpurun_out/r2_b3_ring0_band0.txt || exit 1
RTM3D_CONV_RING=3 RTM3D_CONV256_BAND=0 timeout -k 10 200 python bench.py $B > gpurun_out/r2_b3_ring3_band0.json 2> gpurun_out/r2_b3_ring3_band0.txt || exit 1
RTM3D_CONV_RING=3 RTM3D_CONV256_BAND=32 timeout -k 10 200 python bench.py $B > gpurun_out/r2_b3_ring3_band32.json 2> gpurun_out/r2_b3_ring3_band32.txt || exit 1
RTM3D_CONV_RING=3 RTM3D_CONV256_BAND=16 timeout -k 10 200 python bench.py $B > gpurun_out/r2_b3_ring3_band16.json 2> gpurun_out/r2_b3_ring3_band16.txt || exit 1
RTM3D_CONV_RING=3 RTM3D_CONV256_BAND=8 timeout -k 10 200 python bench.py $B > gpurun_out/r2_b3_ring3_band8.json 2> gpurun_out/r2_b3_ring3_band8.txt || exit 1
RTM3D_CONV_RING=3 timeout -k 10 200 python bench.py --batch 1 --steps 50 --warmup 10 --per-op --no-cpu-baseline --no-parity > gpurun_out/r2_b3_bs1.json 2> gpurun_out/r2_b3_bs1.txt || exit 1
echo done
