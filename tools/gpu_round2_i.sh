#!/bin/bash
# GPU session I: fused DLA stem
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "fused_dla_stem" > gpurun_out/r2_t10a.log 2>&1; rc=$?
tail -6 gpurun_out/r2_t10a.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 600 python -m pytest tests -m gpu -q -x > gpurun_out/r2_t10.log 2>&1; rc=$?
tail -4 gpurun_out/r2_t10.log
if [ $rc -ne 0 ]; then exit 1; fi
B="--steps 20 --warmup 5 --per-op --no-cpu-baseline --no-parity"
timeout -k 10 200 python bench.py $B > gpurun_out/r2_i_dla.json 2> gpurun_out/r2_i_dla.txt || exit 1
timeout -k 10 200 python bench.py --batch 1 --steps 50 --warmup 10 --per-op --no-cpu-baseline --no-parity > gpurun_out/r2_i_bs1.json 2> gpurun_out/r2_i_bs1.txt || exit 1
echo done
