#!/bin/bash
# GPU session H: conv64 halo kernel: tests, per-op DLA-34 bs=32 and ResNet-18 bs=8/32
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x > gpurun_out/r2_t9a.log 2>&1; rc=$?
tail -4 gpurun_out/r2_t9a.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 600 python -m pytest tests -m gpu -q -x > gpurun_out/r2_t9.log 2>&1; rc=$?
tail -4 gpurun_out/r2_t9.log
if [ $rc -ne 0 ]; then exit 1; fi
B="--steps 20 --warmup 5 --per-op --no-cpu-baseline --no-parity"
timeout -k 10 200 python bench.py $B > gpurun_out/r2_h_dla.json 2> gpurun_out/r2_h_dla.txt || exit 1
timeout -k 10 200 python bench.py $B --backbone RESNET-18 --batch 8 > gpurun_out/r2_h_r18bs8.json 2> gpurun_out/r2_h_r18bs8.txt || exit 1
timeout -k 10 200 python bench.py $B --backbone RESNET-18 > gpurun_out/r2_h_r18bs32.json 2> gpurun_out/r2_h_r18bs32.txt || exit 1
echo done
