#!/bin/bash
# quick GPU check: kernel tests matching $1, then a per-op DLA-34 bs=32 bench into gpurun_out/$2.{json,txt}
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "$1" > gpurun_out/quick_test.log 2>&1; rc=$?
tail -5 gpurun_out/quick_test.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --per-op --no-cpu-baseline --no-parity > gpurun_out/$2.json 2> gpurun_out/$2.txt || exit 1
head -12 gpurun_out/$2.txt | grep -v amdgpu
tail -2 gpurun_out/$2.txt | cut -c1-120
