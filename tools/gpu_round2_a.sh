#!/bin/bash
# GPU session A of round 2: new tests, world-1 torchrun rehearsal of the multi-GPU bench path, small-batch per-op profiles
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q -x > gpurun_out/r2_t2.log 2>&1; rc=$?
tail -4 gpurun_out/r2_t2.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-parity > gpurun_out/r2_b2_torchrun1.json 2> gpurun_out/r2_b2_torchrun1.err || { echo torchrun failed; tail -5 gpurun_out/r2_b2_torchrun1.err; }
timeout -k 10 200 python bench.py --batch 1 --steps 50 --warmup 10 --per-op --no-cpu-baseline --no-parity > gpurun_out/r2_b2_bs1.json 2> gpurun_out/r2_b2_bs1_perop.txt || exit 1
timeout -k 10 200 python bench.py --backbone RESNET-18 --batch 8 --steps 30 --warmup 5 --per-op --no-cpu-baseline --no-parity > gpurun_out/r2_b2_r18bs8.json 2> gpurun_out/r2_b2_r18bs8_perop.txt || exit 1
echo done
