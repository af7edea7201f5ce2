#!/bin/bash
# soak: long pipelined runs of the three configurations (ticket counters, graph replay, per-slot decode streams) + the saturated top-k case
set -o pipefail
mkdir -p gpurun_out
run() { timeout -k 10 300 python bench.py --no-cpu-baseline --no-parity "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-70s ms/step %.3f images/s %.0f det %d' % (' '.join(sys.argv[1:]), d['ms_per_step'], d['value'], d['config']['detections_per_batch_rank0']))" "$@" | tee -a gpurun_out/soak.txt || exit 1; }
: > gpurun_out/soak.txt
run --steps 2000 --warmup 10
run --batch 1 --steps 20000 --warmup 50
run --backbone RESNET-18 --batch 8 --steps 5000 --warmup 20
run --steps 500 --warmup 10 --heat-bias 2
run --backbone RESNET-34 --batch 4 --steps 2000 --warmup 20
# round 3: the same with the peaks-only regression heads (patch plan + gather + finish per step), and the uint8 feed
run --steps 2000 --warmup 10 --sparse-heads
run --batch 1 --steps 10000 --warmup 50 --sparse-heads
run --backbone RESNET-18 --batch 8 --steps 3000 --warmup 20 --sparse-heads
run --steps 500 --warmup 10 --heat-bias 2 --sparse-heads
run --steps 500 --warmup 10 --from-uint8 step --sparse-heads
# round 6: the opt-in direct solver form next to the default (published) one
run --steps 1000 --warmup 10 --solver-form direct
run --batch 1 --steps 10000 --warmup 50 --solver-form direct
run --steps 500 --warmup 10 --heat-bias 2 --solver-form direct
