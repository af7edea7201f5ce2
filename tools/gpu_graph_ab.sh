#!/bin/bash
# A/B on one box: eager replay vs hipGraph replay of the bs=32 plan (interleaved, 3 rounds)
set -o pipefail
mkdir -p gpurun_out
for r in 1 2 3; do
  for m in "" "--graph"; do
    timeout -k 10 200 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-parity $m 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('round $r mode[$m] ms/step %.3f images/s %.0f' % (d['ms_per_step'], d['value']))" | tee -a gpurun_out/graph_ab.txt || exit 1
  done
done
