#!/bin/bash
# A/B on one box: layers with one round of 256 x 256 tiles (DLA level4: 240) on the generic 128-pixel kernel (--v2-min-tiles 512,
# the rule up to round 2) or on the persistent conv256 kernel (default rule of plan.choose_variant), pipelined step with the 3D decode beside it
set -o pipefail
mkdir -p gpurun_out
for r in 1 2 3; do
  for m in "--v2-min-tiles 512" ""; do
    timeout -k 10 200 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-parity $m 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('round $r [$m] ms/step %.3f images/s %.0f' % (d['ms_per_step'], d['value']))" | tee -a gpurun_out/v2tiles_ab.txt || exit 1
  done
done
