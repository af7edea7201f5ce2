#!/bin/bash
# bs=1: pipeline depth / side-stream count / side-stream priority sweep
set -o pipefail
mkdir -p gpurun_out
B="python bench.py --batch 1 --steps 200 --warmup 20 --no-cpu-baseline --no-parity"
for cfg in "2 1 -1" "2 2 -1" "3 1 -1" "3 3 -1" "3 3 0" "2 2 0" "4 4 0"; do
  set -- $cfg
  RTM3D_SIDE_PRIO=$3 timeout -k 10 200 $B --depth $1 --side-streams $2 > gpurun_out/sweep.json 2>/dev/null || exit 1
  python - <<PY
import json
d=json.loads(open('gpurun_out/sweep.json').read().strip().splitlines()[-1])
print('depth $1 sides $2 prio $3: ms/step %.3f' % d['ms_per_step'])
PY
done
