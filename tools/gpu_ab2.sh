#!/bin/bash
# A/B of environment settings given as arguments "VAR=val[,VAR2=val2]": pipelined bench (no per-op) for each
set -o pipefail
mkdir -p gpurun_out
for cfg in "$@"; do
  envs=$(echo $cfg | tr ',' ' ')
  env $envs timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity > gpurun_out/ab2.json 2> gpurun_out/ab2.err || exit 1
  python - <<PY
import json
d=json.loads(open('gpurun_out/ab2.json').read().strip().splitlines()[-1])
print('$cfg', 'images/s %.0f ms/step %.2f' % (d['value'], d['ms_per_step']))
PY
done
