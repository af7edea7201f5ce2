#!/bin/bash
# A/B: "ENV=val,... -- extra bench args" per argument, pipelined bench
set -o pipefail
mkdir -p gpurun_out
for cfg in "$@"; do
  envs=$(echo "${cfg%%--*}" | tr ',' ' ')
  extra=""
  case "$cfg" in *--*) extra="${cfg#*--}";; esac
  env $envs timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity $extra > gpurun_out/ab3.json 2> gpurun_out/ab3.err || exit 1
  python - <<PY
import json
d=json.loads(open('gpurun_out/ab3.json').read().strip().splitlines()[-1])
print('$cfg', '| images/s %.0f ms/step %.2f' % (d['value'], d['ms_per_step']))
PY
done
