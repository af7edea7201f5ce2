#!/bin/bash
# A/B of two library builds (rtm3d_amd/_C/abA, abB) on one box: per-op time of the ops matching $1 (grep pattern) + pipelined step, 3 interleaved rounds
set -o pipefail
mkdir -p gpurun_out
for rep in 1 2 3; do for lib in abA abB; do
  timeout -k 10 200 python tools/ab_lib.py rtm3d_amd/_C/$lib/librtm3d_hip.so --no-cpu-baseline --no-parity --steps 30 --warmup 6 --per-op > gpurun_out/ab_perop.json 2> gpurun_out/ab_perop.txt || exit 1
  python - "$lib" "$1" <<PY
import json, sys, re
d=json.loads(open('gpurun_out/ab_perop.json').read().strip().splitlines()[-1])
ops=[l.split() for l in open('gpurun_out/ab_perop.txt') if re.search(sys.argv[2], l)]
print('%s  ms/step %.3f  %s' % (sys.argv[1], d['ms_per_step'], '  '.join('%s %s' % (o[0], o[-3]) for o in ops)))
PY
done; done
