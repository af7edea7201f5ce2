#!/bin/bash
# A/B of an environment switch: kernel tests matching $1, then per-op + pipelined bench for every value in $3.. of env var $2
set -o pipefail
mkdir -p gpurun_out
K="$1"; VAR="$2"; shift 2
timeout -k 10 400 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "$K" > gpurun_out/ab_test.log 2>&1; rc=$?
tail -5 gpurun_out/ab_test.log
if [ $rc -ne 0 ]; then exit 1; fi
for v in "$@"; do
  env $VAR=$v timeout -k 10 200 python bench.py --steps 20 --warmup 5 --per-op --no-cpu-baseline --no-parity > gpurun_out/ab_${VAR}_$v.json 2> gpurun_out/ab_${VAR}_$v.txt || exit 1
  python - <<PY
import json
d=json.loads(open('gpurun_out/ab_${VAR}_$v.json').read().strip().splitlines()[-1])
t=[l for l in open('gpurun_out/ab_${VAR}_$v.txt') if l.startswith('forward total')]
print('$VAR=$v', 'images/s %.0f ms/step %.2f' % (d['value'], d['ms_per_step']), t[0].strip() if t else '')
PY
done
