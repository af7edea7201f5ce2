#!/bin/bash
# quick GPU check: pytest expression $1 over tests/, then per-op + e2e DLA bench into gpurun_out/$2
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests -m gpu -q -x -k "$1" > gpurun_out/quick_test.log 2>&1; rc=$?
tail -4 gpurun_out/quick_test.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --per-op --no-cpu-baseline --no-parity > gpurun_out/$2.json 2> gpurun_out/$2.txt || exit 1
head -5 gpurun_out/$2.txt | grep -v amdgpu
python - <<PY
import json
d=json.loads(open('gpurun_out/$2.json').read().strip().splitlines()[-1])
t=[l for l in open('gpurun_out/$2.txt') if l.startswith('forward total')]
print('images/s %.0f ms/step %.2f' % (d['value'], d['ms_per_step']), t[0].strip() if t else '')
PY
