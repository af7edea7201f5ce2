#!/bin/bash
# kernel tests, then the bs=1 DLA-34 per-op table and the plain pipelined step
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x > gpurun_out/quick_test.log 2>&1; rc=$?
tail -4 gpurun_out/quick_test.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 200 python bench.py --batch 1 --steps 200 --warmup 20 --per-op --no-cpu-baseline --no-parity > gpurun_out/$1_bs1.json 2> gpurun_out/$1_bs1.txt || exit 1
python - <<PY
import json
d=json.loads(open('gpurun_out/$1_bs1.json').read().strip().splitlines()[-1])
t=[l for l in open('gpurun_out/$1_bs1.txt') if l.startswith('forward total')]
print('bs1', 'images/s %.0f ms/step %.3f' % (d['value'], d['ms_per_step']), t[0].strip() if t else '')
PY
