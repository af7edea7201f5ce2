#!/bin/bash
# kernel tests + bs=32 per-op table (quick A/B of a kernel change)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x > gpurun_out/quick_test.log 2>&1; rc=$?
tail -2 gpurun_out/quick_test.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --per-op --no-cpu-baseline --no-parity > gpurun_out/$1_bs32.json 2> gpurun_out/$1_bs32.txt || exit 1
python - <<PY
import json
d=json.loads(open('gpurun_out/$1_bs32.json').read().strip().splitlines()[-1])
t=[l for l in open('gpurun_out/$1_bs32.txt') if l.startswith('forward total')]
print('bs32 images/s %.0f ms/step %.3f' % (d['value'], d['ms_per_step']), t[0].strip() if t else '')
PY
