#!/bin/bash
# kernel + parity tests, then bs=1 DLA and bs=8 R18 per-op benches
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -q -x > gpurun_out/quick_test.log 2>&1; rc=$?
tail -4 gpurun_out/quick_test.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 200 python bench.py --batch 1 --steps 100 --warmup 20 --per-op --no-cpu-baseline --no-parity > gpurun_out/$1_bs1.json 2> gpurun_out/$1_bs1.txt || exit 1
timeout -k 10 200 python bench.py --backbone RESNET-18 --batch 8 --steps 50 --warmup 10 --per-op --no-cpu-baseline --no-parity > gpurun_out/$1_r18bs8.json 2> gpurun_out/$1_r18bs8.txt || exit 1
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --per-op --no-cpu-baseline --no-parity > gpurun_out/$1_bs32.json 2> gpurun_out/$1_bs32.txt || exit 1
for f in bs1 r18bs8 bs32; do python - <<PY
import json
d=json.loads(open('gpurun_out/$1_$f.json').read().strip().splitlines()[-1])
t=[l for l in open('gpurun_out/$1_$f.txt') if l.startswith('forward total')]
print('$f', 'images/s %.0f ms/step %.3f' % (d['value'], d['ms_per_step']), t[0].strip() if t else '')
PY
done
