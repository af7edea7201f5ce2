#!/bin/bash
# full GPU check: all -m gpu tests, default bench (with cpu baseline + parity) into gpurun_out/$1.json, per-op table
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -q -x > gpurun_out/full_test.log 2>&1; rc=$?
tail -4 gpurun_out/full_test.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print(\"smoke ok\")" 2>&1 | tail -2 || exit 1
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/$1.json 2> gpurun_out/$1.err || { tail -5 gpurun_out/$1.err; exit 1; }
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --per-op --no-cpu-baseline --no-parity > gpurun_out/$1_perop.json 2> gpurun_out/$1_perop.txt || exit 1
tail -1 gpurun_out/$1_perop.txt | cut -c1-100
python - <<PY
import json
d=json.loads(open('gpurun_out/$1.json').read().strip().splitlines()[-1])
print('images/s %.0f  ms/step %.2f  frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['frac']))
print({k: d['parity'][k] for k in ('stage_box_linf','e2e_vert_linf_px','e2e_box_linf','matched','missed')})
print(d['cpu_baseline']['value'])
PY
