#!/bin/bash
# decode3d change check: all GPU tests, then bs=1 plain/serial and bs=32 plain/no-decode/saturated steps
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -q -x > gpurun_out/quick_test.log 2>&1; rc=$?
tail -4 gpurun_out/quick_test.log
if [ $rc -ne 0 ]; then exit 1; fi
run() { timeout -k 10 200 python bench.py --no-cpu-baseline --no-parity "$@" > gpurun_out/sweep.json 2>/dev/null || exit 1
python - "$*" <<PY
import json, sys
d=json.loads(open('gpurun_out/sweep.json').read().strip().splitlines()[-1])
print('%-60s ms/step %.3f  images/s %.0f' % (sys.argv[1], d['ms_per_step'], d['value']))
PY
}
run --batch 1 --steps 200 --warmup 20
run --batch 1 --steps 200 --warmup 20 --serial
run --steps 40 --warmup 8
run --steps 40 --warmup 8 --diag-no-decode3d
run --steps 40 --warmup 8
run --steps 20 --warmup 5 --heat-bias 2
run --steps 20 --warmup 5 --heat-bias 2 --diag-no-decode3d
