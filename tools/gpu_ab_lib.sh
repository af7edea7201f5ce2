#!/bin/bash
# A/B of two library builds (rtm3d_amd/_C/abA, abB) on one box, interleaved: bs=1 serial, bs=32 plain, bs=32 saturated
set -o pipefail
mkdir -p gpurun_out
run() { lib=$1; shift
timeout -k 10 200 python tools/ab_lib.py rtm3d_amd/_C/$lib/librtm3d_hip.so --no-cpu-baseline --no-parity "$@" > gpurun_out/sweep.json 2>/dev/null || exit 1
python - "$lib $*" <<PY
import json, sys
d=json.loads(open('gpurun_out/sweep.json').read().strip().splitlines()[-1])
print('%-50s ms/step %.3f' % (sys.argv[1], d['ms_per_step']))
PY
}
for rep in 1 2; do for lib in abA abB; do
run $lib --batch 1 --steps 200 --warmup 20 --serial
run $lib --steps 30 --warmup 6
run $lib --steps 15 --warmup 4 --heat-bias 2
done; done
