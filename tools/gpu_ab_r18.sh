#!/bin/bash
# A/B of two library builds (rtm3d_amd/_C/abA, abB) on one box, interleaved: ResNet-18 bs=8 and DLA-34 bs=32
set -o pipefail
mkdir -p gpurun_out
run() { lib=$1; shift
timeout -k 10 200 python tools/ab_lib.py rtm3d_amd/_C/$lib/librtm3d_hip.so --no-cpu-baseline --no-parity "$@" > gpurun_out/sweep.json 2>/dev/null || exit 1
python - "$lib $*" <<PY
import json, sys
d=json.loads(open('gpurun_out/sweep.json').read().strip().splitlines()[-1])
print('%-60s ms/step %.3f' % (sys.argv[1], d['ms_per_step']))
PY
}
for rep in 1 2 3; do for lib in abA abB; do
run $lib --backbone RESNET-18 --batch 8 --steps 100 --warmup 20
run $lib --steps 30 --warmup 6
done; done
