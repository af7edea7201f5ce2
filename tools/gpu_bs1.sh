#!/bin/bash
# bs=1 latency diagnosis: plain / no-decode3d / serial runs and a kernel trace of the plain run
set -o pipefail
mkdir -p gpurun_out
B="python bench.py --batch 1 --steps 200 --warmup 20 --no-cpu-baseline --no-parity"
timeout -k 10 200 $B > gpurun_out/$1_plain.json 2>/dev/null || exit 1
timeout -k 10 200 $B --diag-no-decode3d > gpurun_out/$1_nod3.json 2>/dev/null || exit 1
timeout -k 10 200 $B --serial > gpurun_out/$1_serial.json 2>/dev/null || exit 1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_bs1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_bs1 -o bs1 --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --batch 1 --steps 200 --warmup 20 --no-cpu-baseline --no-parity > $GRAFT_REPO_ROOT/gpurun_out/$1_prof.log 2>&1 || exit 1
cd $GRAFT_REPO_ROOT
find /tmp/prof_bs1 -name '*kernel_stats.csv' -exec cp {} gpurun_out/$1_kernel_stats.csv \;
for f in plain nod3 serial; do python - <<PY
import json
d=json.loads(open('gpurun_out/$1_$f.json').read().strip().splitlines()[-1])
print('$f', 'images/s %.0f ms/step %.3f' % (d['value'], d['ms_per_step']))
PY
done
