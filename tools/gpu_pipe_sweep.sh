#!/bin/bash
set -o pipefail
# pipeline settings sweep at bs=32 (side-stream priority, depth, decode streams, serial) - after the direct-form decode all pipelined variants are equal within noise (13.97-14.06 ms/step; serial 15.03-15.06)
mkdir -p gpurun_out
run() { pre="$1"; shift
env $pre timeout -k 10 200 python bench.py --no-cpu-baseline --no-parity --steps 40 --warmup 8 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-40s ms/step %.3f' % (' '.join(sys.argv[1:]), d['ms_per_step']))" "$pre" "$@" | tee -a gpurun_out/pipe_sweep.txt || exit 1; }
: > gpurun_out/pipe_sweep.txt
for r in 1 2; do
run X=1
run RTM3D_SIDE_PRIO=0
run X=1 --depth 3 --side-streams 1
run X=1 --depth 3 --side-streams 3
run X=1 --serial
done
