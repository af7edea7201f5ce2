#!/bin/bash
# GPU session G: decode3d wave priority; level4 on the persistent kernel; full tests
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q -x > gpurun_out/r2_t8.log 2>&1; rc=$?
tail -4 gpurun_out/r2_t8.log
if [ $rc -ne 0 ]; then exit 1; fi
B="--steps 20 --warmup 5 --no-cpu-baseline --no-parity"
timeout -k 10 200 python bench.py $B > gpurun_out/r2_g_norm.json 2> gpurun_out/r2_g.err || exit 1
timeout -k 10 200 python bench.py $B --diag-no-decode3d > gpurun_out/r2_g_norm_nod3.json 2>> gpurun_out/r2_g.err || exit 1
timeout -k 10 200 python bench.py $B --v2-min-tiles 200 > gpurun_out/r2_g_v2.json 2>> gpurun_out/r2_g.err || exit 1
timeout -k 10 200 python bench.py $B --heat-bias 2 > gpurun_out/r2_g_sat.json 2>> gpurun_out/r2_g.err || exit 1
timeout -k 10 200 python bench.py $B > gpurun_out/r2_g_norm2.json 2>> gpurun_out/r2_g.err || exit 1
echo done
