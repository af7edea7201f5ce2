#!/bin/bash
# round 4: conv64s2_halo (stride-2 entry conv): kernel tests, parity tests, per-op A/B on one box
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "conv_kernels" > gpurun_out/r4_s2_test.log 2>&1; rc=$?
tail -8 gpurun_out/r4_s2_test.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "golden or stages" > gpurun_out/r4_s2_parity.log 2>&1; rc=$?
tail -4 gpurun_out/r4_s2_parity.log
if [ $rc -ne 0 ]; then exit 1; fi
for v in 1 0 1 0; do
  export RTM3D_CONV64S2=$v
  timeout -k 10 120 python tools/gpu_variants.py rtm3d_amd/_C/librtm3d_hip.so "level3.tree1.tree1.conv1" 2>/dev/null | sed "s/^_C /s2=$v /" || exit 1
done
