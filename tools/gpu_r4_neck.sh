#!/bin/bash
# round 4: neck up-fold (kfpn_up3 + proj3+head2 in one launch): tests, then per-op A/B on one box
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "level_tail" > gpurun_out/r4_neck_test.log 2>&1; rc=$?
tail -8 gpurun_out/r4_neck_test.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "neck_up_fold or config2 or golden" > gpurun_out/r4_neck_parity.log 2>&1; rc=$?
tail -12 gpurun_out/r4_neck_parity.log
if [ $rc -ne 0 ]; then exit 1; fi
for v in 1 0 1 0; do
  export RTM3D_FOLD_NECK_UP=$v
  timeout -k 10 120 python tools/gpu_variants.py rtm3d_amd/_C/librtm3d_hip.so "kfpn_up|kfpn_proj|level2.tree2.conv2|tree2.root" 2>/dev/null | sed "s/^_C /fold=$v /" || exit 1
done
