#!/bin/bash
# round 4: level-2 tail without the ordinary copy of its output (conv64s2_halo reads the space-to-depth copy): tests + same-box A/B
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kernels.py -m gpu -q -x -k "neck_up_fold or config2 or golden or level_tail or conv_kernels or stages" > gpurun_out/r4_s2donly_test.log 2>&1; rc=$?
tail -8 gpurun_out/r4_s2donly_test.log
if [ $rc -ne 0 ]; then exit 1; fi
for v in 1 0 1 0; do
  export RTM3D_S2D_ONLY=$v
  timeout -k 10 120 python tools/gpu_variants.py rtm3d_amd/_C/librtm3d_hip.so "level2.tree2.conv2|level3.tree2.root|level4.tree2.root|level[45].downsample|level4.tree1.tree1.conv1|level5.tree1.conv1" 2>/dev/null | sed "s/^_C /s2d_only=$v /" || exit 1
done
