#!/bin/bash
# Round 5: in-kernel timeline of heads.conv_d6 (persistent 256 x 256 convolution).  Build first (in the container):
#   make -C rtm3d_amd/csrc OUT=../_C/stamps/librtm3d_hip.so OBJ=../_C/stamps/obj EXTRA=-DC256_STAMPS
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R && timeout -k 10 300 python3 tools/gpu_c256_stamps.py rtm3d_amd/_C/stamps/librtm3d_hip.so 32 2>&1 | grep -v amdgpu.ids | tee gpurun_out/c256_stamps.txt
