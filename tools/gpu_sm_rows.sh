#!/bin/bash
# softmax apply pass: rows per workgroup (RTM3D_SM_APPLY_ROWS) and column segments per row (RTM3D_SM_XSPLIT), same box
set -o pipefail
for rep in 1 2; do for c in "2 1" "1 1" "1 2" "1 4"; do
  set -- $c
  RTM3D_SM_APPLY_ROWS=$1 RTM3D_SM_XSPLIT=$2 timeout -k 10 120 python tools/gpu_variants.py rtm3d_amd/_C/librtm3d_hip.so "softmax" 2>/dev/null | sed "s/^_C /rows=$1 xsplit=$2 /" || exit 1
done; done
