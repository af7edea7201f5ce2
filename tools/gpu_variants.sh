#!/bin/bash
# A/B/C... of library builds rtm3d_amd/_C/<name>/librtm3d_hip.so on one box: tools/gpu_variants.sh '<op regex>' name1 name2 ...  (3 interleaved rounds)
set -o pipefail
pat=$1; shift
for rep in 1 2 3; do for v in "$@"; do
  timeout -k 10 120 python tools/gpu_variants.py rtm3d_amd/_C/$v/librtm3d_hip.so "$pat" 2>/dev/null || exit 1
done; done
