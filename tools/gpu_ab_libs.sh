#!/bin/bash
# tests given by -k expression against one library build, then tools/gpu_variants.sh on several:  gpu_ab_libs.sh '<pytest -k>' '<op regex>' lib1 lib2 ...
# The product library is put back when the script ends, however it ends (a later test or bench run must not silently use the last A/B build).
set -o pipefail
k=$1; pat=$2; shift 2
P=rtm3d_amd/_C/librtm3d_hip.so
cp $P $P.ab_backup || exit 1
trap 'mv -f $P.ab_backup $P' EXIT
for v in "$@"; do
  cp rtm3d_amd/_C/$v/librtm3d_hip.so $P
  timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -m gpu -q -x -k "$k" 2>&1 | tail -2 || exit 1
done
bash tools/gpu_variants.sh "$pat" "$@"
