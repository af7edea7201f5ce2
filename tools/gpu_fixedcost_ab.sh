#!/bin/bash
# fixed cost per launch of the halo kernels: per-op time at several batch sizes for library builds rtm3d_amd/_C/<name>/ (one box)
for rep in 1 2; do for v in "$@"; do for b in 8 32; do
  timeout -k 10 120 python tools/gpu_variants.py rtm3d_amd/_C/$v/librtm3d_hip.so "level2.tree2.conv1|level2.tree1.conv2" $b 2>/dev/null | sed "s/^$v /$v B=$b /"
done; done; done
