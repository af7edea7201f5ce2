#!/bin/bash
# per-launch fixed cost of the persistent conv256 kernels: per-op time of the head convs / deconv phases against the batch size
set -o pipefail
mkdir -p gpurun_out
: > gpurun_out/fixedcost.txt
for B in 4 8 12 16 24 32; do
  timeout -k 10 200 python bench.py --batch $B --steps 6 --warmup 2 --per-op --no-cpu-baseline --no-parity > /dev/null 2> gpurun_out/fc_$B.txt || exit 1
  for op in heads.conv_d6 heads.conv_d1 kfpn_up3 fusion_up5.2 kfpn_softmax_fuse heads.out_convs; do
    grep "^$op " gpurun_out/fc_$B.txt | awk -v B=$B '{print "B=" B, $1, $2, $3}' >> gpurun_out/fixedcost.txt
  done
done
cat gpurun_out/fixedcost.txt
