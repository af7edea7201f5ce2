#!/bin/bash
# ResNet-18 bs=8 (BASELINE config[1]): plan switches on / off on one box (pipelined ms/step)
set -o pipefail
run() { timeout -k 10 200 python bench.py --backbone RESNET-18 --batch 8 --steps 1500 --warmup 20 --no-cpu-baseline --no-parity --no-sparse-probe 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-40s ms/step %.3f' % (sys.argv[1], d['ms_per_step']))" "$1" || exit 1; }
for rep in 1 2; do
  RTM3D_S2D_ONLY=1 run "s2d_only=1"
  RTM3D_S2D_ONLY=0 run "s2d_only=0"
  RTM3D_S2D_ONLY=0 RTM3D_FOLD_NECK_UP=0 run "s2d_only=0 fold_neck_up=0"
done
