#!/bin/bash
# A/B on one box: output-channel tile of the 128-pixel conv kernel for the generic-kernel ops of the DLA-34 bs=32 plan
# (bench.py --bn-tile OP=N, per-op table).  Prints ms of the affected ops for the default choice (128 channels where the layer has them)
# and for 64-channel tiles.  Round 3: 64 loses everywhere (level5 3x3s 0.085 -> 0.122 ms, stride-2 entries 0.065 -> 0.073, roots 0.057 -> 0.069).
set -o pipefail
mkdir -p gpurun_out
OPS="backbone.level3.tree1.tree1.conv1 backbone.level4.tree1.tree1.conv1 backbone.level5.tree1.conv1 backbone.level5.tree1.conv2 backbone.level5.tree2.conv1 backbone.level5.tree2.conv2 backbone.level2.root backbone.level3.tree1.root backbone.level3.tree2.root backbone.level5.root backbone.level3.tree1.project backbone.level4.tree1.project backbone.level5.project kfpn_head5"
run() { tag=$1; shift
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --per-op "$@" > gpurun_out/bn.json 2> gpurun_out/bn.txt || exit 1
  python - "$tag" <<PY
import sys
rows=[l.split() for l in open('gpurun_out/bn.txt') if 'conv1x1_mfma ' in l or 'conv3x3_mfma ' in l]
print('%-8s' % sys.argv[1], ' '.join('%s' % r[-3] for r in rows))
PY
}
for rep in 1 2; do
  run default
  A=""; for o in $OPS; do A="$A --bn-tile $o=64"; done
  run bn64 $A
done
python - <<PY
rows=[l.split() for l in open('gpurun_out/bn.txt') if 'conv1x1_mfma ' in l or 'conv3x3_mfma ' in l]
print('ops     ', ' '.join(r[0][-12:] for r in rows))
PY
