#!/bin/bash
# round 4: project fold (tap_dc) + fused level-2 tail: kernel tests, DLA parity tests, bs=32 per-op table with the fold at level3 on and off
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "project_fold or level_tail or conv_kernels" > gpurun_out/r4_fold_test.log 2>&1; rc=$?
tail -15 gpurun_out/r4_fold_test.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "dla or stages or config2" > gpurun_out/r4_fold_parity.log 2>&1; rc=$?
tail -5 gpurun_out/r4_fold_parity.log
if [ $rc -ne 0 ]; then exit 1; fi
for v in a b a b; do
  if [ $v = a ]; then export RTM3D_FOLD_C128=1; else export RTM3D_FOLD_C128=0; fi
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --per-op --no-cpu-baseline --no-parity --no-sparse-probe > gpurun_out/r4_fold_$v.json 2> gpurun_out/r4_fold_$v.txt || exit 1
  echo "== FOLD_C128=$RTM3D_FOLD_C128"
  grep -E "backbone.level[345].*(project|tree1.conv2|tree1.tree1.conv2|tree1.t)|forward total" gpurun_out/r4_fold_$v.txt | cut -c1-100
  python - <<PY
import json
d=json.loads(open('gpurun_out/r4_fold_$v.json').read().strip().splitlines()[-1])
print('bs32 images/s %.0f ms/step %.3f backbone_ms %.4f frac %.4f' % (d['value'], d['ms_per_step'], d['roofline'].get('backbone_ms'), d['roofline'].get('backbone_frac')))
PY
done
