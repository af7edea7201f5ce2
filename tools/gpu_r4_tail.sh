#!/bin/bash
# round 4: fused level-2 tail (conv64_root.hip): its kernel test, the DLA parity tests, then the bs=32 per-op table
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "level_tail or level_entry or conv_kernels" > gpurun_out/r4_tail_test.log 2>&1; rc=$?
tail -15 gpurun_out/r4_tail_test.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "dla or stages or config2" > gpurun_out/r4_tail_parity.log 2>&1; rc=$?
tail -5 gpurun_out/r4_tail_parity.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --per-op --no-cpu-baseline --no-parity > gpurun_out/r4_tail_bs32.json 2> gpurun_out/r4_tail_bs32.txt || exit 1
grep -E "backbone.level2|backbone.level3.down|forward total" gpurun_out/r4_tail_bs32.txt
python - <<PY
import json
d=json.loads(open('gpurun_out/r4_tail_bs32.json').read().strip().splitlines()[-1])
print('bs32 images/s %.0f ms/step %.3f backbone_ms %s frac %s' % (d['value'], d['ms_per_step'], d['roofline'].get('backbone_ms'), d['roofline'].get('backbone_frac')))
PY
