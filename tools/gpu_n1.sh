#!/bin/bash
# row n1 measured: the bench step fed by uint8 camera-style images (Resize + letterbox + normalise on the device) vs the fp32 feed, one box
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_preprocess.py -m gpu -q -x > gpurun_out/n1_test.log 2>&1 || { tail -20 gpurun_out/n1_test.log; exit 1; }
tail -2 gpurun_out/n1_test.log
: > gpurun_out/n1_input_path.txt
for r in 1 2; do
  for m in "--from-uint8 once" "--from-uint8 step"; do
    timeout -k 10 200 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-parity $m 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('round $r [$m] ms/step %.3f images/s %.0f detections/batch %d' % (d['ms_per_step'], d['value'], d['config']['detections_per_batch_rank0']))" | tee -a gpurun_out/n1_input_path.txt || exit 1
  done
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/n1_prof -o n1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity --from-uint8 step > $GRAFT_REPO_ROOT/gpurun_out/n1_prof.log 2>&1 || { tail -5 $GRAFT_REPO_ROOT/gpurun_out/n1_prof.log; exit 1; }
f=$(find /tmp/n1_prof -name "*kernel_stats.csv" | head -1)
cd $GRAFT_REPO_ROOT && { head -1 "$f"; grep "pre_" "$f"; } | cut -c1-200 | tee -a gpurun_out/n1_input_path.txt
