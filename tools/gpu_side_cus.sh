#!/bin/bash
# pipelined ms/step against the number of CUs the 3D decode's side stream may use (rtm3d_stream_create_cumask), interleaved on one box;
# "nodecode" = the DIAGNOSTIC run without the 3D decode: what the forward + 2D decode alone take per step
set -o pipefail
cd $GRAFT_REPO_ROOT
B="--steps 60 --warmup 5 --no-cpu-baseline --no-parity --no-sparse-probe"
for rep in 1 2 3; do
  for cus in 0 8 12 16 24 32 48; do
    timeout -k 10 200 python3 bench.py $B --side-cus $cus 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('side_cus %3s  %.3f ms/step  %.0f images/s' % (sys.argv[1], d['ms_per_step'], d['value']))" $cus || exit 1
  done
  timeout -k 10 200 python3 bench.py $B --diag-no-decode3d 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nodecode      %.3f ms/step' % d['ms_per_step'])" || exit 1
done
