#!/bin/bash
# pipelined ms/step of library builds, interleaved: ab_step.sh name1 name2 ...
cd $GRAFT_REPO_ROOT
P=rtm3d_amd/_C/librtm3d_hip.so
cp $P $P.ab_backup; trap 'mv -f $P.ab_backup $P' EXIT
for rep in 1 2 3; do for v in "$@"; do
  if [ "$v" = "product" ]; then cp $P.ab_backup $P; else cp rtm3d_amd/_C/$v/librtm3d_hip.so $P; fi
  timeout -k 10 200 python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-parity --no-sparse-probe 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d['value'],1), round(d['ms_per_step'],3))" $v
done; done
