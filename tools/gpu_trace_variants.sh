#!/bin/bash
# kernel timeline (rocprofv3 --kernel-trace) of the last pipelined steps of bench.py for several library builds:
#   tools/gpu_trace_variants.sh name1 name2 ...   -> gpurun_out/trace_<name>.txt (tools/trace_step.py)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
P=$R/rtm3d_amd/_C/librtm3d_hip.so
cp $P $P.ab_backup; trap 'mv -f $P.ab_backup $P' EXIT
for v in "$@"; do
  cp $R/rtm3d_amd/_C/$v/librtm3d_hip.so $P
  rm -rf /tmp/tr_$v
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$v -o t -- python3 $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-parity --no-sparse-probe > /tmp/tr_$v.log 2>&1 || { tail -5 /tmp/tr_$v.log; exit 1; }
  python3 $R/tools/trace_step.py /tmp/tr_$v 400 > $R/gpurun_out/trace_$v.txt
  grep -c . $R/gpurun_out/trace_$v.txt
done
