#!/bin/bash
# FETCH_SIZE per dispatch of one kernel (one counter per pass: FETCH_SIZE and WRITE_SIZE in one pass aborted the profiler) for several library builds (rocprofv3 --pmc, bench.py --serial):
#   tools/gpu_pmc_fetch_variants.sh KERNEL_SUBSTRING name1 name2 ...
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
K=$1; shift
P=$R/rtm3d_amd/_C/librtm3d_hip.so
cp $P $P.ab_backup; trap 'mv -f $P.ab_backup $P' EXIT
for v in "$@"; do
  cp $R/rtm3d_amd/_C/$v/librtm3d_hip.so $P
  rm -rf /tmp/pf_$v
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pf_$v -o p -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-parity --no-sparse-probe --serial > /tmp/pf_$v.log 2>&1 || { tail -3 /tmp/pf_$v.log; exit 1; }
  python3 - $v "$K" <<'PY'
import csv, glob, sys, collections
v, k = sys.argv[1], sys.argv[2]
f = glob.glob('/tmp/pf_%s/**/*counter_collection.csv' % v, recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if k in r['Kernel_Name']:
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
fs = acc['FETCH_SIZE']
print('%-8s %s: dispatches %d, FETCH_SIZE %.0f KB -> x 2 (gfx950 counts 64 B per 128-B request) = %.2f GB fetched per launch'
      % (v, k, len(fs), sum(fs) / len(fs), 2 * sum(fs) / len(fs) * 1024 / 1e9))
PY
done
