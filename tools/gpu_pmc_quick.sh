#!/bin/bash
# one PMC pass (FETCH_SIZE + LDS conflict counters) of a short serial bench run; summary per kernel:  gpu_pmc_quick.sh [kernel-name regex]
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_quick; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc -o pmc -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-sparse-probe --serial > $O/pmc.log 2>&1 || { tail -5 $O/pmc.log; exit 1; }
python3 $R/tools/pmc_summary.py $O/pmc > $O/summary.csv
find $O -name "*.db" -delete
grep -E "${1:-.}" $O/summary.csv | cut -c1-150
