#!/bin/bash
# Round 5: the bare MFMA rate of the box (tools/microbench/mfma_rate.hip) - plain run, the same on all-zero operands, and the
# counter view of the random-data run (GRBM_GUI_ACTIVE / 8 / duration = effective clock, SQ_VALU_MFMA_BUSY_CYCLES).
# The binary is built in the container (hipcc cross-compiles) and travels with the snapshot.
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/mfma_rate
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BIN=$R/tools/microbench/mfma_rate.out
timeout -k 10 200 $BIN 2.5 4000 0 > $O/random.txt 2>&1 || { tail -5 $O/random.txt; exit 1; }
cat $O/random.txt
timeout -k 10 200 $BIN 2.5 4000 1 > $O/zeros.txt 2>&1 || { tail -5 $O/zeros.txt; exit 1; }
cat $O/zeros.txt
timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc -o pmc -- $BIN 1.0 4000 0 > $O/pmc.log 2>&1 || { tail -5 $O/pmc.log; exit 1; }
find $O -name "*.db" -delete
python3 $R/tools/mfma_rate_pmc.py $O/pmc > $O/pmc_summary.txt 2>&1 || true
cat $O/pmc_summary.txt
