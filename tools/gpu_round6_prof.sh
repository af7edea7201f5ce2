#!/bin/bash
# GPU profiling session (round 6): per-op table, rocprofv3 kernel trace + stats (CSV), three PMC passes, and the head convs on
# all-zero weights (what the same kernels do when the MFMA operands carry no energy): per-op table + GRBM_GUI_ACTIVE with
# kernel durations of the same run -> effective clock = GRBM_GUI_ACTIVE / 8 / duration (MI355X_MICROARCH.md, DVFS give-back)
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6_prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# provenance of everything collected here (tools/pmc_heads.py copies it into the summary; bench.py prints it with `roofline.traffic`)
python3 - <<PYEOF
import json, sys, time
sys.path.insert(0, "$R")
import bench
json.dump({"collected_utc": time.strftime("%Y-%m-%dT%H:%M:%S", time.gmtime()), "csrc_sha1": bench.csrc_sha1()}, open("$O/provenance.json", "w"))
PYEOF
B="--steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-sparse-probe"
timeout -k 10 200 python3 $R/bench.py $B --per-op > $O/bench_n1.json 2> $O/bench_per_op.txt || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o trace -- python3 $R/bench.py $B > $O/trace.log 2>&1 || { tail -5 $O/trace.log; exit 1; }
echo trace done
P="--steps 5 --warmup 2 --no-cpu-baseline --no-parity --no-sparse-probe --serial"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o pmc -- python3 $R/bench.py $P > $O/pmc_fetch.log 2>&1 || { tail -5 $O/pmc_fetch.log; exit 1; }
echo pmc1 done
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_write -o pmc -- python3 $R/bench.py $P > $O/pmc_write.log 2>&1 || { tail -5 $O/pmc_write.log; exit 1; }
echo pmc2 done
timeout -k 10 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $O/pmc_sq -o pmc -- python3 $R/bench.py $P > $O/pmc_sq.log 2>&1 || { tail -5 $O/pmc_sq.log; exit 1; }
echo pmc3 done
# zeros
timeout -k 10 200 python3 $R/bench.py $B --per-op --zero-weights > $O/zero_n1.json 2> $O/zero_per_op.txt || exit 1
timeout -k 10 400 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_zero -o pmc -- python3 $R/bench.py $P --zero-weights > $O/pmc_zero.log 2>&1 || { tail -5 $O/pmc_zero.log; exit 1; }
echo zero done
# keep only the small CSVs (the merge back is limited to 64 MiB)
find $O -name "*.db" -delete
du -sh $O
