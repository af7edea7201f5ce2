#!/bin/bash
# clock + MFMA-busy of a kernel's longest dispatches for several library builds: tools/gpu_variant_clock.sh <kernel substring> name1 name2 ...
set -o pipefail
R=$GRAFT_REPO_ROOT
sub=$1; shift
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  O=$R/gpurun_out/vclock/$v
  rm -rf $O && mkdir -p $O
  timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $O -o pmc -- python3 $R/tools/gpu_variants.py $R/rtm3d_amd/_C/$v/librtm3d_hip.so "heads" > $O/log.txt 2>&1 || { tail -5 $O/log.txt; exit 1; }
  find $O -name "*.db" -delete
  python3 $R/tools/variant_clock.py $O "$sub" 8 $v
done
