#!/bin/bash
# Diagnostic: builds a -DLBW_PROF=1 copy of the library (the in-tree one stays untouched) and runs tools/prof_lbw.py
# on it.  Build HERE (no GPU needed):  bash tools/prof_lbw.sh build ; then  gpurun -- 'bash tools/prof_lbw.sh run'
set -e
cd $(dirname $0)/..
for k in 1 2 3; do
  D=$PWD/rtm3d_amd/_C/prof$k
  if [ "$1" = build ]; then
    mkdir -p $D/obj
    make -C rtm3d_amd/csrc OUT=$D/librtm3d_hip.so OBJ=$D/obj EXTRA=-DLBW_PROF=$k -j8 > $D/make.log 2>&1 || { tail -5 $D/make.log; exit 1; }
  else
    python tools/prof_lbw.py $D/librtm3d_hip.so $k
  fi
done
