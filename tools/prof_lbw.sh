#!/bin/bash
# builds a -DLBW_PROF copy of the library into gpurun_out/prof_lib (the in-tree one stays untouched) and runs
# tools/prof_lbw.py on it:   gpurun -- 'bash tools/prof_lbw.sh'
set -e
cd $(dirname $0)/..
D=$PWD/gpurun_out/prof_lib
mkdir -p $D/obj
make -C rtm3d_amd/csrc OUT=$D/librtm3d_hip.so OBJ=$D/obj EXTRA=-DLBW_PROF -j8 > $D/make.log 2>&1 || { tail -5 $D/make.log; exit 1; }
python tools/prof_lbw.py $D/librtm3d_hip.so
rm -rf $D/obj
