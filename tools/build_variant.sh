#!/bin/bash
# Builds a library variant for same-box A/Bs (tools/gpu_variants.sh): tools/build_variant.sh <name> [git-rev | -] [EXTRA flags]
#   rev given : the kernels as of that commit (sources copied to rtm3d_amd/_C/<name>/src), e.g. `build_variant.sh base HEAD`
#   rev "-"   : the working tree, e.g. `build_variant.sh timing - -DC256_STAMPS`
# -> rtm3d_amd/_C/<name>/librtm3d_hip.so (git-ignored; travels to the GPU box with the snapshot)
set -e
cd "$(dirname "$0")/.."
name=$1; rev=${2:--}; shift; shift || true
D=rtm3d_amd/_C/$name
mkdir -p $D
if [ "$rev" = "-" ]; then
  make -s -C rtm3d_amd/csrc -j8 OUT=../_C/$name/librtm3d_hip.so OBJ=../_C/$name/obj EXTRA="$*"
else
  rm -rf $D/src && mkdir -p $D/src
  git archive "$rev" rtm3d_amd/csrc | tar -x -C $D/src --strip-components=2
  # the sources include "../../include/rtm3d_hip.h" relative to rtm3d_amd/csrc: resolve it from there
  make -s -C $D/src -j8 OUT=../librtm3d_hip.so OBJ=../obj EXTRA="-I$(pwd)/rtm3d_amd/csrc $*"
fi
ls -la $D/librtm3d_hip.so
