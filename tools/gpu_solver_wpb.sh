#!/bin/bash
# Round 6: what the published (SciPy-faithful, now default) solver form costs per pipelined bs=32 step against objects per workgroup
# (library builds with -DD3_WPB_PUB=n: tools/build_variant.sh pubN - -DD3_WPB_PUB=N) and against the direct form, interleaved on one box.
#   tools/gpu_solver_wpb.sh name1 name2 ...    ("product" = the library in place; "direct" = the product library with --solver-form direct)
cd $GRAFT_REPO_ROOT
P=rtm3d_amd/_C/librtm3d_hip.so
cp $P $P.ab_backup; trap 'mv -f $P.ab_backup $P' EXIT
for rep in 1 2 3; do for v in "$@"; do
  extra=""
  if [ "$v" = "product" ]; then cp $P.ab_backup $P; elif [ "$v" = "direct" ]; then cp $P.ab_backup $P; extra="--solver-form direct"; else cp rtm3d_amd/_C/$v/librtm3d_hip.so $P; fi
  timeout -k 10 200 python3 bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-parity --no-sparse-probe $extra 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d['value'],1), round(d['ms_per_step'],3), d['config']['solver_form'])" $v || exit 1
done; done
