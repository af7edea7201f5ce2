cd $GRAFT_REPO_ROOT
P=rtm3d_amd/_C/librtm3d_hip.so
cp $P $P.ab_backup; trap 'mv -f $P.ab_backup $P' EXIT
for rep in 1 2 3; do for v in pre1l cur; do
  cp rtm3d_amd/_C/$v/librtm3d_hip.so $P
  for cfg in "--batch 1 --steps 3000 --warmup 50" "--backbone RESNET-18 --batch 8 --steps 1000 --warmup 20"; do
  timeout -k 10 200 python3 bench.py $cfg --no-cpu-baseline --no-parity --no-sparse-probe 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], sys.argv[2:4], round(d['value'],1), round(d['ms_per_step'],4))" $v $cfg || exit 1
  done
done; done
