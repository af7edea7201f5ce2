for rep in 1 2 3; do for p in -1 0; do
  RTM3D_SIDE_PRIO=$p timeout -k 10 200 python3 bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-parity --no-sparse-probe 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('side_prio', sys.argv[1], round(d['value'],1), round(d['ms_per_step'],3))" $p || exit 1
done; done
