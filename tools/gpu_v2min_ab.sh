for rep in 1 2 3; do for v in 200 512; do
  timeout -k 10 200 python3 bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-parity --no-sparse-probe --v2-min-tiles $v 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d['value'],1), round(d['ms_per_step'],3), d['config']['solver_form'], round(d['roofline']['backbone_ms'],3))" v2min$v || exit 1
done; done
