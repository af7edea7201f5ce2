set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "fp32_verify" > gpurun_out/verify32_test.log 2>&1 || { tail -30 gpurun_out/verify32_test.log; exit 1; }
tail -3 gpurun_out/verify32_test.log
timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/s2_v32.json 2> gpurun_out/s2_v32.err || { tail -5 gpurun_out/s2_v32.err; exit 1; }
python - <<PY
import json
d=json.loads(open('gpurun_out/s2_v32.json').read().strip().splitlines()[-1])
print('images/s %.0f  ms/step %.2f' % (d['value'], d['ms_per_step']))
p=d['parity']
print({k: p[k] for k in p if not isinstance(p[k], dict) and k!='note'})
print('fp32 nat', p['e2e_fp32_mode'])
print('fp32 planted', p['e2e_fp32_mode_planted'])
print('fp16 planted', p['e2e_planted'])
PY
