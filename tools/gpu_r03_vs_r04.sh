#!/bin/bash
# Same-box comparison of the round-3 tree (git worktree _r03 at dc77af1, its own library) with this tree: per-stage times of the
# per-op pass and the pipelined ms/step, two interleaved rounds.
set -o pipefail
sum() { python - "$1" <<'PY'
import sys, re
st = {'backbone': 0.0, 'neck': 0.0, 'heads': 0.0}
for l in open(sys.argv[1]):
    f = l.split()
    if len(f) < 5 or not re.match(r'^[a-z]', l) or f[0] == 'op':
        continue
    try:
        ms = float(f[-3])
    except ValueError:
        continue
    k = 'backbone' if f[0].startswith('backbone') else 'heads' if f[0].startswith('heads') else 'neck' if f[0].startswith(('kfpn', 'fusion')) else None
    if k:
        st[k] += ms
print('backbone %.3f  neck %.3f  heads %.3f  sum %.3f' % (st['backbone'], st['neck'], st['heads'], sum(st.values())))
PY
}
for rep in 1 2; do for t in _r03 .; do
  ( cd $t && timeout -k 10 200 python bench.py --steps 30 --warmup 5 --per-op --no-cpu-baseline --no-parity --no-sparse-probe > /tmp/po.json 2> /tmp/po.txt ) || exit 1
  echo "$t per-op: $(sum /tmp/po.txt)   pipelined ms/step $(python -c "import json; print('%.3f' % json.loads(open('/tmp/po.json').read().strip().splitlines()[-1])['ms_per_step'])")"
done; done
