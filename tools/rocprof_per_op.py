"""Per-op durations of the conv256 kernels (one kernel name serves several ops) from a rocprofv3 --kernel-trace
CSV: dispatches are attributed to ops by their order inside a forward, as in tools/pmc_heads.py.
usage: rocprof_per_op.py <bench.py --per-op table> <..._kernel_trace.csv> > profiles/rNN_rocprof_conv256_per_op.csv"""
import csv, sys, collections
ops = [l.split()[0] for l in open(sys.argv[1]) if 'mfma256' in l and not l.startswith('{')]
rows = [r for r in csv.DictReader(open(sys.argv[2])) if 'conv_mfma256' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Dispatch_Id']))
assert len(rows) % len(ops) == 0, (len(rows), len(ops))
d = collections.defaultdict(list)
for i, r in enumerate(rows):
    d[ops[i % len(ops)]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
print('op,kernel,launches,avg_ms,min_ms,max_ms')
for i, o in enumerate(ops):
    v = d[o]
    print('%s,%s,%d,%.4f,%.4f,%.4f' % (o, rows[i]['Kernel_Name'].split('(')[0].replace('void ', ''), len(v), sum(v) / len(v), min(v), max(v)))
