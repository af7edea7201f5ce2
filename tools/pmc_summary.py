"""Summarise rocprofv3 --pmc counter CSVs per kernel name (mean per dispatch)."""
import csv, sys, glob, collections
out = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            out[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
names = sorted({c for k in out for c in out[k]})
print('kernel,dispatches,' + ','.join(names))
for k in sorted(out, key=lambda k: -sum(out[k].get('GRBM_GUI_ACTIVE', [0]))):
    n = max(len(v) for v in out[k].values())
    print('"%s",%d,' % (k, n) + ','.join('%.4g' % (sum(out[k][c]) / len(out[k][c])) if c in out[k] else '' for c in names))
