"""profiles/r03_heads_clock.txt: the head convolutions on the benchmark weights against the same kernels on all-zero weights.
usage: heads_clock.py <per-op table> <pmc dir with kernel trace> <zero per-op table> <zero pmc dir with kernel trace>
Per op: hipEvent time of the un-profiled per-op pass -> TFLOP/s; under rocprofv3 (PMC pass with --kernel-trace): dispatch
duration and GRBM_GUI_ACTIVE of the SAME dispatches -> effective clock = GRBM_GUI_ACTIVE / 8 / duration (the counter is summed
over the 8 XCDs; MI355X_MICROARCH.md, DVFS give-back), MFMA-busy share where the pass collected it."""
import collections
import csv
import glob
import sys

KERNEL = ('conv_mfma256_persistent_kernel', 'conv_mfma256_halo_kernel', 'conv_mfma256_lattice_kernel')


def perop(path):
    rows = {}
    for l in open(path):
        p = l.split()
        if len(p) >= 5 and 'mfma256' in l and not l.startswith('{'):
            rows[p[0]] = (float(p[-3]), float(p[-2]))               # ms, TFLOP/s
    order = [l.split()[0] for l in open(path) if 'mfma256' in l and not l.startswith('{')]
    return rows, order


def pmc(d, order):
    trace = [f for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True)][0]
    dur = {int(r['Dispatch_Id']): (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) for r in csv.DictReader(open(trace))
           if any(k in r['Kernel_Name'] for k in KERNEL)}
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        per = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if any(k in r['Kernel_Name'] for k in KERNEL):
                per[r['Counter_Name']].append((int(r['Dispatch_Id']), float(r['Counter_Value'])))
        for c, lst in per.items():
            lst.sort()
            assert len(lst) % len(order) == 0, (c, len(lst), len(order))
            for i, (did, v) in enumerate(lst):
                vals[order[i % len(order)]][c].append((v, dur.get(did)))
    out = {}
    for op, cs in vals.items():
        g = [(v, t) for v, t in cs.get('GRBM_GUI_ACTIVE', []) if t]
        m = [(v, t) for v, t in cs.get('SQ_VALU_MFMA_BUSY_CYCLES', []) if t]
        ghz = sum(v / 8.0 / t for v, t in g) / len(g) if g else None
        ms = sum(t for _, t in g) / len(g) / 1e6 if g else None
        busy = None
        if m and g:
            busy = (sum(v for v, _ in m) / len(m) / 4 / 256) / (sum(v for v, _ in g) / len(g) / 8)
        out[op] = (ms, ghz, busy)
    return out


a, order = perop(sys.argv[1])
pa = pmc(sys.argv[2], order)
z, order_z = perop(sys.argv[3])
pz = pmc(sys.argv[4], order_z)
print('# head convolutions (and the other conv256 launches) of the bs=32 DLA-34 plan: benchmark weights vs ALL-ZERO weights (every activation 0)')
print('# un-profiled: hipEvent per-op pass of bench.py --per-op;  profiled: rocprofv3 --pmc GRBM_GUI_ACTIVE ... --kernel-trace, bench.py --serial')
print('# clock = GRBM_GUI_ACTIVE / 8 / dispatch duration of the same dispatches; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 x 256) / (GRBM_GUI_ACTIVE / 8)')
print('%-22s | %8s %9s | %8s %7s | %8s %9s | %8s %7s %9s | %s' % ('op', 'ms', 'TFLOP/s', 'prof ms', 'GHz', 'zero ms', 'TFLOP/s', 'prof ms', 'GHz', 'mfma_busy', 'zero/bench TFLOP/s'))
for op in order:
    if op not in z:
        continue
    ms, tf = a[op]
    zms, ztf = z[op]
    p, q = pa.get(op, (None,) * 3), pz.get(op, (None,) * 3)
    f = lambda v, n: ('%' + n) % v if v is not None else '-'
    print('%-22s | %8.3f %9.1f | %8s %7s | %8.3f %9.1f | %8s %7s %9s | %.2f' % (op[:22], ms, tf, f(p[0], '.3f'), f(p[1], '.2f'), zms, ztf,
                                                                                f(q[0], '.3f'), f(q[1], '.2f'), f(q[2], '.2f'), ztf / tf if tf else 0))
