"""Timeline of the neck's launches in the LAST forward of a rocprofv3 --kernel-trace run (bench.py --serial): start / end relative to
the first neck launch, queue, and which launches overlap - do the replay lanes (rtm3d_op_schedule) run side by side?
usage: trace_lanes.py <rocprofv3 output dir>"""
import csv, sys, glob
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last forward: from the last stem launch on
stems = [i for i, r in enumerate(rows) if 'stem_fused' in r['Kernel_Name']]
fw = rows[stems[-1]:]
# neck = between the last backbone launch and the softmax apply
end = next(i for i, r in enumerate(fw) if 'softmax' in r['Kernel_Name'] and 'apply' in r['Kernel_Name'])
start = max(0, end - 13)
t0 = int(fw[start]['Start_Timestamp'])
for r in fw[start:end + 2]:
    s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    print('%-46s q=%-3s %9.1f .. %9.1f us  (%7.1f)' % (r['Kernel_Name'][:46], r.get('Queue_Id'), s / 1e3, e / 1e3, (e - s) / 1e3))
