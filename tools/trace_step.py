"""Kernel timeline of the last pipelined step(s) of a rocprofv3 --kernel-trace run of bench.py: every launch from the second-to-last
stem on with start / end (us), queue, and the idle gap of the chip in front of it (no kernel of any queue running).
usage: trace_step.py <rocprofv3 output dir> [max rows]"""
import csv, sys, glob
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
stems = [i for i, r in enumerate(rows) if 'stem_fused' in r['Kernel_Name']]
fw = rows[stems[-3]:stems[-1]] if len(stems) >= 3 else rows
t0 = int(fw[0]['Start_Timestamp'])
busy_until = 0
for r in fw[:int(sys.argv[2]) if len(sys.argv) > 2 else 400]:
    s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    gap = s - busy_until if busy_until and s > busy_until else 0
    print('%-44s q=%-3s %9.1f .. %9.1f (%7.1f)%s' % (r['Kernel_Name'][:44], r.get('Queue_Id'), s / 1e3, e / 1e3, (e - s) / 1e3, '   IDLE %.1f us before' % (gap / 1e3) if gap > 5000 else ''))
    busy_until = max(busy_until, e)
