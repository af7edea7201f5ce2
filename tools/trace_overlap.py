import csv, sys, glob
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
t0 = int(rows[0]['Start_Timestamp'])
d3 = [r for r in rows if 'decode3d' in r['Kernel_Name']]
print('cols', list(rows[0].keys()))
for r in d3[-3:]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    ov = [q for q in rows if q is not r and int(q['Start_Timestamp']) < e and int(q['End_Timestamp']) > s]
    print('decode3d %.3f..%.3f ms (%.3f ms) queue %s; overlapping kernels: %d' % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, r.get('Queue_Id'), len(ov)))
    for q in ov[:12]:
        print('    %-40s %.3f..%.3f q=%s' % (q['Kernel_Name'][:40], (int(q['Start_Timestamp']) - t0) / 1e6, (int(q['End_Timestamp']) - t0) / 1e6, q.get('Queue_Id')))
