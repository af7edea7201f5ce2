"""Counter view of tools/microbench/mfma_rate.out under `rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES
--kernel-trace` (tools/gpu_mfma_rate.sh): per kernel variant the mean dispatch duration, the effective clock
GRBM_GUI_ACTIVE / 8 / duration (the counter is summed over the 8 XCDs; MI355X_MICROARCH.md, DVFS give-back) and the matrix-pipe
share SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs) / (GRBM_GUI_ACTIVE / 8).
usage: mfma_rate_pmc.py <rocprofv3 output dir>"""
import collections
import csv
import glob
import re
import sys

d = sys.argv[1]
trace = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
dur = {}
name = {}
for r in csv.DictReader(open(trace)):
    if 'mfma_rate_kernel' in r['Kernel_Name']:
        dur[int(r['Dispatch_Id'])] = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
        name[int(r['Dispatch_Id'])] = r['Kernel_Name']
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        did = int(r['Dispatch_Id'])
        if did in dur:
            vals[name[did]][r['Counter_Name']].append((float(r['Counter_Value']), dur[did]))
LABEL = {('0', '4', '8'): 'reg2', ('0', '8', '4'): 'reg1', ('1', '4', '8'): 'lds384', ('1', '8', '4'): 'lds256', ('2', '4', '8'): 'lds384b'}
print('# rocprofv3 view of mfma_rate.out (random fp16 operands): last third of each variant\'s dispatches')
print('%-8s %6s %10s %9s %10s' % ('variant', 'n', 'ms', 'clock GHz', 'mfma_busy'))
for k, cs in vals.items():
    m = re.search(r'<(\d+), *(\d+), *(\d+), *(\d+)>', k) or re.search(r'ILi(\d)ELi(\d)ELi(\d)ELi(\d)E', k)
    lab = LABEL.get((m.group(1), m.group(3), m.group(4)), k[:40]) if m else k[:40]
    g = cs.get('GRBM_GUI_ACTIVE', [])
    b = cs.get('SQ_VALU_MFMA_BUSY_CYCLES', [])
    n = len(g)
    g = g[n - max(1, n // 3):]
    b = b[len(b) - max(1, len(b) // 3):] if b else []
    ghz = sum(v / 8.0 / t for v, t in g) / len(g)
    ms = sum(t for _, t in g) / len(g) / 1e6
    busy = (sum(v for v, _ in b) / len(b) / 4 / 256) / (sum(v for v, _ in g) / len(g) / 8) if b else float('nan')
    print('%-8s %6d %10.3f %9.3f %10.3f' % (lab, n, ms, ghz, busy))
