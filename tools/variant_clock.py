"""Clock and matrix-pipe share of the longest dispatches of one kernel under `rocprofv3 --pmc GRBM_GUI_ACTIVE
SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -- python tools/gpu_variants.py <lib> <regex>` (tools/gpu_variant_clock.sh):
usage: variant_clock.py <rocprofv3 output dir> <kernel name substring> [n longest]
clock = GRBM_GUI_ACTIVE / 8 / duration; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 x 256) / (GRBM_GUI_ACTIVE / 8)."""
import collections
import csv
import glob
import sys

d, sub = sys.argv[1], sys.argv[2]
nlong = int(sys.argv[3]) if len(sys.argv) > 3 else 8
trace = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
dur = {int(r['Dispatch_Id']): int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in csv.DictReader(open(trace)) if sub in r['Kernel_Name']}
vals = collections.defaultdict(dict)
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        did = int(r['Dispatch_Id'])
        if did in dur:
            vals[did][r['Counter_Name']] = float(r['Counter_Value'])
longest = sorted(dur, key=lambda k: -dur[k])[:nlong]
ms = sum(dur[k] for k in longest) / len(longest) / 1e6
g = sum(vals[k]['GRBM_GUI_ACTIVE'] for k in longest) / len(longest)
b = sum(vals[k].get('SQ_VALU_MFMA_BUSY_CYCLES', float('nan')) for k in longest) / len(longest)
print('%-10s %s: %d longest dispatches: %.3f ms  clock %.3f GHz  mfma_busy %.3f' % (sys.argv[4] if len(sys.argv) > 4 else '', sub, len(longest), ms, g / 8 / (ms * 1e6), b / 1024 / (g / 8)))
