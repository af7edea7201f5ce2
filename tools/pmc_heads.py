"""Per-op PMC figures for the ops that share one kernel name (the persistent 256x256 conv kernel runs 11
ops per forward): dispatches of that kernel are attributed to ops by their order inside a forward.
usage: pmc_heads.py <per-op table of bench.py --per-op> <pmc dir> [<pmc dir> ...] > profiles/rNN_pmc_heads.json"""
import csv, sys, glob, json, collections
perop, dirs = sys.argv[1], sys.argv[2:]
ops = [l.split()[0] for l in open(perop) if 'mfma256' in l and not l.startswith('{')]
KERNEL = ('conv_mfma256_persistent_kernel', 'conv_mfma256_halo_kernel', 'conv_mfma256_lattice_kernel')
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for d in dirs:
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if any(k in r['Kernel_Name'] for k in KERNEL)]
        per_counter = collections.defaultdict(list)
        for r in rows:
            per_counter[r['Counter_Name']].append((int(r['Dispatch_Id']), float(r['Counter_Value'])))
        for c, lst in per_counter.items():
            lst.sort()
            assert len(lst) % len(ops) == 0, (c, len(lst), len(ops))
            for i, (_, v) in enumerate(lst):
                vals[ops[i % len(ops)]][c].append(v)
out = {'source': 'rocprofv3 --pmc, 3 separate passes (FETCH_SIZE | WRITE_SIZE GRBM_GUI_ACTIVE | SQ_*), bench.py --serial bs=32 DLA-34, MI355X; '
                 'dispatches of %s attributed to ops by order within a forward (%s)' % (' / '.join(KERNEL), ', '.join(ops)),
       'correction': 'traffic = (2*FETCH_SIZE + WRITE_SIZE) KB * 1024: gfx950 FETCH_SIZE counts 64 B per 128-B request (MI355X_MICROARCH.md, HBM)',
       'kernels': {}}
# provenance (round 5): when and on which kernel sources the passes ran (provenance.json, written on the GPU box by the profiling
# script beside the pass directories) and the commit this summary was made on: bench.py prints them in `roofline.traffic_source`
import os, subprocess
for d in dirs:
    for cand in (os.path.join(d, 'provenance.json'), os.path.join(os.path.dirname(d.rstrip('/')), 'provenance.json')):
        if os.path.exists(cand):
            out.update(json.load(open(cand)))
            break
try:
    out['commit'] = subprocess.check_output(['git', 'rev-parse', '--short', 'HEAD'], cwd=os.path.dirname(os.path.abspath(__file__)), text=True).strip()
except Exception:
    pass
for op in ops:
    k = {c: sum(v) / len(v) for c, v in vals[op].items()}
    if 'FETCH_SIZE' in k and 'WRITE_SIZE' in k:
        k['hbm_bytes_corrected'] = (2 * k['FETCH_SIZE'] + k['WRITE_SIZE']) * 1024
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in k and 'GRBM_GUI_ACTIVE' in k:
        # MFMA busy is summed over the 4 SIMDs of 256 CUs, GRBM_GUI_ACTIVE over the 8 XCDs
        k['mfma_busy_frac'] = k['SQ_VALU_MFMA_BUSY_CYCLES'] / 4 / 256 / (k['GRBM_GUI_ACTIVE'] / 8)
    out['kernels'][op] = k
json.dump(out, sys.stdout, indent=1)
