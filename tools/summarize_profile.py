#!/usr/bin/env python3
"""Copies the rocprofv3 summaries of tools/profile_r1.sh from gpurun_out/ into profiles/ and derives
profiles/r1_pmc_traffic.json (HBM bytes per step_kernel launch, gfx950 FETCH_SIZE x2 correction)."""
import collections, csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, 'gpurun_out', 'prof_r1')
dst = os.path.join(ROOT, 'profiles')
tag = sys.argv[1] if len(sys.argv) > 1 else 'r1'
os.makedirs(dst, exist_ok=True)
newest = lambda pat: max(glob.glob(pat), key=os.path.getmtime)
ks = newest(os.path.join(src, 'trace', '*', '*_kernel_stats.csv'))
shutil.copy(ks, os.path.join(dst, '%s_ur_high_5_16384_kernel_stats.csv' % tag))
means = {}
for name in ('pmc_fetch', 'pmc_write', 'pmc_sq', 'pmc_sq2'):
    files = glob.glob(os.path.join(src, name, '*', '*_counter_collection.csv'))
    if not files:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(max(files, key=os.path.getmtime))):
        if 'step_kernel' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in agg.items():
        means[k] = sum(v) / len(v)
with open(os.path.join(dst, '%s_ur_high_5_16384_pmc_step_kernel.json' % tag), 'w') as fh:
    json.dump({'per_launch_means': means, 'note': 'rocprofv3 --pmc, separate passes, step_kernel dispatches only; SQ_* cycle counters are quad-cycles'}, fh, indent=1)
if 'FETCH_SIZE' in means and 'WRITE_SIZE' in means:
    # FETCH_SIZE / WRITE_SIZE are in KiB; gfx950 FETCH_SIZE under-reports coalesced reads by 2x (MI355X_MICROARCH.md, HBM);
    # our reads are 4 B/lane coalesced rows, a width the guide marks uncalibrated -- the x2 is applied as prescribed.
    traffic = (2.0 * means['FETCH_SIZE'] + means['WRITE_SIZE']) * 1024.0
    json.dump({'ur_high_5': traffic, 'fetch_size_kib': means['FETCH_SIZE'], 'write_size_kib': means['WRITE_SIZE'],
               'formula': '(2*FETCH_SIZE + WRITE_SIZE) * 1024 bytes per step_kernel launch'},
              open(os.path.join(dst, 'r1_pmc_traffic.json'), 'w'), indent=1)
print(json.dumps(means, indent=1))
