#!/usr/bin/env python3
"""Copies the rocprofv3 summaries of tools/profile.sh <tag> from gpurun_out/prof_<tag> into profiles/ (tracked) and derives the
per-launch counter means of the dominant kernels: profiles/<tag>_*_kernel_stats.csv, <tag>_*_pmc_*.json and the bench lines
of the same commands.  HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (both counters are in KiB; gfx950's
FETCH_SIZE counts half of a coalesced read -- MI355X_MICROARCH.md, HBM / rocprofv3 section)."""
import collections, csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else 'r4'
src = os.path.join(ROOT, 'gpurun_out', 'prof_%s' % tag)
dst = os.path.join(ROOT, 'profiles')
os.makedirs(dst, exist_ok=True)
newest = lambda pat: max(glob.glob(pat, recursive=True), key=os.path.getmtime)
SIZES = {'ur_high_5': 16384, 'from_the_readme': 1024, 'r2d2_maze': 4096}
for name, envs in SIZES.items():
    try:
        ks = newest(os.path.join(src, 'trace_' + name, '**', '*_kernel_stats.csv'))
    except ValueError:
        continue
    shutil.copy(ks, os.path.join(dst, '%s_%s_%d_kernel_stats.csv' % (tag, name, envs)))
    b = os.path.join(src, 'bench_%s.json' % name)
    if os.path.isfile(b):
        lines = [l for l in open(b) if l.startswith('{')]
        if lines:
            json.dump(json.loads(lines[0]), open(os.path.join(dst, '%s_%s_%d_bench_line_under_rocprof.json' % (tag, name, envs)), 'w'), indent=1)


def means(parts, kernel):
    out = {}
    for part in parts:
        files = glob.glob(os.path.join(src, 'pmc_' + part, '**', '*_counter_collection.csv'), recursive=True)
        if not files:
            continue
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(max(files, key=os.path.getmtime))):
            if kernel in r['Kernel_Name']:
                agg[r['Counter_Name']].append(float(r['Counter_Value']))
        for k, v in agg.items():
            out[k] = sum(v) / len(v)
            out[k + '__launches'] = len(v)
    return out


note = 'rocprofv3 --pmc, separate passes, named kernel only, mean over its launches; SQ_* cycle counters are quad-cycles'
ur = means(['ur_fetch', 'ur_write', 'ur_sq', 'ur_sq2'], 'step_kernel')
if ur:
    if 'FETCH_SIZE' in ur and 'WRITE_SIZE' in ur:
        ur['hbm_bytes_per_launch'] = (2.0 * ur['FETCH_SIZE'] + ur['WRITE_SIZE']) * 1024.0
    json.dump({'kernel': 'step_kernel_par', 'workload': 'ur_high_5 x 16384', 'per_launch_means': ur, 'note': note},
              open(os.path.join(dst, '%s_ur_high_5_16384_pmc_step_kernel.json' % tag), 'w'), indent=1)
cam = means(['cam_fetch', 'cam_write', 'cam_sq'], 'render_kernel')
if cam:
    if 'FETCH_SIZE' in cam and 'WRITE_SIZE' in cam:
        cam['hbm_bytes_per_launch'] = (2.0 * cam['FETCH_SIZE'] + cam['WRITE_SIZE']) * 1024.0
    cam['algorithmic_image_bytes_per_launch'] = 1024 * 200 * 200 * 16
    json.dump({'kernel': 'render_kernel', 'workload': 'from_the_readme x 1024, 200x200 rgb + depth', 'per_launch_means': cam, 'note': note},
              open(os.path.join(dst, '%s_from_the_readme_1024_pmc_render_kernel.json' % tag), 'w'), indent=1)
for src_name, dst_name in (('bench_ur_high_5_default.json', 'ur_high_5_16384_bench_line_default.json'), ('bench_from_the_readme_default.json', 'from_the_readme_1024_bench_line_default.json')):
    b = os.path.join(src, src_name)
    if os.path.isfile(b):
        lines = [l for l in open(b) if l.startswith('{')]
        if lines:
            json.dump(json.loads(lines[0]), open(os.path.join(dst, '%s_%s' % (tag, dst_name)), 'w'), indent=1)
print(json.dumps({'ur': {k: v for k, v in ur.items() if not k.endswith('__launches')}, 'cam': {k: v for k, v in cam.items() if not k.endswith('__launches')}}, indent=1))
for name, envs in SIZES.items():
    p = os.path.join(dst, '%s_%s_%d_kernel_stats.csv' % (tag, name, envs))
    if os.path.isfile(p):
        print(name); print(''.join(open(p).readlines()[:4]))
