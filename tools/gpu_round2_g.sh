#!/bin/bash
# SQ counters of the render kernel
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out; rm -rf gpurun_out/pmc_cam
DIAGS=0 timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d gpurun_out/pmc_cam/a -- python3 tools/gpu_cam_bench.py > gpurun_out/pmc_cam_a.log 2>&1
echo "a rc=$?"
DIAGS=0 timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/pmc_cam/b -- python3 tools/gpu_cam_bench.py > gpurun_out/pmc_cam_b.log 2>&1
echo "b rc=$?"
python3 - <<'PY'
import csv, glob, collections
for part in ('a', 'b'):
    rows = []
    for f in glob.glob('gpurun_out/pmc_cam/' + part + '/**/*counter_collection.csv', recursive=True):
        rows += list(csv.DictReader(open(f)))
    r = [x for x in rows if 'render_kernel' in x['Kernel_Name']]
    by = collections.defaultdict(list)
    for x in r: by[x['Counter_Name']].append(float(x['Counter_Value']))
    for k, v in sorted(by.items()):
        print('%-24s mean %14.0f  first-half %14.0f  second-half %14.0f (n=%d)' % (k, sum(v) / len(v), sum(v[:len(v)//2]) / max(1, len(v)//2), sum(v[len(v)//2:]) / max(1, len(v) - len(v)//2), len(v)))
PY
