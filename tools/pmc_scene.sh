#!/bin/bash
# SQ counters of the step kernel on a settled scene: tools/pmc_scene.sh <config> <envs> <settle> <steps> <scale>
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/pmc_$1
mkdir -p $OUT
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/a -- python3 tools/gpu_scene_loop.py "$@" > $OUT/a.log 2>&1
echo "a rc=$?"
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/b -- python3 tools/gpu_scene_loop.py "$@" > $OUT/b.log 2>&1
echo "b rc=$?"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for part in ('a', 'b'):
    rows = []
    for f in glob.glob(out + '/' + part + '/**/*counter_collection.csv', recursive=True):
        rows += list(csv.DictReader(open(f)))
    step = [r for r in rows if 'step_kernel' in r['Kernel_Name']]
    # the last dispatches are the settled ones: average the final 25 % per counter
    by = collections.defaultdict(list)
    for r in step: by[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in sorted(by.items()):
        tail = v[-max(1, len(v) // 4):]
        print('%-24s %14.0f  (n=%d)' % (k, sum(tail) / len(tail), len(v)))
PY
