#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r2_pytest5.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -4 gpurun_out/r2_pytest5.log
[ $rc -le 1 ] || exit $rc
for w in from_the_readme r2d2_maze ur5_gripper; do
timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline --no-api --age-steps 0 > gpurun_out/r2_bench_$w.json 2> gpurun_out/r2_bench_$w.err || { tail -5 gpurun_out/r2_bench_$w.err; exit 1; }
python3 -c "
import json,sys; d=json.loads([l for l in open('gpurun_out/r2_bench_$w.json') if l.startswith('{')][0]); print('$w', d['value'], d['ms_per_step'], d['roofline']['step_kernel_ms'], d['roofline']['render_kernel_ms'], d['config']['envs_per_wavefront'], d['solver']['contacts_per_env'])"
done
SETTLE=40 ACT_SCALE=0.2 timeout -k 10 300 python tools/gpu_stamps.py readme 1024 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2_stamps_readme.log
