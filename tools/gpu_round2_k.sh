#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r2_pytest6.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -4 gpurun_out/r2_pytest6.log
[ $rc -le 1 ] || exit $rc
for w in ur5_child_gripper r2d2_maze from_the_readme drone_pilot marbles; do
timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline --no-api --age-steps 0 > gpurun_out/r2_bench_$w.json 2> gpurun_out/r2_bench_$w.err || { tail -5 gpurun_out/r2_bench_$w.err; exit 1; }
python3 -c "
import json,sys; d=json.loads([l for l in open('gpurun_out/r2_bench_$w.json') if l.startswith('{')][0]); print('$w', d['value'], d['ms_per_step'], d['roofline']['step_kernel_ms'], d['roofline']['render_kernel_ms'], d['config']['envs_per_wavefront'])"
done
