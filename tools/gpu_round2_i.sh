#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r2_pytest4.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -4 gpurun_out/r2_pytest4.log
[ $rc -le 1 ] || exit $rc
timeout -k 10 300 python bench.py --workload r2d2_maze --no-cpu-baseline --no-api --age-steps 0 > gpurun_out/r2_bench_maze.json 2> gpurun_out/r2_bench_maze.err || { tail -5 gpurun_out/r2_bench_maze.err; exit 1; }
python3 -c "
import json; d=json.loads([l for l in open('gpurun_out/r2_bench_maze.json') if l.startswith('{')][0]); print('maze', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['solver']['pgs_iterations_last_substep'], d['solver']['contacts_per_env'], d['config']['envs_per_wavefront'])"
ACT_SCALE=10 SETTLE=30 timeout -k 10 300 python tools/gpu_stamps.py maze 4096 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2_stamps_maze.log
for w in marbles drone_pilot; do
timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline --no-api --age-steps 0 > gpurun_out/r2_bench_$w.json 2> gpurun_out/r2_bench_$w.err || { tail -5 gpurun_out/r2_bench_$w.err; exit 1; }
python3 -c "
import json,sys; d=json.loads([l for l in open('gpurun_out/r2_bench_$w.json') if l.startswith('{')][0]); print('$w', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['envs_per_wavefront'])"
done
