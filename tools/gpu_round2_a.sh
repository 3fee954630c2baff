#!/bin/bash
# first GPU pass of round 2: tests, tolerance data, bench line (+ --pmc), aged-rollout drift
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r2_pytest1.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -3 gpurun_out/r2_pytest1.log
[ $rc -le 1 ] || exit $rc
timeout -k 10 300 python tools/gpu_tolerances.py > gpurun_out/r2_tol.log 2>&1 || exit 1
echo tol done
timeout -k 10 400 python bench.py > gpurun_out/r2_bench1.json 2> gpurun_out/r2_bench1.err || { tail -5 gpurun_out/r2_bench1.err; exit 1; }
echo bench done
timeout -k 10 500 python bench.py --pmc --no-cpu-baseline --no-api --age-steps 0 > gpurun_out/r2_bench_pmc.json 2> gpurun_out/r2_bench_pmc.err || { tail -5 gpurun_out/r2_bench_pmc.err; exit 1; }
echo pmc done
timeout -k 10 300 python bench.py --workload from_the_readme --no-cpu-baseline > gpurun_out/r2_bench_readme.json 2> gpurun_out/r2_bench_readme.err || { tail -5 gpurun_out/r2_bench_readme.err; exit 1; }
echo readme done
timeout -k 10 300 python bench.py --workload ur5_gripper --no-cpu-baseline --age-steps 0 > gpurun_out/r2_bench_gripper.json 2> gpurun_out/r2_bench_gripper.err || { tail -5 gpurun_out/r2_bench_gripper.err; exit 1; }
echo gripper done
