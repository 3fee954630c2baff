#!/bin/bash
# full GPU suite + headline bench + aged-rollout drift
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r2_pytest3.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -4 gpurun_out/r2_pytest3.log
[ $rc -le 1 ] || exit $rc
timeout -k 10 400 python bench.py --no-cpu-baseline > gpurun_out/r2_bench3.json 2> gpurun_out/r2_bench3.err || { tail -5 gpurun_out/r2_bench3.err; exit 1; }
echo bench done
timeout -k 10 300 python tools/gpu_drift.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2_drift.log
