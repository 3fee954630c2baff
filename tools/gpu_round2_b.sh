#!/bin/bash
# GPU pass: tests + headline bench + in-kernel stamps of the helper-wave kernel
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -k "randomizer or ur_high_5_ik" > gpurun_out/r2_pytest2.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -3 gpurun_out/r2_pytest2.log
[ $rc -le 1 ] || exit $rc
timeout -k 10 60 tools/micro/valu_issue2 > gpurun_out/r2_micro_valu2.log 2>&1; cat gpurun_out/r2_micro_valu2.log
timeout -k 10 200 python tools/gpu_ik_cost.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2_ik_cost.log
timeout -k 10 400 python bench.py --no-cpu-baseline > gpurun_out/r2_bench2.json 2> gpurun_out/r2_bench2.err || { tail -5 gpurun_out/r2_bench2.err; exit 1; }
echo bench done
timeout -k 10 300 python tools/gpu_stamps.py ur_ik 16384 > gpurun_out/r2_stamps_ur.log 2>&1 || { tail -5 gpurun_out/r2_stamps_ur.log; exit 1; }
tail -16 gpurun_out/r2_stamps_ur.log
