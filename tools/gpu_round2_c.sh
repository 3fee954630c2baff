#!/bin/bash
# stamps (fresh and aged) of the helper-wave kernel with the per-wavefront timeline
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 300 python tools/gpu_stamps.py ur_ik 16384 > gpurun_out/r2_stamps_ur.log 2>&1 || { tail -5 gpurun_out/r2_stamps_ur.log; exit 1; }
grep -v amdgpu.ids gpurun_out/r2_stamps_ur.log
SETTLE=4400 timeout -k 10 300 python tools/gpu_stamps.py ur_ik 16384 > gpurun_out/r2_stamps_ur_aged.log 2>&1 || { tail -5 gpurun_out/r2_stamps_ur_aged.log; exit 1; }
grep -v amdgpu.ids gpurun_out/r2_stamps_ur_aged.log
