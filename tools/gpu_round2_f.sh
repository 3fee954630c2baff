#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_camera.py tests/test_parity_gpu.py -m gpu -q -k "camera or render or readme" > gpurun_out/r2_pytest_cam.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -5 gpurun_out/r2_pytest_cam.log
[ $rc -le 1 ] || exit $rc
DIAGS=0,2,4 timeout -k 10 300 python3 tools/gpu_cam_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2_cam_bench.log
