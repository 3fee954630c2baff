#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 300 python tools/gpu_stamps.py child 16384 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2_stamps_child.log
