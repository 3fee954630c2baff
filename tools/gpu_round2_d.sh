#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 120 python tools/gpu_ft_debug.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2_ft_debug.log
