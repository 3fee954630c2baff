#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 60 tools/micro/valu_issue3 2>&1 | tee gpurun_out/r2_micro_valu3.log
