#!/bin/bash
# final evidence pass of the round: rocprofv3 traces + counter passes, in-kernel stamps, aged-rollout drift, render timing
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
bash tools/profile_r2.sh > gpurun_out/profile_r2.log 2>&1; echo "profile rc=$?"; tail -3 gpurun_out/profile_r2.log
python tools/gpu_stamps.py ur_ik 16384 2>&1 | grep -v amdgpu.ids > gpurun_out/stamps_ur_ik.txt; echo "stamps ur rc=$?"
python tools/gpu_stamps.py maze 4096 2>&1 | grep -v amdgpu.ids > gpurun_out/stamps_maze.txt
python tools/gpu_stamps.py readme 1024 2>&1 | grep -v amdgpu.ids > gpurun_out/stamps_readme.txt
python tools/gpu_stamps.py child 16384 2>&1 | grep -v amdgpu.ids > gpurun_out/stamps_child.txt; echo "stamps done"
timeout -k 10 300 python tools/gpu_drift.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r2_drift.log; echo "drift rc=$?"
DIAGS=0,2,4 python tools/gpu_cam_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/cam_bench.txt; echo "cam rc=$?"
for w in ur5_child_gripper drone_pilot marbles; do
timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline --no-api --age-steps 0 > gpurun_out/r2_bench_$w.json 2> gpurun_out/r2_bench_$w.err; echo "$w rc=$?"
done
