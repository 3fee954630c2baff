#!/bin/bash
# step time of a contact workload per envs-per-wavefront mode (DG_MAX_LANES)
cd "$GRAFT_REPO_ROOT" || exit 1
w=${1:-r2d2_maze}
for l in ${LANES_LIST:-16 8 4 1}; do
DG_MAX_LANES=$l timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline --no-api --age-steps 0 > gpurun_out/lanes_$w_$l.json 2> gpurun_out/lanes_$w_$l.err || { tail -5 gpurun_out/lanes_$w_$l.err; exit 1; }
python3 -c "
import json,sys; d=json.loads([l for l in open('gpurun_out/lanes_$w_$l.json') if l.startswith('{')][0]); print('$w lanes=$l', d['ms_per_step'], d['roofline']['step_kernel_ms'], d['config']['envs_per_wavefront'])"
done
