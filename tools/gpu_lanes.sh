#!/bin/bash
# lane-mode sweep: envs per wavefront vs batch size on the headline workload (prints kernel_ms per setting)
for envs in 16384 65536; do
  for cfg in "64 0" "64 1" "32 1" "16 1"; do
    set -- $cfg
    if [ "$2" = "1" ]; then export DG_NO_HELPER_WAVE=1; else unset DG_NO_HELPER_WAVE; fi
    DG_MAX_LANES=$1 python bench.py --no-cpu-baseline --steps 100 --warmup 20 --envs-per-gpu $envs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('envs',$envs,'lanes',$1,'nohelper',$2,'ms_per_step',d['ms_per_step'],'value',d['value'])"
  done
done
