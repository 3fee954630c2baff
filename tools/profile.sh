#!/bin/bash
# Profiling recipe (run on the GPU box through gpurun): bash tools/profile.sh <tag>   (tag: r4, ...).  (`--legs none`: the headline loop alone, so that a
# kernel's row in kernel_stats.csv is that loop's launches and nothing else.)  Kernel trace + stats per workload, then the HBM / SQ
# counters in their own passes (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950; never --pmc together with a trace).
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
TAG=${1:-r4}
OUT=gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
trace() {  # name, bench args...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$name -- python3 bench.py "$@" --no-pmc --no-cpu-baseline --no-api --legs none > $OUT/bench_$name.json 2> $OUT/trace_$name.err
  echo "trace $name rc=$?"
}
trace ur_high_5 --steps 104 --warmup 20 --age-steps 0
trace from_the_readme --workload from_the_readme --steps 40 --warmup 30 --age-steps 0
trace r2d2_maze --workload r2d2_maze --steps 40 --warmup 20 --age-steps 0
pmc() {  # name, counters..., then -- bench args
  local name=$1; shift; local counters=(); while [ "$1" != "--" ]; do counters+=("$1"); shift; done; shift
  rocprofv3 --pmc "${counters[@]}" --output-format csv -d $OUT/pmc_$name -- python3 bench.py "$@" --no-pmc --no-cpu-baseline --no-api --age-steps 0 --legs none --inner > /dev/null 2> $OUT/pmc_$name.err
  echo "pmc $name rc=$?"
}
pmc ur_fetch FETCH_SIZE -- --steps 40 --warmup 10
pmc ur_write WRITE_SIZE -- --steps 40 --warmup 10
pmc ur_sq SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES -- --steps 40 --warmup 10
pmc ur_sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA -- --steps 40 --warmup 10
pmc cam_write WRITE_SIZE -- --workload from_the_readme --steps 16 --warmup 30
pmc cam_fetch FETCH_SIZE -- --workload from_the_readme --steps 16 --warmup 30
pmc cam_sq SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY -- --workload from_the_readme --steps 16 --warmup 30
# the bench line with its own counter passes (traffic / VALU share measured in the same invocation)
python3 bench.py > $OUT/bench_ur_high_5_default.json 2> $OUT/bench_default.err; echo "default bench rc=$?"
python3 bench.py --workload from_the_readme --steps 40 --warmup 30 --age-steps 0 --no-api > $OUT/bench_from_the_readme_default.json 2> $OUT/bench_readme_default.err; echo "readme bench rc=$?"
find $OUT -name "*kernel_stats.csv" -o -name "*counter_collection.csv" | head -40
