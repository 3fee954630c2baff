#!/bin/bash
# Round-1 profiling recipe (run on the GPU box through gpurun).  Kernel trace + stats first, then the
# HBM counters in their own passes (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950).
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/prof_r1
mkdir -p $OUT
ARGS="bench.py --steps 60 --warmup 10 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err
echo "trace rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/bench_fetch.json 2> $OUT/fetch.err
echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ARGS > $OUT/bench_write.json 2> $OUT/write.err
echo "write rc=$?"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq -- python3 $ARGS > $OUT/bench_sq.json 2> $OUT/sq.err
echo "sq rc=$?"
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_sq2 -- python3 $ARGS > $OUT/bench_sq2.json 2> $OUT/sq2.err
echo "sq2 rc=$?"
find $OUT -name "*.csv" | head -40
