#!/bin/bash
# Kernel trace of bench.py's step on the GPU box: bash tools/profile_step.sh <tag> [multi]
# Writes gpurun_out/<tag>_kernel_stats.csv and gpurun_out/<tag>_by_grid.txt (copy the ones to keep into profiles/).
# Default: one stream (MUNIT_NO_SIDE_STREAM / MUNIT_NO_BRANCH_STREAMS) so that per-kernel durations are those of the
# kernel alone; "multi" keeps the three-stream schedule of the timed region.
set -e
TAG=${1:-prof}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/${TAG}_trace
if [ "$2" != "multi" ]; then export MUNIT_NO_SIDE_STREAM=1 MUNIT_NO_BRANCH_STREAMS=1; fi
export TMPDIR=/tmp
cd $ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o run -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-modes $BENCH_ARGS > $ROOT/gpurun_out/${TAG}_bench.log 2>&1
TRACE=$(find $OUT -name "*kernel_trace.csv" | head -1)
STATS=$(find $OUT -name "*kernel_stats.csv" | head -1)
cp $STATS $ROOT/gpurun_out/${TAG}_kernel_stats.csv
# bench.py runs 1 warm-up + 2 timed steps = 3 steps in the trace
python3 tools/trace_summary.py $TRACE 3 60 > $ROOT/gpurun_out/${TAG}_by_grid.txt
rm -rf $OUT
head -45 $ROOT/gpurun_out/${TAG}_by_grid.txt
