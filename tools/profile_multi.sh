#!/bin/bash
# Kernel trace of bench.py's step WITH the three-stream schedule of the timed region + timeline statistics.
# bash tools/profile_multi.sh <tag>  -> gpurun_out/<tag>_timeline.txt, gpurun_out/<tag>_by_grid_multi.txt
TAG=${1:-multi}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/${TAG}_trace
export TMPDIR=/tmp
cd $ROOT
rocprofv3 --kernel-trace --output-format csv -d $OUT -o run -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-modes $BENCH_ARGS > $ROOT/gpurun_out/${TAG}_bench.log 2>&1
TRACE=$(find $OUT -name "*kernel_trace.csv" | head -1)
python3 tools/trace_timeline.py $TRACE 3 > $ROOT/gpurun_out/${TAG}_timeline.txt 2>&1
python3 tools/trace_summary.py $TRACE 3 40 > $ROOT/gpurun_out/${TAG}_by_grid_multi.txt
rm -rf $OUT
cat $ROOT/gpurun_out/${TAG}_timeline.txt
