#!/bin/bash
# HBM traffic + MFMA utilisation of bench.py's step: three separate rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE /
# SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE), single stream, summarised by tools/pmc_summary.py into
# gpurun_out/<tag>_pmc_hbm_mfma.txt (copy into profiles/).   bash tools/pmc_step.sh <tag>
TAG=${1:-pmc}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp MUNIT_NO_SIDE_STREAM=1 MUNIT_NO_BRANCH_STREAMS=1
cd $ROOT
for pass in fetch write mfma; do
  case $pass in
    fetch) C="FETCH_SIZE";;
    write) C="WRITE_SIZE";;
    mfma)  C="SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE";;
  esac
  rocprofv3 --pmc $C --output-format csv -d $ROOT/gpurun_out/${TAG}_$pass -o run -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-modes $BENCH_ARGS > $ROOT/gpurun_out/${TAG}_$pass.log 2>&1
done
F=$(find $ROOT/gpurun_out/${TAG}_fetch -name "*counter_collection.csv" | head -1)
W=$(find $ROOT/gpurun_out/${TAG}_write -name "*counter_collection.csv" | head -1)
M=$(find $ROOT/gpurun_out/${TAG}_mfma -name "*counter_collection.csv" | head -1)
python3 tools/pmc_summary.py $F $W $M $ROOT/gpurun_out/${TAG}_pmc_hbm_mfma.txt > /dev/null
rm -rf $ROOT/gpurun_out/${TAG}_fetch $ROOT/gpurun_out/${TAG}_write $ROOT/gpurun_out/${TAG}_mfma
head -30 $ROOT/gpurun_out/${TAG}_pmc_hbm_mfma.txt
