#!/bin/bash
# PMC counters of the resblock conv micro-benchmark (tools/time_conv.py): bash tools/pmc_conv.sh <tag>
TAG=${1:-pmc}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
cd $ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $ROOT/gpurun_out/${TAG}_p1 -o run -- python3 tools/time_conv.py > $ROOT/gpurun_out/${TAG}_p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $ROOT/gpurun_out/${TAG}_p2 -o run -- python3 tools/time_conv.py > $ROOT/gpurun_out/${TAG}_p2.log 2>&1
python3 tools/pmc_any.py $(find $ROOT/gpurun_out/${TAG}_p1 $ROOT/gpurun_out/${TAG}_p2 -name "*counter_collection.csv") > $ROOT/gpurun_out/${TAG}_summary.txt
rm -rf $ROOT/gpurun_out/${TAG}_p1 $ROOT/gpurun_out/${TAG}_p2
cat $ROOT/gpurun_out/${TAG}_summary.txt
