#!/bin/bash
# profiles/run_sq.sh <tag> -- SQ counter passes (one rocprofv3 --pmc run each, no tracing) of `python3 bench.py`, for the
# per-kernel picture of what the wavefronts wait for (LDS pipe, VALU, memory).  Run on the GPU box from the repo root.
set -e
TAG=$1; shift
REPO=$(pwd)
export TMPDIR=/tmp
OUT=$REPO/gpurun_out/sq_$TAG
mkdir -p $OUT
cd /tmp
ARGS="--steps 2 --warmup 1 --cpu-sample 0 --no-auto --no-extras --no-verify"
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS SQ_INSTS_LDS" \
           "SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_LDS_ATOMIC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_DATA_FIFO_FULL"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --output-format csv -d $OUT/p$i -- python3 $REPO/bench.py $ARGS "$@" > $OUT/p$i.log 2>&1
done
find $OUT -name "*counter_collection.csv" | head
