#!/bin/bash
# profiles/run_profile.sh <tag> [bench args...] -- run on the GPU box (via gpurun) from the repo root.
# 1. rocprofv3 --kernel-trace --stats of `python3 bench.py`  -> gpurun_out/prof_<tag>/
# 2. two separate PMC passes (FETCH_SIZE, WRITE_SIZE; they do not fit one pass on gfx950)
# Copy the summaries you want judged into profiles/ afterwards (profiles/summarize.py does it).
set -e
TAG=$1; shift
REPO=$(pwd)
export TMPDIR=/tmp
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py --cpu-sample 0 --no-auto --no-extras "$@" > $OUT/bench_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/bench.py --cpu-sample 0 --no-auto --no-extras --no-verify "$@" > $OUT/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $REPO/bench.py --cpu-sample 0 --no-auto --no-extras --no-verify "$@" > $OUT/bench_write.log 2>&1
find $OUT -name "*.csv" | head -20
