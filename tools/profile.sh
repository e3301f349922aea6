#!/bin/bash
# rocprofv3 passes for one round (run on the GPU box through gpurun):
#   1. --kernel-trace --stats  : per-kernel time (must agree with bench.py's HIP-event averages)
#   2. --pmc FETCH_SIZE        : HBM read traffic   (own pass, counters only + kernel-trace)
#   3. --pmc WRITE_SIZE        : HBM write traffic  (own pass)
# Outputs land in gpurun_out/prof_<tag>/ ; tools/summarize_profile.py condenses them into profiles/.
set -u
TAG=${1:-r01}
ARGS=${2:-"--steps 2 --warmup 1 --no-cpu-baseline --timed-only"}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $REPO/bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o fetch -- python $REPO/bench.py $ARGS > $OUT/bench_fetch.json 2> $OUT/fetch.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o write -- python $REPO/bench.py $ARGS > $OUT/bench_write.json 2> $OUT/write.log
cd $REPO
python tools/summarize_profile.py $OUT $TAG
ls -la $OUT $OUT/* | head -40
# gpurun merges only gpurun_out/ back: leave copies of the condensed files there (copy them into profiles/ and commit)
mkdir -p $REPO/gpurun_out/profiles && cp $REPO/profiles/${TAG}_* $REPO/profiles/pmc_traffic.json $REPO/gpurun_out/profiles/ 2>/dev/null
