#!/bin/bash
# Steady-state timeline of the contract-batch step (global batch 8192 pairs):
# rocprofv3 --kernel-trace --stats of a bench run whose headline batch IS 8192, summarised per kernel.
TAG="${1:-small}"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
ARGS="${BENCH_ARGS:---batch 8192 --steps 400 --warmup 50 --no-cpu-baseline --large-batch 0 --other-mode-steps 0 --frames 2000000}"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python bench.py $ARGS > $OUT/bench.json 2> $OUT/bench.err
echo "stats exit $?"
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
python tools/trace_gaps.py $OUT/stats 200 > $OUT/step_timeline.txt 2>&1
rm -rf $OUT/stats
cat $OUT/step_timeline.txt
