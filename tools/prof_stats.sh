#!/bin/bash
# rocprofv3 --kernel-trace --stats of a python command (no counter passes): gpurun_out/prof_<tag>/{kernel_stats.csv,kernel_trace_summary.txt}
# usage: tools/prof_stats.sh <tag> <script.py> [args...]
TAG="$1"; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 "$@" > $OUT/cmd_stats.out 2> $OUT/cmd_stats.err
echo "stats exit $?"
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
python tools/pmc_summary.py $OUT/stats $OUT/kernel_trace_summary.json > $OUT/kernel_trace_summary.txt 2>&1
rm -rf $OUT/stats
head -12 $OUT/kernel_trace_summary.txt
