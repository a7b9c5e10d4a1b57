#!/bin/bash
# rocprofv3 evidence for an arbitrary python command, written under gpurun_out/prof_<tag>/ as small summaries:
#   kernel_stats.csv / kernel_trace_summary.txt   --kernel-trace --stats of the command
#   p1 (SQ: waves, waits, MFMA busy) p3 (FETCH_SIZE + GRBM_GUI_ACTIVE) p4 (WRITE_SIZE + L2 hit/miss): one --pmc pass each,
#   no trace domain besides --kernel-trace, aggregated per (kernel, grid) by tools/pmc_summary.py (raw CSVs deleted)
# usage: tools/prof_cmd.sh <tag> <script.py> [args...]      (the program itself follows `--`: python3 <script>)
TAG="$1"; shift
CMDLINE="${PMC_CMDLINE:-$*}"   # the counter passes serialise every dispatch: give them a shorter run (PMC_CMDLINE) where the command is long
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 "$@" > $OUT/cmd_stats.out 2> $OUT/cmd_stats.err
echo "stats exit $?"
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
python tools/pmc_summary.py $OUT/stats $OUT/kernel_trace_summary.json > $OUT/kernel_trace_summary.txt 2>&1
rm -rf $OUT/stats
run() { # name counters...
  name=$1; shift
  # a pass that stalls inside the profiler (seen once: no kernel ever launched) must not take the whole call with it
  timeout -k 10 ${PMC_PASS_TIMEOUT:-330} rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -- python3 $CMDLINE > $OUT/$name.out 2> $OUT/$name.err
  echo "$name exit $?"
  python tools/pmc_summary.py $OUT/$name $OUT/$name.summary.json > $OUT/$name.summary.txt 2>&1
  rm -rf $OUT/$name
}
run p1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F32 SQ_WAVES SQ_INSTS_VALU
run p3 FETCH_SIZE GRBM_GUI_ACTIVE
run p4 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
grep -h -A12 "lagged\|gemm_kernel<2\|kmeans_step\|project_kernel\|col_stats_kernel\|normalize_kernel" $OUT/kernel_trace_summary.txt | head -60
