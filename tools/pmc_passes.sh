#!/bin/bash
# rocprofv3 evidence for a bench run, written under gpurun_out/prof_<tag>/ as small summaries:
#   kernel_stats.csv   --kernel-trace --stats summary of the timed command
#   p1..p4.*           one --pmc pass per counter set (no trace domains besides --kernel-trace),
#                      aggregated per kernel by tools/pmc_summary.py (raw CSVs are deleted)
TAG="${1:-r01}"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
STATS_ARGS="${STATS_ARGS:---batch 524208 --steps 61 --warmup 5 --no-cpu-baseline --large-batch 0 --other-mode-steps 0}"
ARGS="${BENCH_ARGS:---batch 524208 --steps 61 --warmup 5 --no-cpu-baseline --large-batch 0 --other-mode-steps 0 --frames 5000000}"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python bench.py $STATS_ARGS > $OUT/bench_stats.json 2> $OUT/bench_stats.err
echo "stats exit $?"
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
python tools/pmc_summary.py $OUT/stats $OUT/kernel_trace_summary.json > $OUT/kernel_trace_summary.txt 2>&1
python tools/trace_gaps.py $OUT/stats 40 > $OUT/step_timeline.txt 2>&1
rm -rf $OUT/stats
run() { # name counters...
  name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -- python bench.py $ARGS > $OUT/$name.json 2> $OUT/$name.err
  echo "$name exit $?"
  python tools/pmc_summary.py $OUT/$name $OUT/$name.summary.json > $OUT/$name.summary.txt 2>&1
  rm -rf $OUT/$name
}
run p1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F32 SQ_WAVES
run p2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU
run p3 FETCH_SIZE GRBM_GUI_ACTIVE
run p4 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
