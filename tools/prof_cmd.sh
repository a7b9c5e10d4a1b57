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
# keep the stalled-pass evidence of an earlier call of the same tag
if [ -d $OUT ]; then for d in $OUT/*.stalled; do [ -d "$d" ] && mkdir -p gpurun_out/stalled_keep && mv "$d" gpurun_out/stalled_keep/${TAG}_$(basename $d)_$(date -u +%H%M%S); done; fi
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 "$@" > $OUT/cmd_stats.out 2> $OUT/cmd_stats.err
echo "stats exit $?"
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
python tools/pmc_summary.py $OUT/stats $OUT/kernel_trace_summary.json > $OUT/kernel_trace_summary.txt 2>&1
rm -rf $OUT/stats
FAILED=0
run() { # name counters...
  name=$1; shift
  # a pass that stalls inside the profiler (seen once in round 3: `HSA ... initialized`, then nothing -- no kernel of the
  # program was ever launched; the pass's own output was overwritten by the retake, so the cause was never established)
  # must not take the whole call with it: every pass runs under its own timeout, well inside gpurun's 7-minute silence rule.
  # A pass that hits the timeout (124 / 137) KEEPS what it wrote (.out / .err / the partial rocprofv3 directory listing) as
  # $OUT/$name.stalled/ -- evidence for the next diagnosis -- produces NO summary (a truncated counter table must not pass
  # for a measurement) and makes the script exit non-zero.
  timeout -k 10 ${PMC_PASS_TIMEOUT:-330} rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -- python3 $CMDLINE > $OUT/$name.out 2> $OUT/$name.err
  rc=$?
  echo "$name exit $rc ($(date -u +%T))"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then
    mkdir -p $OUT/$name.stalled
    mv $OUT/$name.out $OUT/$name.err $OUT/$name.stalled/ 2>/dev/null
    { echo "pass $name: rocprofv3 --pmc $* -- python3 $CMDLINE"; echo "timeout ${PMC_PASS_TIMEOUT:-330} s, exit $rc, $(date -u +%FT%TZ)"; ls -lR $OUT/$name 2>/dev/null | head -40; } > $OUT/$name.stalled/what.txt
    rm -rf $OUT/$name
    FAILED=1
    return $rc
  fi
  python tools/pmc_summary.py $OUT/$name $OUT/$name.summary.json > $OUT/$name.summary.txt 2>&1
  rm -rf $OUT/$name
}
run p1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F32 SQ_WAVES SQ_INSTS_VALU
run p3 FETCH_SIZE GRBM_GUI_ACTIVE
run p4 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
grep -h -A12 "lagged\|gemm_kernel<2\|kmeans_step\|project_kernel\|col_stats_kernel\|normalize_kernel" $OUT/kernel_trace_summary.txt | head -60
exit $FAILED
