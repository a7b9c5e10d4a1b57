#!/bin/bash
# Copies the summaries tools/prof_round.sh <tag> left under gpurun_out/ (scratch) into profiles/ (tracked), one naming scheme
# per round: profiles/<tag>_<part>_<what>.  Raw rocprofv3 output is never copied (prof_cmd.sh / prof_stats.sh delete it).
TAG="${1:-r04}"
cd "$(dirname "$0")/.."
cp_if() { [ -f "$1" ] && cp "$1" "$2" && echo "  $2"; }
for part in bench8k shuffled8k benchL c2 c3; do
  d=gpurun_out/prof_${TAG}_$part
  cp_if $d/kernel_stats.csv profiles/${TAG}_${part}_kernel_stats.csv
  cp_if $d/kernel_trace_summary.txt profiles/${TAG}_${part}_kernel_trace_summary.txt
  for p in p1 p3 p4; do cp_if $d/$p.summary.txt profiles/${TAG}_${part}_pmc_$p.summary.txt; done
  for st in $d/*.stalled; do [ -d "$st" ] && mkdir -p profiles/${TAG}_stalled && cp -r "$st" profiles/${TAG}_stalled/${part}_$(basename $st) && echo "  stalled: $st"; done
done
for part in c5 refsmall; do
  d=gpurun_out/prof_${TAG}_$part
  cp_if $d/kernel_stats.csv profiles/${TAG}_${part}_kernel_stats.csv
  cp_if $d/kernel_trace_summary.txt profiles/${TAG}_${part}_kernel_trace_summary.txt
done
cp_if gpurun_out/prof_${TAG}_timeline/step_timeline.txt profiles/${TAG}_bench8k_step_timeline.txt
cp_if gpurun_out/prof_${TAG}_shuffled/step_timeline.txt profiles/${TAG}_shuffled_step_timeline.txt
cp_if gpurun_out/prof_${TAG}_shuffled/kernel_stats.csv profiles/${TAG}_shuffled_kernel_stats.csv
cp_if gpurun_out/b_${TAG}_driver.json profiles/${TAG}_bench_line_driver_form.json
cp_if gpurun_out/b_${TAG}_default.json profiles/${TAG}_bench_line.json
cp_if gpurun_out/b_c2_${TAG}.json profiles/${TAG}_bench_c2_line.json
cp_if gpurun_out/c5_${TAG}.json profiles/${TAG}_c5_numbers.json
cp_if gpurun_out/c3_${TAG}.json profiles/${TAG}_c3_numbers.json
cp_if gpurun_out/t_${TAG}_final.log profiles/${TAG}_gpu_tests.log
