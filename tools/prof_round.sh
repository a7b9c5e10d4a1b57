#!/bin/bash
# The profile set of a round from ONE build: test suite, smoke, step timelines (contract batch, shuffled, small networks),
# kernel-trace + PMC passes of the contract batch, the large batch, C2 (autoencoder), C3 (TICA / hTICA), kernel-trace of C5
# (k-means pass / centroid search) and of the small-network Deep-TICA fits.  Everything lands under gpurun_out/.  A line is
# printed between passes (gpurun's silence rule); every counter pass runs under its own timeout (tools/prof_cmd.sh) and a
# pass that stalls is kept as <tag>/<pass>.stalled/ and makes this script exit non-zero.
#   tools/prof_round.sh <tag> [part ...]      parts: tests timeline bench8k benchL c2 c3 c5 small lines   (default: all)
TAG="${1:-r04}"; shift
PARTS="${*:-tests timeline bench8k benchL c2 c3 c5 small lines}"
cd $GRAFT_REPO_ROOT
RC=0
has() { case " $PARTS " in *" $1 "*) return 0;; esac; return 1; }
COMMON="--no-cpu-baseline --other-mode-steps 0 --shuffled-steps 0 --c2-steps 0 --ref-small-steps 0 --fit-epochs 0"
if has tests; then
  echo "--- tests $(date -u +%T)"
  python -m pytest tests -m gpu -q > gpurun_out/t_${TAG}_final.log 2>&1; tail -2 gpurun_out/t_${TAG}_final.log; grep -E "^FAILED|^ERROR" gpurun_out/t_${TAG}_final.log | head
  python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
fi
if has timeline; then
  echo "--- step timelines $(date -u +%T)"
  BENCH_ARGS="--batch 8192 --steps 400 --warmup 50 --large-batch 0 --frames 2000000 $COMMON" bash tools/prof_small.sh ${TAG}_timeline | tail -9
  BENCH_ARGS="--steps 20 --warmup 5 --large-batch 0 --frames 4000000 --no-cpu-baseline --other-mode-steps 0 --shuffled-steps 300 --c2-steps 0 --ref-small-steps 0 --fit-epochs 0" bash tools/prof_small.sh ${TAG}_shuffled | tail -9
fi
if has bench8k; then
  echo "--- bench8k passes $(date -u +%T)"
  PMC_CMDLINE="bench.py --steps 100 --warmup 20 --large-batch 0 --frames 2000000 $COMMON" \
    bash tools/prof_cmd.sh ${TAG}_bench8k bench.py --steps 800 --warmup 50 --large-batch 0 --frames 5000000 $COMMON | grep -E "exit"; [ ${PIPESTATUS[0]} -ne 0 ] && RC=1
  echo "--- shuffled passes $(date -u +%T)"
  PMC_CMDLINE="bench.py --steps 5 --warmup 5 --large-batch 0 --frames 2000000 --no-cpu-baseline --other-mode-steps 0 --shuffled-steps 100 --c2-steps 0 --ref-small-steps 0 --fit-epochs 0" \
    bash tools/prof_cmd.sh ${TAG}_shuffled8k bench.py --steps 5 --warmup 5 --large-batch 0 --frames 4000000 --no-cpu-baseline --other-mode-steps 0 --shuffled-steps 600 --c2-steps 0 --ref-small-steps 0 --fit-epochs 0 | grep -E "exit"; [ ${PIPESTATUS[0]} -ne 0 ] && RC=1
fi
if has benchL; then
  echo "--- benchL passes $(date -u +%T)"
  PMC_CMDLINE="bench.py --batch 524208 --steps 12 --warmup 3 --large-batch 0 --frames 2000000 --profile-every 1 $COMMON" \
    bash tools/prof_cmd.sh ${TAG}_benchL bench.py --batch 524208 --steps 61 --warmup 5 --large-batch 0 --profile-every 1 $COMMON | grep -E "exit"; [ ${PIPESTATUS[0]} -ne 0 ] && RC=1
fi
if has c2; then
  echo "--- c2 $(date -u +%T)"
  PMC_CMDLINE="bench.py --config c2 --steps 100 --warmup 20 --no-cpu-baseline" bash tools/prof_cmd.sh ${TAG}_c2 bench.py --config c2 --steps 400 --warmup 50 --no-cpu-baseline | grep -E "exit"; [ ${PIPESTATUS[0]} -ne 0 ] && RC=1
fi
if has c3; then
  echo "--- c3 $(date -u +%T)"
  bash tools/prof_cmd.sh ${TAG}_c3 tools/bench_configs.py c3 | grep -E "exit"; [ ${PIPESTATUS[0]} -ne 0 ] && RC=1
fi
if has c5; then
  echo "--- c5 $(date -u +%T)"
  bash tools/prof_stats.sh ${TAG}_c5 tools/bench_configs.py c5 | head -3
  python tools/bench_configs.py c5 2>/dev/null | tail -1 > gpurun_out/c5_${TAG}.json; cut -c1-260 gpurun_out/c5_${TAG}.json
  python tools/bench_configs.py c3 2>/dev/null | tail -1 > gpurun_out/c3_${TAG}.json; cut -c1-400 gpurun_out/c3_${TAG}.json
fi
if has small; then
  echo "--- small networks $(date -u +%T)"
  bash tools/prof_stats.sh ${TAG}_refsmall bench.py --config ref_small --steps 300 --no-cpu-baseline | head -3
fi
if has lines; then
  echo "--- bench lines $(date -u +%T)"
  python bench.py --steps 20 --warmup 5 > gpurun_out/b_${TAG}_driver.json 2> gpurun_out/b_${TAG}_driver.err; cut -c1-200 gpurun_out/b_${TAG}_driver.json
  python bench.py > gpurun_out/b_${TAG}_default.json 2> gpurun_out/b_${TAG}_default.err; cut -c1-200 gpurun_out/b_${TAG}_default.json
  python bench.py --config c2 > gpurun_out/b_c2_${TAG}.json 2> gpurun_out/b_c2_${TAG}.err; cut -c1-200 gpurun_out/b_c2_${TAG}.json
fi
echo "--- done rc=$RC $(date -u +%T)"
exit $RC
