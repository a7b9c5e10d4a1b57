#!/bin/bash
# The profile set of a round from ONE build: test suite, smoke, step timeline, kernel-trace + PMC passes of the contract
# batch, the large batch, C2 (autoencoder), C3 (TICA / hTICA), C5 (k-means pass).  Everything lands under gpurun_out/.
#   tools/prof_round.sh <tag>
TAG="${1:-r03}"
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q > gpurun_out/t_${TAG}_final.log 2>&1; tail -2 gpurun_out/t_${TAG}_final.log; grep -E "^FAILED|^ERROR" gpurun_out/t_${TAG}_final.log | head
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
BENCH_ARGS="--batch 8192 --steps 400 --warmup 50 --no-cpu-baseline --large-batch 0 --other-mode-steps 0 --frames 2000000" bash tools/prof_small.sh ${TAG}_timeline | tail -9
echo "--- bench8k passes"
PMC_CMDLINE="bench.py --steps 100 --warmup 20 --no-cpu-baseline --large-batch 0 --other-mode-steps 0 --frames 2000000" \
  bash tools/prof_cmd.sh ${TAG}_bench8k bench.py --steps 800 --warmup 50 --no-cpu-baseline --large-batch 0 --other-mode-steps 0 --frames 5000000 | grep -E "exit" 
echo "--- benchL passes"
PMC_CMDLINE="bench.py --batch 524208 --steps 12 --warmup 3 --no-cpu-baseline --large-batch 0 --other-mode-steps 0 --frames 2000000 --profile-every 1" \
  bash tools/prof_cmd.sh ${TAG}_benchL bench.py --batch 524208 --steps 61 --warmup 5 --no-cpu-baseline --large-batch 0 --other-mode-steps 0 --profile-every 1 | grep -E "exit"
echo "--- c2"
PMC_CMDLINE="bench.py --config c2 --steps 100 --warmup 20 --no-cpu-baseline" bash tools/prof_cmd.sh ${TAG}_c2 bench.py --config c2 --steps 400 --warmup 50 --no-cpu-baseline | grep -E "exit"
echo "--- c3"
bash tools/prof_cmd.sh ${TAG}_c3 tools/bench_configs.py c3 | grep -E "exit"
echo "--- c5"
bash tools/prof_stats.sh ${TAG}_c5 tools/bench_configs.py c5 | head -3
python bench.py --config c2 > gpurun_out/b_c2_${TAG}.json 2> gpurun_out/b_c2_${TAG}.err; cut -c1-300 gpurun_out/b_c2_${TAG}.json
echo "--- done"
