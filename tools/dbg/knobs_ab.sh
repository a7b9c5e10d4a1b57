# contract-batch step under each launch-time knob flipped from its default (one box, 600 steps each, baseline in between)
run() { env "$@" python bench.py --steps 600 --warmup 30 --no-cpu-baseline --other-mode-steps 0 --shuffled-steps 0 --c2-steps 0 --ref-small-steps 0 --fit-epochs 0 --large-batch 0 --frames 4000000 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f us/step' % (d['ms_per_step']*1e3))"; }
for k in BASE=1 DCV_XCD_REMAP=0 DCV_WT=0 BASE=2 DCV_TAIL_KSPLIT=0 DCV_NO_PAIR=1 DCV_NO_HEAD_FUSION=1 BASE=3 DCV_NO_ACT_MASK=1 DCV_NO_QUARTER_NT=1 DCV_REDUCE_QUAD=0 BASE=4; do
  printf "%-24s " $k; run $k
done
