# A/B script of an experiment that was NOT kept: DCV_OPT_PREFETCH exists only in commit 6cd40db (reverted by 58f0bcd).
for rep in 1 2; do for pf in 0 1; do
echo "== DCV_OPT_PREFETCH=$pf (rep $rep)"
DCV_OPT_PREFETCH=$pf python bench.py --config ref_small --steps 600 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(' '.join(f\"{r['network']}/{r['batch']}:{r['us_per_step']:.1f}\" for r in d['runs']))"
DCV_OPT_PREFETCH=$pf python bench.py --config c2 --steps 800 --warmup 50 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2', round(d['value']/1e6,2), round(d['ms_per_step']*1e3,2))"
DCV_OPT_PREFETCH=$pf python bench.py --steps 400 --warmup 20 --no-cpu-baseline --other-mode-steps 0 --shuffled-steps 0 --c2-steps 0 --ref-small-steps 0 --large-batch 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c4', round(d['value']/1e6,2), round(d['ms_per_step']*1e3,2))"
done; done
