# Generic A/B over the small-network measurements for a 0 / 1 environment switch of the library.  Its last use, DCV_SNET_XFIRST
# (input rows requested ahead of the weight-image copy), was an experiment that was not kept and not committed.
# usage: ab_env.sh VAR  -- alternates VAR=0 / VAR=1 twice over the small-network measurements
V=$1
for rep in 1 2; do for x in 0 1; do
echo "== $V=$x (rep $rep)"
env $V=$x python bench.py --config ref_small --steps 600 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(' '.join('%s/%d:%.1f' % (r['network'], r['batch'], r['us_per_step']) for r in d['runs']))"
env $V=$x python tools/dbg/fit_ab.py 2>/dev/null | tail -1
env $V=$x python bench.py --config c2 --steps 800 --warmup 50 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2', round(d['value']/1e6,2), round(d['ms_per_step']*1e3,2))"
done; done
