for extra in "" "--native-rccl"; do
DCV_FORCE_DIST=1 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --other-mode-steps 0 --shuffled-steps 0 --c2-steps 0 --ref-small-steps 0 --large-steps 20 $extra 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
ct=d['config']['collective_timing']
print('$extra', 'dp world1:', round(d['value']/1e6,2), 'M', round(d['ms_per_step']*1e3,1), 'us/step; stats allreduce', round(ct['statistics']['us_per_allreduce'],1), 'us, grads', round(ct['gradients']['us_per_allreduce'],1), 'us; large', round(d['large_batch']['value']/1e6,1) if 'large_batch' in d else None)"
done
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --other-mode-steps 0 --shuffled-steps 0 --c2-steps 0 --ref-small-steps 0 --large-steps 20 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('single process:', round(d['value']/1e6,2), 'M', round(d['ms_per_step']*1e3,1), 'us/step; large', round(d['large_batch']['value']/1e6,1))"
