# in-frame A/B of the sweep kernels (HIP events around the two eager sweep launches of the replayed frame)
for v in 1 0; do BMV_SWEEP_WIN=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); lv=d['roofline']['levels']; print('BMV_SWEEP_WIN=$v', round(d['value'],1), {k:(round(x['avg_us'],2), round(x['min_us'],2)) for k,x in lv.items()})"; done
