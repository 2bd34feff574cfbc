for rep in 1 2; do
for PR in 0 1; do
  SLAM_BENCH_PYR_PRIO=$PR timeout 300 python bench.py --no-cpu --no-ba --no-sweep --steps 100 --warmup 10 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('pyr prio $PR', round(d['value']), d['roofline']['frac'])"
done; done
