for rep in 1 2; do
for A in 2 3; do
  SLAM_BENCH_KP_AHEAD=$A timeout 300 python bench.py --no-cpu --no-ba --no-sweep --steps 100 --warmup 10 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('ahead $A', round(d['value']), d['roofline']['frac'])"
done; done
