timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
timeout 600 python bench.py --no-cpu > gpurun_out/b.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/b.json').read().strip().split('\n')[-1]); print(round(d['value']), d['streams_sweep'], round(d['pose']['frontend_with_pose']['value']))"
