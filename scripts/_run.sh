python bench.py --only single --steps 40 --warmup 4 > gpurun_out/b_single.json 2> gpurun_out/b_single.err; echo rc $?; tail -2 gpurun_out/b_single.err
python -c "
import json; d=json.loads(open('gpurun_out/b_single.json').read().strip().split('\n')[-1]); s=d['single_stream']; print(s['value'], s['by_builds_in_flight'], s['tracked_kpts_per_frame'])"
python -m pytest tests/test_gpu_batch.py tests/test_gpu_pyramid.py -x -q 2>&1 | tail -2
