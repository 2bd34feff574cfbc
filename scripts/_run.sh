timeout 900 python bench.py --gpus 1 --steps 20 --warmup 3 > gpurun_out/drv.json 2> gpurun_out/drv.err; echo rc $?
python -c "
import json; d=json.loads(open('gpurun_out/drv.json').read().strip().split('\n')[-1]); print(d['metric'][:60], round(d['value']), d['steps'], d['warmup'], d['ms_per_step'], d['n_gpus'], d['scaling'], d['dtype'], d['config']['workload'][:80])"
