timeout 900 python -X faulthandler bench.py --steps 100 --warmup 10 --no-cpu > gpurun_out/r02n_bench.json 2> gpurun_out/r02n_bench.err
echo rc=$?
grep -v "^{" gpurun_out/r02n_bench.json | head -3; tail -6 gpurun_out/r02n_bench.err
python - <<'PY'
import json
lines=[l for l in open('gpurun_out/r02n_bench.json').read().strip().splitlines() if l.startswith('{')]
j=json.loads(lines[-1])
print({k:(round(v,1) if isinstance(v,float) else v) for k,v in j.items() if k in ('value','ms_per_step')})
print('ingest', {k:(round(v['value']), round(v['pyramid_build_ms_mean'],3)) for k,v in j['ingest'].items()})
print('roofline frac', j['roofline']['frac'], 'build us', j['roofline']['avg_launch_us'], j['roofline']['min_launch_us'])
print('host_protocol', j.get('host_protocol',{}).get('value'))
print('single', j['single_stream']['value'], 'ba', j['ba']['ms_per_iter'])
print('ba_sharded', j.get('ba_sharded'))
PY
