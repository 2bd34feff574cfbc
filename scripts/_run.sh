R=$GRAFT_REPO_ROOT
timeout 900 python bench.py > gpurun_out/r02b_bench.json 2> gpurun_out/r02b_bench.err
echo rc=$? lines=$(wc -l < gpurun_out/r02b_bench.json)
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -o b -- python3 $R/bench.py --no-cpu > $R/gpurun_out/r02b_bench_profiled.json 2> $R/gpurun_out/r02b_prof.err
echo rc=$?
cd $R
cp gpurun_out/prof_bench/b_kernel_stats.csv gpurun_out/r02b_bench_kernel_stats.csv
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r02b_bench.json').read().strip().splitlines()[-1])
print({k:(round(v,1) if isinstance(v,float) else v) for k,v in j.items() if k in ('value','ms_per_step','steps')})
print('ingest', {k:(round(v['value']), round(v['pyramid_build_ms_mean'],3)) for k,v in j['ingest'].items()})
r=j['roofline']; print('roofline', r['frac'], r['avg_launch_us'], r['isolated_launch_us'], r['frac_isolated'], r['serial_launches_us'], r['kernel']['avg_launch_us'], r['kernel']['frac'])
print('host_protocol', j['host_protocol']['value'], 'single', j['single_stream']['value'], 'tol', j['tolerance_mode'].get('single_stream'))
print('ba', j['ba']['ms_per_iter'], j['ba'].get('cpu_ms_per_iter_schur'), 'sharded', j['ba_sharded'].get('ms_per_iter_wall'))
print('cpu', j['cpu_baseline']['value'], j['cpu_baseline']['sample'][:60])
print('pose', j['pose']['ms_per_call'], j['pose']['five_point']['ms_per_call'], j['pose']['batch']['ms_per_step'], j['pose'].get('frontend_with_pose',{}).get('value'))
PY
