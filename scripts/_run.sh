timeout 1500 python bench.py > gpurun_out/r02g_bench.json 2> gpurun_out/r02g_bench.err; echo rc $?
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r02g -o p -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu > $GRAFT_REPO_ROOT/gpurun_out/r02g_bench_profiled.json 2> $GRAFT_REPO_ROOT/gpurun_out/r02g_prof.err; echo rc $?
cd $GRAFT_REPO_ROOT; find gpurun_out/prof_r02g -name "*kernel_stats.csv" -exec cp {} gpurun_out/r02g_bench_kernel_stats.csv \; ; rm -rf gpurun_out/prof_r02g
