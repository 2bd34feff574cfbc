timeout 1500 python bench.py > gpurun_out/r02f_bench.json 2> gpurun_out/r02f_bench.err; echo rc $?
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r02f -o p -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu > $GRAFT_REPO_ROOT/gpurun_out/r02f_bench_profiled.json 2> $GRAFT_REPO_ROOT/gpurun_out/r02f_prof.err; echo rc $?
cd $GRAFT_REPO_ROOT; find gpurun_out/prof_r02f -name "*kernel_stats.csv" -exec cp {} gpurun_out/r02f_bench_kernel_stats.csv \; ; rm -rf gpurun_out/prof_r02f
