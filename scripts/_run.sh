cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py > gpurun_out/bench_plain.json 2> gpurun_out/bench_plain.err; echo plain rc $?
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -o bench -- python3 bench.py --no-cpu > gpurun_out/bench_profiled.json 2> gpurun_out/bench_profiled.err; echo prof rc $?
ls gpurun_out/prof_bench | head
