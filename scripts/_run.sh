timeout 900 python bench.py > gpurun_out/final.json 2> gpurun_out/final.err; echo bench rc $?
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -o bench -- python3 bench.py --no-cpu > gpurun_out/final_profiled.json 2> /dev/null; echo prof rc $?
