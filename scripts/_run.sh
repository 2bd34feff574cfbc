timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
timeout 600 python bench.py --no-cpu > gpurun_out/b.json 2> gpurun_out/b.err; echo rc $?
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ba -o ba -- python3 scripts/prof_ba.py > /dev/null 2>&1
head -8 gpurun_out/prof_ba/ba_kernel_stats.csv | cut -c1-110
