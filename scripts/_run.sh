python -m pytest tests/test_gpu_ba.py tests/test_gpu_edges.py tests/test_gpu_configs.py tests/test_golden.py -x -q 2>&1 | tail -3
python scripts/prof_ba.py 50 10000 | tail -1
python scripts/prof_ba.py 100 40000 | tail -1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ba -o ba -- python3 $R/scripts/prof_ba.py 50 10000 > /dev/null 2>&1
cd $R
head -14 gpurun_out/prof_ba/ba_kernel_stats.csv | cut -c1-120
