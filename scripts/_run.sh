python -m pytest tests/test_gpu_batch.py tests/test_gpu_pyramid.py -x -q 2>&1 | tail -15
python scripts/prof_pyr_batch.py 32
python scripts/prof_pyr_batch.py 16
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_fused -o fused -- python3 $R/scripts/prof_pyr_batch.py 32 10 > /dev/null 2>&1
cd $R
for f in $(find gpurun_out/prof_fused -name "*kernel_stats.csv"); do echo == $f; head -12 $f | cut -c1-200; done
