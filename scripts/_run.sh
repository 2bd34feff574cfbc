timeout 300 python -m pytest tests/test_gpu_batch.py tests/test_gpu_pyramid.py -x -q 2>&1 | tail -1
for i in 1 2; do
SLAMHIP_LIB=$PWD/scripts/ubench/libslamhip_base.so python scripts/prof_pyr_batch.py 32 30 u8 2>&1 | tail -1 | cut -c1-60 | sed 's/^/base /'
python scripts/prof_pyr_batch.py 32 30 u8 2>&1 | tail -1 | cut -c1-60 | sed 's/^/new  /'
done
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d $R/gpurun_out/kt -o p --output-format csv -- python3 $R/scripts/prof_pyr_batch.py 32 10 > /dev/null 2>&1
python3 $R/scripts/kernel_durations.py $R/gpurun_out/kt | grep cum_fused | head -3; rm -rf $R/gpurun_out/kt
