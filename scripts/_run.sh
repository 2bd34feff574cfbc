timeout 300 python -m pytest tests/test_gpu_batch.py -x -q 2>&1 | tail -2
for i in 1 2; do
SLAM_BENCH_RIGHT_FULL=1 timeout 200 python scripts/prof_headline.py host_u8 2>&1 | tail -1 | sed "s/^/full   /"
timeout 200 python scripts/prof_headline.py host_u8 2>&1 | tail -1 | sed "s/^/target /"
done
