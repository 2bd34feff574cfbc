#!/bin/bash
# Exercises bench.py's self-spawning N > 1 path on a 1-GPU box: `python bench.py --gpus 2` starts two rank processes that share
# GPU 0 (SLAM_BENCH_ONE_GPU: LOCAL_RANK 0 for both, collectives over gloo).
SLAM_BENCH_ONE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 40 --warmup 6 --streams 8 --no-cpu > gpurun_out/two_rank.log 2>&1
echo "rc=$?"
tail -c 1500 gpurun_out/two_rank.log
