#!/bin/bash
# Exercises bench.py's N>1 code path on a 1-GPU box: two ranks share GPU 0, collectives over gloo.
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 WORLD_SIZE=2 SLAM_BENCH_BACKEND=gloo LOCAL_RANK=0
RANK=1 python bench.py --gpus 2 --steps 60 --warmup 10 --streams 8 > gpurun_out/rank1.log 2>&1 &
P1=$!
RANK=0 timeout 500 python bench.py --gpus 2 --steps 60 --warmup 10 --streams 8 > gpurun_out/rank0.log 2>&1
RC=$?
wait $P1
echo "rc0=$RC rc1=$?"
tail -c 1800 gpurun_out/rank0.log
tail -3 gpurun_out/rank1.log
