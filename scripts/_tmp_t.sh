cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gpu_ba_batch.py -x -q -m gpu 2>&1 | tail -2
python3 scripts/probes/ba_batch_time.py 128 P5_free_20_const 2>&1 | grep -v amdgpu | tail -1
timeout 500 python tests/fuzz/ba_fuzz.py 320 31000 16 2>&1 | tail -1
