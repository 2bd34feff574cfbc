cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gpu_ba_batch.py -x -q -m gpu 2>&1 | tail -4
python3 scripts/probes/ba_batch_time.py 128 P5_free_20_const 2>&1 | grep -v amdgpu | tail -1
SLAMHIP_BA_WINDOW_ONE=1 python3 scripts/probes/ba_batch_time.py 128 P5_free_20_const 2>&1 | grep -v amdgpu | tail -1
