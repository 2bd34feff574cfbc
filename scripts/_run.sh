for i in 1 2; do
python scripts/prof_pyr_batch.py 32 30 u8 2>&1 | tail -1 | cut -c1-60 | sed 's/^/base /'
SLAMHIP_SPLIT_DAG=1 python scripts/prof_pyr_batch.py 32 30 u8 2>&1 | tail -1 | cut -c1-60 | sed 's/^/split /'
done
SLAMHIP_SPLIT_DAG=1 timeout 300 python -m pytest tests/test_gpu_batch.py -x -q 2>&1 | tail -1
timeout 200 python scripts/prof_headline.py host_u8 2>&1 | tail -1 | sed "s/^/base /"
SLAMHIP_SPLIT_DAG=1 timeout 200 python scripts/prof_headline.py host_u8 2>&1 | tail -1 | sed "s/^/split /"
