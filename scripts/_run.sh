timeout 60 python scripts/prof_ba.py 2>&1 | tail -2
timeout 60 python scripts/prof_ba.py 100 40000 | tail -1
timeout 900 python -m pytest $(grep -ln "local_ba\|bundle_adjustment\|ShardedBA\|slam_ba" tests/test_gpu*.py) -x -q 2>&1 | tail -3
