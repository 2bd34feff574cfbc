timeout 900 python -m pytest $(grep -ln "local_ba\|bundle_adjustment\|ShardedBA\|slam_ba" tests/test_gpu*.py) -x -q 2>&1 | tail -3
timeout 60 python scripts/prof_ba.py | tail -1
