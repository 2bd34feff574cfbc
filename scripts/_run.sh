python scripts/prof_single_loop.py 1 2>&1 | tail -1
python scripts/prof_single_loop.py 3 2>&1 | tail -1
