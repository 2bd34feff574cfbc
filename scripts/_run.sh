timeout 60 python scripts/prof_ba.py 2>&1 | tail -2
