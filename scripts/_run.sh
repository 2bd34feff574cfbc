timeout 800 python scripts/ba_repeat.py 300
