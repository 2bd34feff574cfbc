SLAMHIP_BA_HOSTTIME=1 timeout 60 python scripts/prof_ba.py 2>&1 | grep "host:" | tail -2
SLAMHIP_BA_HOSTTIME=1 timeout 60 python scripts/prof_ba.py 20 4000 2>&1 | tail -2
