for L in 0 40960 20480; do
SLAMHIP_ROWS_LDS=$L timeout 300 python bench.py --no-cpu --no-ba --no-sweep --steps 100 --warmup 10 > gpurun_out/bl_$L.json 2>/dev/null; echo "lds $L rc $?"
done
