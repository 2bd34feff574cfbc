for v in ckbase cknt ckbase cknt; do
SLAMHIP_LIB=$PWD/slam.jl_amd/libslamhip_$v.so python scripts/prof_pyr_batch.py 64 30 u8 2>&1 | tail -1
done
