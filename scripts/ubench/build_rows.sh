#!/bin/bash
# scratch: builds scripts/ubench/rows_ck_bench against the library objects
cd "$(dirname "$0")"
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Wno-unused-result $EXTRA -I../../slam.jl_amd/csrc -c -o /tmp/rows_ck_bench.o rows_ck_bench.hip > /tmp/cc.log 2>&1 || { grep -A5 error /tmp/cc.log | head -30; exit 1; }
hipcc --offload-arch=gfx950 -o ${OUT:-rows_ck_bench} /tmp/rows_ck_bench.o ../../slam.jl_amd/csrc/{ctx,detect,lk,ba,brief,triangulate,pose,fivepoint,comm,kpset}.o -ldl > /tmp/ld.log 2>&1 || { head -c 800 /tmp/ld.log; exit 1; }
