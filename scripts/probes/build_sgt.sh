#!/bin/bash
# tracing build of ba.hip (per-phase clocks of k_schur_groups printed from the device): slam.jl_amd/libslamhip_sgt.so, used via SLAMHIP_LIB
set -e
cd "$(dirname "$0")/../slam.jl_amd/csrc"
make >/dev/null
mkdir -p /tmp/bas
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -DSG_TRACE -c ba.hip -o /tmp/bas/ba.sgt.o
hipcc --offload-arch=gfx950 -shared -fPIC -o ../libslamhip_sgt.so /tmp/bas/ba.sgt.o $(ls *.o | grep -v "^ba.o\|trace") -ldl
