#!/bin/bash
# tracing build of detect.hip (per-phase clocks of one cell printed from the device): slam.jl_amd/libslamhip_dett.so, used via SLAMHIP_LIB
set -e
cd "$(dirname "$0")/../../slam.jl_amd/csrc"
make >/dev/null
mkdir -p /tmp/bas
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -DDET_TRACE -c detect.hip -o /tmp/bas/detect.t.o
hipcc --offload-arch=gfx950 -shared -fPIC -o ../libslamhip_dett.so /tmp/bas/detect.t.o $(ls *.o | grep -v "^detect.o\|trace") -ldl
