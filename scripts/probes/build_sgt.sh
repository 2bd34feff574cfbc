#!/bin/bash
# tracing build of the BA sources (per-phase clocks of the point-group builds printed from the device; `make -C slam.jl_amd/csrc trace` builds all tracing macros): slam.jl_amd/libslamhip_sgt.so, used via SLAMHIP_LIB
set -e
cd "$(dirname "$0")/../../slam.jl_amd/csrc"
make >/dev/null
mkdir -p /tmp/bas
for f in ba_single ba_batch; do hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -mllvm -amdgpu-mfma-vgpr-form=1 -DSG_TRACE -c $f.hip -o /tmp/bas/$f.sgt.o; done
hipcc --offload-arch=gfx950 -shared -fPIC -o ../libslamhip_sgt.so /tmp/bas/ba_single.sgt.o /tmp/bas/ba_batch.sgt.o $(ls *.o | grep -v "^ba_single.o\|^ba_batch.o\|trace") -ldl
