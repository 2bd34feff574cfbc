#!/bin/bash
# A/B builds of the tracking kernels (slam.jl_amd/libslamhip_lk_<tag>.so, selected with SLAMHIP_LIB=<path>):
#   scripts/build_lk_variants.sh "<tag>:<hipcc -D flags>" ...      e.g.  "a3:-DLK_TMPL_LDS -DLK_WAVES=3"
set -e
cd "$(dirname "$0")/../slam.jl_amd/csrc"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-unused-function"
OBJ=$(ls *.o | grep -v trace | grep -v '^lk' | tr '\n' ' ')
for spec in "$@"; do
  tag="${spec%%:*}"; defs="${spec#*:}"
  /opt/rocm/bin/hipcc $FLAGS $defs -c lk.hip -o /tmp/lk_$tag.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libslamhip_lk_$tag.so $OBJ /tmp/lk_$tag.o -ldl
  echo built ../libslamhip_lk_$tag.so "($defs)"
done
