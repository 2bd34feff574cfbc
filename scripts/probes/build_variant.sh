#!/bin/bash
# A/B builds of ONE source of the library (slam.jl_amd/libslamhip_<tag>.so, selected with SLAMHIP_LIB=<path>):
#   scripts/build_variant.sh <file>.hip "<tag>:<hipcc -D flags>" ...      e.g.  scripts/build_variant.sh lk.hip "a3:-DLK_TMPL_LDS -DLK_WAVES=3"
set -e
cd "$(dirname "$0")/../../slam.jl_amd/csrc"
src="$1"; shift
base="${src%.hip}"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-unused-function"
OBJ=$(ls *.o | grep -v trace | grep -v "^$base\.o" | tr '\n' ' ')
for spec in "$@"; do
  tag="${spec%%:*}"; defs="${spec#*:}"
  /opt/rocm/bin/hipcc $FLAGS $defs -c $src -o /tmp/${base}_$tag.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libslamhip_$tag.so $OBJ /tmp/${base}_$tag.o -ldl
  echo built ../libslamhip_$tag.so "($defs)"
done
