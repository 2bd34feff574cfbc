#!/bin/bash
# kernel timeline of ONE single-image build (exact and tolerance): gpurun -- bash scripts/single_build_timeline.sh
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for m in tol exact; do
  rm -rf gpurun_out/tl_$m
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl_$m -o tl -- python3 scripts/probes/single_tol_build.py $m 20 2>/dev/null | tail -1
  python3 scripts/probes/last_build_timeline.py gpurun_out/tl_$m
  rm -rf gpurun_out/tl_$m
done
