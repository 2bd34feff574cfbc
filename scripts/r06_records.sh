#!/bin/bash
# the round's records in one gpurun call: gpurun -- bash scripts/r06_records.sh TAG
TAG=${1:-r06a}
git rev-parse --short HEAD > .commit_id 2>/dev/null
python bench.py --steps 20 --warmup 5 2> gpurun_out/${TAG}_bench.err | tail -1 > gpurun_out/${TAG}_bench.json; echo "bench rc=$?"
cp bench_detail.json gpurun_out/${TAG}_bench_detail.json
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for leg in headline tolbatch ba; do
  rm -rf gpurun_out/prof_$leg
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$leg -o $leg -- python3 bench.py --only $leg --steps 20 --warmup 5 > gpurun_out/${TAG}_${leg}_profiled.json 2>/dev/null
  cp gpurun_out/prof_$leg/${leg}_kernel_stats.csv gpurun_out/${TAG}_${leg}_kernel_stats.csv
  tail -1 gpurun_out/${TAG}_${leg}_profiled.json > gpurun_out/${TAG}_${leg}_profiled.json.tmp && mv gpurun_out/${TAG}_${leg}_profiled.json.tmp gpurun_out/${TAG}_${leg}_profiled.json
  rm -rf gpurun_out/prof_$leg
done
bash scripts/tol_pmc.sh 128 u8 gpurun_out/${TAG}_pmc_pyramid_batch_s128.json | tail -4
bash scripts/tol_pmc.sh 128 u8tol gpurun_out/${TAG}_pmc_pyramid_tol_batch_s128.json | tail -4
head -c 600 gpurun_out/${TAG}_bench.json; echo
bash scripts/ba_mfma_pmc.sh gpurun_out/${TAG}_ba_mfma_pmc.txt | tail -40
bash scripts/ba_batch_prof.sh 128 ${TAG}_p20 P20 | head -12 > gpurun_out/${TAG}_ba_batch_p20_kernels.txt; cat gpurun_out/${TAG}_ba_batch_p20_kernels.txt
