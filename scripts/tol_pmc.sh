#!/bin/bash
# HBM traffic of one batched build from the PMC counters, per kernel: bash scripts/tol_pmc.sh S MODE OUT.json  (MODE: u8 | u8tol)
# separate --pmc passes with --kernel-trace only (MI355X_MICROARCH.md; gpurun refuses --pmc with the trace domains)
S=${1:-128}; MODE=${2:-u8tol}; OUT=${3:-gpurun_out/pmc_${MODE}_s${S}.json}
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmc_f gpurun_out/pmc_w
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f -- python3 scripts/pmc_probe_batch.py $S $MODE > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w -- python3 scripts/pmc_probe_batch.py $S $MODE > /dev/null 2>&1
python scripts/pmc_batch_json.py gpurun_out/pmc_f gpurun_out/pmc_w $S $OUT "$(cat .commit_id 2>/dev/null || echo unrecorded)" $MODE | tail -12
python - <<PY
import json
j=json.load(open("$OUT"))
F=j["FETCH_SIZE"]["per_kernel_avg_KB"]; W=j["WRITE_SIZE"]["per_kernel_avg_KB"]
for k in sorted(set(F)|set(W), key=lambda k:-(F.get(k,0)*2+W.get(k,0)))[:10]:
    print(f"{k[:70]:70s} R {F.get(k,0)*2*1024/1e6:9.1f} MB  W {W.get(k,0)*1024/1e6:9.1f} MB")
PY
rm -rf gpurun_out/pmc_f gpurun_out/pmc_w
