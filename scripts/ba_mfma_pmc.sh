#!/bin/bash
# k_schur_groups_m (the Schur products of the batched local BA on the matrix cores): instruction mix, matrix-core busy share and HBM bytes per launch of
# 128 x P20 windows (rocprofv3 --pmc in separate passes, kernel-trace only): gpurun -- bash scripts/ba_mfma_pmc.sh [out.txt]
OUT=${1:-gpurun_out/ba_mfma_pmc.txt}
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmc_m1 gpurun_out/pmc_m2 gpurun_out/pmc_m3 gpurun_out/pmc_m4
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d gpurun_out/pmc_m1 -- python3 scripts/probes/ba_batch_time.py 128 P20 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_m2 -- python3 scripts/probes/ba_batch_time.py 128 P20 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_m3 -- python3 scripts/probes/ba_batch_time.py 128 P20 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_m4 -- python3 scripts/probes/ba_batch_time.py 128 P20 > /dev/null 2>&1
python3 - > $OUT <<PY
import csv, glob, collections
print("rocprofv3 --pmc, 128 windows x P20 (40 000 observations each) per launch; averages over the launches of the active LM iterations")
for name in ("k_schur_groups_m", "k_update_groups_b", "k_schur_reduce_b"):
    print(name)
    for d in ("gpurun_out/pmc_m1", "gpurun_out/pmc_m2", "gpurun_out/pmc_m3", "gpurun_out/pmc_m4"):
        acc = collections.defaultdict(list)
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if name in r["Kernel_Name"]:
                    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in sorted(acc.items()):
            note = ""
            if k == "FETCH_SIZE": note = f"   = {sum(v) / len(v) * 1024 * 2 / 1e6:.1f} MB read (KB x 2: the guide's gfx950 correction)"
            if k == "WRITE_SIZE": note = f"   = {sum(v) / len(v) * 1024 / 1e6:.1f} MB written"
            print(f"  {k:28s} per launch {sum(v) / len(v):16.0f}   ({len(v)} launches){note}")
PY
cat $OUT
rm -rf gpurun_out/pmc_m1 gpurun_out/pmc_m2 gpurun_out/pmc_m3 gpurun_out/pmc_m4
