#!/bin/bash
# instruction mix of k_ba_window (rocprofv3 --pmc, kernel-trace only): gpurun -- bash scripts/ba_window_pmc.sh
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmc_bw1 gpurun_out/pmc_bw2
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d gpurun_out/pmc_bw1 -- python3 scripts/probes/ba_batch_time.py 128 P5_free_20_const > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_FLAT SQ_INSTS_SMEM GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_bw2 -- python3 scripts/probes/ba_batch_time.py 128 P5_free_20_const > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
for d in ("gpurun_out/pmc_bw1", "gpurun_out/pmc_bw2"):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_ba_window" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(f"{k:24s} per launch {sum(v) / len(v):16.0f}   ({len(v)} launches)")
PY
rm -rf gpurun_out/pmc_bw1 gpurun_out/pmc_bw2
