#!/bin/bash
# memory traffic + instruction mix of k_kpset_match alone (rocprofv3 --pmc, kernel-trace only): gpurun -- bash scripts/lk_pmc.sh [exact|tol]
MODE=${1:-exact}
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for P in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"; do
  rm -rf gpurun_out/pmc_lk
  timeout 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d gpurun_out/pmc_lk -- python3 scripts/probes/prof_kpset_match.py 128 $MODE 3 > gpurun_out/lk_pmc_run.log 2>&1
  python3 scripts/pmc_kernel.py gpurun_out/pmc_lk k_kpset_match
done
tail -1 gpurun_out/lk_pmc_run.log
rm -rf gpurun_out/pmc_lk
