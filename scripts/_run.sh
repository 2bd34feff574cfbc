cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_lk1 -o a -- python3 $GRAFT_REPO_ROOT/scripts/prof_flow.py 32 > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_WAVES SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_lk2 -o b -- python3 $GRAFT_REPO_ROOT/scripts/prof_flow.py 32 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
(python scripts/pmc_kernel.py gpurun_out/pmc_lk1 flow_match; python scripts/pmc_kernel.py gpurun_out/pmc_lk2 flow_match) > gpurun_out/r03b_pmc_lk_flow_match.txt
cat gpurun_out/r03b_pmc_lk_flow_match.txt
rm -rf gpurun_out/pmc_lk1 gpurun_out/pmc_lk2
