cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d gpurun_out/pmc_lk -o lk -- python3 scripts/prof_flow.py 32 > /dev/null 2>&1; echo rc $?
python scripts/pmc_kernel.py gpurun_out/pmc_lk flow_match
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_lk2 -o lk -- python3 scripts/prof_flow.py 32 > /dev/null 2>&1; echo rc $?
python scripts/pmc_kernel.py gpurun_out/pmc_lk2 flow_match
