R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_f -o f -- python3 $R/scripts/pmc_probe_batch.py 32 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_w -o w -- python3 $R/scripts/pmc_probe_batch.py 32 > /dev/null 2>&1
cd $R
python scripts/pmc_batch_json.py gpurun_out/pmc_f gpurun_out/pmc_w 32 gpurun_out/r02c_pmc_pyramid_batch.json | grep "all_pyramid"
rm -rf gpurun_out/pmc_f gpurun_out/pmc_w
