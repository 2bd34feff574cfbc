cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
S_=64 timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_hl -o hl -- python3 scripts/prof_headline.py > gpurun_out/hl.txt 2>&1
tail -2 gpurun_out/hl.txt
python scripts/queue_gaps.py gpurun_out/prof_hl
