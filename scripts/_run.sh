timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/final.json 2> gpurun_out/final.err; echo bench rc $?
