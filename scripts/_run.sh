python -m pytest tests -m gpu -x -q 2>&1 | tail -4
python scripts/_dbg.py none > gpurun_out/dbg_stdout.txt 2> gpurun_out/dbg_stderr.txt; echo "stdout:"; cat gpurun_out/dbg_stdout.txt | head -5
