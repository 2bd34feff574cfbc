import sys, time, subprocess, threading
sys.path.insert(0, '/root/repo')
import numpy as np
import slam_jl_amd as slam
from slam_jl_amd import synthetic as syn
H, W = 370, 1226
L, R, flows = syn.stereo_stream((H, W), 2, seed=0)
ctx = slam.default_context(0)
cur = slam.LKPyramid(shape=(H, W), levels=3)
def smi():
    time.sleep(1.0)
    for _ in range(3):
        out = subprocess.run(["rocm-smi", "--showclocks", "--showuse", "--showperflevel"], capture_output=True, text=True).stdout
        print("\n".join(l for l in out.splitlines() if any(k in l for k in ("sclk", "mclk", "fclk", "busy", "Perf", "socclk"))))
        time.sleep(0.7)
th = threading.Thread(target=smi); th.start()
t0 = time.time(); n = 0
while time.time() - t0 < 4.0:
    slam.update_(cur, L[0]); n += 1
th.join()
print("pyramids/s", n / 4.0)
