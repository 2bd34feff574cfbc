"""Random shapes and parameters through the front-end kernels against the oracle (tests/fuzz/fuzz_frontend.py: single / batched pyramids
with the bandwidth-bound kernel set forced on small shapes, fb_tracking with random windows / levels / border points, detect with random
cells / keypoint lists, describe on border keypoints) -- a short run of what the script does at length (2 800 trials at the end of round 3, no failure)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_frontend_fuzz_short_run():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz", "fuzz_frontend.py"), "25", "5000"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "parts ['pyr', 'batch', 'tolpyr', 'tolbatch', 'lk', 'detect', 'brief']: 0 failures" in r.stdout, r.stdout[-3000:]


def test_keypoint_set_fuzz_short_run():
    """tests/fuzz/kpset_fuzz.py: key-frame steps on the device-resident lists with random stream counts, shapes and list sizes (empty
    streams, full lists, a stream with everything culled) against the host protocol and the oracle (560 steps at the end of round 3)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz", "kpset_fuzz.py"), "15", "7000"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "15 key-frame steps, 0 failures" in r.stdout, r.stdout[-3000:]


def test_ba_batch_fuzz_short_run():
    """tests/fuzz/ba_fuzz.py in batch mode: 48 random ragged windows through slam_local_ba_batch, 12 per call, against the oracle"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz", "ba_fuzz.py"), "48", "11000", "12"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0 and " 0 failures" in r.stdout, r.stdout[-1500:] + r.stderr[-500:]
