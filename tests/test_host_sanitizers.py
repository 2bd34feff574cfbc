"""CPU: the host half of the local-BA batch path -- ba_plan / ba_emit (structure analysis, pose re-ordering, point groups, staging layout) and
the parked worker pool two estimator threads share -- under AddressSanitizer + UndefinedBehaviorSanitizer and under ThreadSanitizer
(tests/host_sanitize/: csrc/ba_host.hip, ba_batch.hip, ba_single.hip, ba_window.hip + csrc/ctx.hip compiled with the sanitizers on the host code, driven by two caller threads on ragged random
windows incl. loop closures, all-constant, empty and duplicate-observation windows; no GPU, nothing is launched).  It is the only
multi-threaded C++ of the product (reference: the arrays are those of src/estimator.jl:143-266)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HS = os.path.join(ROOT, "tests", "host_sanitize")


@pytest.fixture(scope="module")
def drivers():
    r = subprocess.run(["make", "-C", HS, "-j2", "all"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return HS


@pytest.mark.parametrize("which", ["driver_asan", "driver_tsan"])
def test_host_half_is_clean_under_the_sanitizers(drivers, which):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", TSAN_OPTIONS="halt_on_error=0", SLAMHIP_BA_THREADS="6")
    env.pop("LD_PRELOAD", None)
    r = subprocess.run([os.path.join(drivers, which)], capture_output=True, text=True, timeout=600, env=env)
    log = r.stdout + r.stderr
    assert r.returncode == 0 and "clean" in r.stdout, log[-4000:]
    for bad in ("AddressSanitizer", "ThreadSanitizer", "runtime error", "LeakSanitizer"):
        assert bad not in log, log[-4000:]
