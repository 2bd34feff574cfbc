"""A host of the C ABI that is neither Python nor PyTorch: tests/c_host/abi_host.c, plain C linked against libslamhip.so only, running
under the SYSTEM ROCm runtime (/opt/rocm/lib/libamdhip64) -- the executable stand-in for the Julia `ccall` shim, which cannot run here
(no Julia).  It calls the six seams of SURVEY 8b + slam_local_ba_batch + the pose seams with SLAMHip.jl's argument lists on raw
fixtures exported from tests/golden/*.npz and runs the three-task threading contract on three pthreads (reference: src/SLAM.jl:187-230).
The binary is started as a fresh child process (never an exec from a process that has touched the GPU)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "tests", "c_host")


def _build_and_export(tmp_path):
    r = subprocess.run(["make", "-C", HOST, "abi_host"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    fix = str(tmp_path / "fixtures.bin")
    r = subprocess.run([sys.executable, os.path.join(HOST, "export_fixtures.py"), fix], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    return os.path.join(HOST, "abi_host"), fix


def _clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ("LD_PRELOAD", "PYTHONPATH")}
    env.pop("LD_LIBRARY_PATH", None)            # nothing may steer the loader towards PyTorch's bundled HIP runtime
    return env


def test_c_host_builds_links_the_system_runtime_and_fails_loudly_without_a_device(tmp_path):
    """CPU: the C host compiles against include/slamhip.h with gcc alone, its libamdhip64 resolves to the system ROCm installation (not
    to PyTorch's bundled copy), and without a device it reports slam_ctx_create's error instead of computing anything"""
    exe, fix = _build_and_export(tmp_path)
    ldd = subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
    hip = [l for l in ldd.splitlines() if "libamdhip64" in l]
    assert hip and "/opt/rocm" in hip[0] and "torch" not in ldd, ldd
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("a HIP device is present: the gpu test runs the host in full")
    r = subprocess.run([exe, fix], capture_output=True, text=True, timeout=120, env=_clean_env())
    assert r.returncode != 0 and "no HIP device" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_c_host_runs_the_seams_under_the_system_rocm_runtime(tmp_path):
    exe, fix = _build_and_export(tmp_path)
    ldd = subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
    print(ldd)
    r = subprocess.run([exe, fix], capture_output=True, text=True, timeout=600, env=_clean_env())
    print(r.stdout[-6000:], r.stderr[-2000:])
    assert r.returncode == 0 and "ALL OK" in r.stdout, r.stdout[-4000:] + r.stderr[-2000:]
    assert "HIP runtime mapped: /opt/rocm" in r.stdout and "PyTorch libraries in this process: none" in r.stdout
    for who in ("front-end", "mapper", "estimator"):                 # the three tasks ran their seams
        assert f"[{who}]" in r.stdout
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "c_host.log"), "w") as f:
        f.write(ldd + "\n" + r.stdout)
