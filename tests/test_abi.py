"""CPU: the C-ABI library loads and exports every symbol include/slamhip.h
declares; the Python binding table matches the header; the product path fails
loudly (no CPU fallback) when no HIP device is present."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "slamhip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(slam_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_seams():
    names = _declared()
    for n in ("slam_detect", "slam_describe", "slam_pyr_create", "slam_pyr_update", "slam_pyr_copy", "slam_pyr_clone",
              "slam_fb_track", "slam_local_ba", "slam_pnp_ba", "slam_ba_build", "slam_ba_solve", "slam_last_error"):
        assert n in names


def test_library_exports_every_declared_symbol():
    import ctypes
    import slam_jl_amd
    assert os.path.exists(slam_jl_amd.LIB_PATH), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = ctypes.CDLL(slam_jl_amd.LIB_PATH)
    for name in _declared():
        assert hasattr(lib, name), f"{name} declared in slamhip.h but not exported"


def test_binding_table_matches_header():
    from slam_jl_amd import _lib
    assert sorted(_lib.SIGNATURES) == _declared()
    lib = _lib.load()
    assert lib.slam_version().decode().startswith("slamhip")
    assert lib.slam_ba_reduce_len(50) == 300 * 300 + 600 + 8        # host-only helper, no GPU needed


def test_no_cpu_fallback_without_device():
    import torch
    import slam_jl_amd
    if torch.cuda.is_available():
        pytest.skip("a HIP device is present")
    with pytest.raises(slam_jl_amd.SlamHipError, match="no HIP device|slam_ctx_create"):
        slam_jl_amd.Context(0)


def test_product_never_imports_the_oracle():
    """the package, the examples and the measurement scripts: the oracle is the checker of tests/, smoke() and bench.py's cpu_baseline leg
    only (the randomised parity runs that use it live under tests/fuzz/)"""
    for sub in ("slam.jl_amd", "scripts", "examples", "benchlib"):        # benchlib: bench.py's parts take the oracle module as an ARGUMENT from the cpu_baseline leg
        pkg = os.path.join(ROOT, sub)
        for dirpath, _, files in os.walk(pkg):
            for f in files:
                if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h", ".jl", ".sh")):
                    txt = open(os.path.join(dirpath, f), errors="ignore").read()
                    assert "slam_oracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, os.path.join(sub, f)
    # bench.py itself: exactly one import site, inside the cpu_baseline leg (rank 0, N = 1)
    txt = open(os.path.join(ROOT, "bench.py")).read()
    assert txt.count("from oracle import") == 1 and txt.index("from oracle import") > txt.index('"cpu" in legs:')


def test_bench_batch_sizes_fit_the_library_limit():
    """bench.py's streams-per-GPU (default and per workload) must not exceed what a batch / keypoint set takes (include/slamhip.h:
    slam_pyr_create_batch's 1 <= S <= 128; csrc/pyramid.hip BATCH_MAX)."""
    import importlib.util, os, re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "slamhip.h")).read()
    m = re.search(r"1 <= S <= (\d+)", hdr)
    assert m, "the header documents the batch limit"
    limit = int(m.group(1))
    src = open(os.path.join(root, "slam.jl_amd", "csrc", "pyramid.hip")).read()
    assert int(re.search(r"#define BATCH_MAX (\d+)", src).group(1)) == limit
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
    assert all(w["S"] <= limit for w in b.WORKLOADS.values())
    assert int(re.search(r'"--streams", type=int, default=(\d+)', open(os.path.join(root, "bench.py")).read()).group(1)) <= limit


def test_no_unbuilt_conditional_code_in_the_sources():
    """The library's only conditionally compiled bodies are its tracing macros (csrc/Makefile: TRACE_DEFS); every one of them still compiles
    (syntax check of host + device code with all of them defined), and no other #ifdef / #ifndef names a macro nobody defines."""
    import subprocess
    csrc = os.path.join(ROOT, "slam.jl_amd", "csrc")
    mk = open(os.path.join(csrc, "Makefile")).read()
    trace = re.search(r"TRACE_DEFS = (.*)", mk).group(1).replace("-D", "").split()
    for f in sorted(os.listdir(csrc)):
        if not f.endswith((".hip", ".hpp")):
            continue
        for name in re.findall(r"^\s*#\s*(?:ifdef|ifndef)\s+(\w+)", open(os.path.join(csrc, f)).read(), flags=re.M):
            assert name in trace, f"{f}: #ifdef {name} is neither a tracing macro of the Makefile nor built by anything"
    jobs = [subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-std=c++17", "-ffp-contract=off", "-Wno-unused-function", "-fsyntax-only"]
                             + ["-D" + t for t in trace] + [os.path.join(csrc, f)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            for f in ("ba_single.hip", "ba_batch.hip", "ba_window.hip", "lk.hip", "detect.hip")]
    for j in jobs:
        out, _ = j.communicate(timeout=900)
        assert j.returncode == 0 and "error" not in out, out[-3000:]
