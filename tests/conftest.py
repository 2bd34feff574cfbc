import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def syn():
    import slam_jl_amd  # noqa: F401
    from slam_jl_amd import synthetic
    return synthetic


@pytest.fixture(scope="session")
def slam():
    """The product package with a live HIP context (GPU tests only)."""
    import slam_jl_amd
    slam_jl_amd.default_context(0)      # raises loudly when the HIP library / device is missing
    return slam_jl_amd


@pytest.fixture(scope="session")
def slam_host():
    """The product package WITHOUT a HIP context: host-side helpers only (CPU tests)."""
    import slam_jl_amd
    return slam_jl_amd


_TEX = {}


@pytest.fixture(scope="session")
def texture(syn):
    def get(H, W, n=2, seed=0, step=(1.3, -2.1), disparity=12.4):
        key = (H, W, n, seed, step, disparity)
        if key not in _TEX:
            _TEX[key] = syn.stereo_stream((H, W), n, seed, step, disparity)
        return _TEX[key]
    return get
