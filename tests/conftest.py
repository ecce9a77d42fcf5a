import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "jetracer-orbslam2_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle  # oracle/oracle.py -- the checker, test infrastructure only
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def gpu():
    """torch + the HIP library on cuda:0; fails loudly if either is missing."""
    import torch
    import orbfe
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    built = [orbfe.LIB_PATH, os.path.join(ROOT, "jetracer-orbslam2_amd", "liborbfe_dist.so"),
             os.path.join(ROOT, "examples", "buildstream_port"), os.path.join(ROOT, "examples", "match_port"),
             os.path.join(ROOT, "examples", "multi_gpu_port"), os.path.join(ROOT, "examples", "ingest_port")]
    if not all(os.path.exists(b) for b in built):
        import __graft_entry__  # clean checkout: build the HIP library, the oracle and the example
        __graft_entry__.build()
    orbfe.lib()  # raises if liborbfe.so is not built
    assert orbfe.lib().orbfe_device_count() >= 1
    return torch, orbfe
