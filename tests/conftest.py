import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def O():
    """The CPU oracle (test infrastructure)."""
    from oracle import oracle as orc
    orc.build()
    return orc


@pytest.fixture(scope="session")
def smt():
    """The product package; on a GPU box the HIP library must load (no fallback)."""
    import torch
    import stereo_match_traditional_amd as pkg
    from stereo_match_traditional_amd._lib import lib
    lib()  # raises loudly when libsmt_hip.so is missing
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but torch.cuda.is_available() is False")
    return pkg
