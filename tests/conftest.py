"""Shared fixtures.

``be`` is the backend under test: the CPU oracle (pinning the restatement to
the reference's own known-answer tests; runs anywhere) and the HIP product
(``fidget_amd`` through the C ABI; ``@pytest.mark.gpu``).
"""
import os
import sys

import pytest

try:
    # torch bundles its own ROCm runtime: it has to be loaded before libfidget_hip.so pulls in the system one, or
    # torch.cuda finds "no HIP GPUs" later in the same process (tests that hand torch CUDA tensors to the library)
    import torch  # noqa: F401
except ImportError:
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MODELS = os.path.join(ROOT, "models")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _oracle():
    import oracle
    return oracle


def _hip():
    import fidget_amd
    return fidget_amd


@pytest.fixture(params=["oracle", pytest.param("hip", marks=pytest.mark.gpu)])
def be(request):
    return _oracle() if request.param == "oracle" else _hip()


@pytest.fixture
def oracle_mod():
    return _oracle()


def model_path(name):
    return os.path.join(MODELS, name)
