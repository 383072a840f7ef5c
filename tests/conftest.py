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


def pytest_report_header(config):
    """Names the one environmental reason a device-against-oracle comparison of a transcendental tape can fail: the oracle calls THIS host's
    libm, the device restates glibc 2.35's x86-64 routines (fhip_libm_probe, include/fidget_hip.h)."""
    try:
        n, first = _hip().libm_probe()
    except Exception as e:       # noqa: BLE001  (library not built yet: the build check says so elsewhere)
        return f"fidget-hip libm probe: not run ({e!r})"
    if n == 0:
        return "fidget-hip libm probe: this host's libm is the one the device restates (0 of 288 probe arguments differ)"
    return (f"fidget-hip libm probe: MISMATCH - {n} of 288 probe arguments differ, first {first}.  Tests that compare transcendental "
            "values of the device (or of trans_libm.hpp) with the oracle / the running libm will fail for THIS reason.")
