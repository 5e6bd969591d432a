import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import binding
    binding.build()  # (make: a no-op unless a source of the C restatement is newer than the library)
    return binding.Oracle()


@pytest.fixture(scope="session")
def reflib():
    from oracle.binding import RefLib, ref_available
    if not ref_available():
        pytest.skip("oracle/_ref not built and /root/reference absent")
    return RefLib()
