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
    import oracle as orc
    return orc.Oracle()


@pytest.fixture(scope="session")
def reflib():
    import oracle as orc
    if not orc.ref_available():
        pytest.skip("oracle/_ref/libslowflow_ref.so not built (needs /root/reference)")
    return orc.RefLib()


class _Switches:
    """the library's cross-check / what-if switches through the test hook sfa_debug_set (the library does not read them from the environment
    without SFA_DEBUG=1); everything set here is back at its default when the test ends"""

    def __init__(self):
        self.touched = set()

    def set(self, name, value):
        import slowflow_amd as sfa
        sfa.debug_set(name, value)
        self.touched.add(name)

    def unset(self, name):
        import slowflow_amd as sfa
        sfa.debug_set(name, None)


@pytest.fixture
def switches():
    s = _Switches()
    yield s
    import slowflow_amd as sfa
    for name in s.touched:
        sfa.debug_set(name, None)
