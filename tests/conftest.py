import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def reference():
    """The real reference build (oracle/_ref); tests that need it skip when it is absent/unloadable."""
    from oracle import oracle as O
    mod, arch, flav = O.load_reference()
    if mod is None:
        pytest.skip("oracle/_ref not built (needs /root/reference; run `make -C oracle ref`)")
    return mod, arch, flav
