import os
import sys

import pytest

# The oracle and the reference build are OpenMP code: on a 256-thread host libgomp's default (every hardware thread) makes each of
# their calls ~100x slower than 16 threads (measured on the GPU box: 8 q/s against 720 q/s for a 1 M-code linear scan).
os.environ.setdefault("OMP_NUM_THREADS", str(min(16, os.cpu_count() or 1)))

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
