import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


# The placed allocator (caar_arrays_alloc) samples its physical chunks from a temporary pool; re-creating a 128 GiB pool
# costs ~4 s per allocation once the process has released memory before (the driver wipes it).  The tests exercise the
# same code path with a small pool.
os.environ.setdefault("CAAR_PLACEMENT_POOL_GIB", "16")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle as po
    return po.Oracle()


@pytest.fixture(scope="session", autouse=True)
def _built_artifacts():
    """A clean checkout has no binaries (they are git-ignored): build the HIP library, the
    C++ host side and the C oracle once per session.  hipcc cross-compiles without a GPU."""
    from tinman_sandbox_amd import build as b
    from oracle import pyoracle as po
    if not os.path.exists(b.LIB) or not os.path.exists(os.path.join(ROOT, "tinman_sandbox_amd", "host", "caar_driver")):
        b.build_all()
    if not os.path.exists(os.path.join(ROOT, "oracle", "libcaar_oracle.so")):
        po.build(ref=False)
    yield
