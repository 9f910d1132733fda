import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

REFERENCE = Path("/root/reference")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The product has no CPU fallback: make sure the HIP library (and the oracle, the checker) exist.
    # hipcc cross-compiles gfx950 without a GPU; this is a no-op when everything is up to date.
    from powdr_amd import build as _build

    _build.build()
    from oracle import apc_model as _om

    _om.build_c_oracle()


@pytest.fixture(scope="session")
def reference_dir():
    if not REFERENCE.exists():
        pytest.skip("/root/reference is not mounted on this machine")
    return REFERENCE
