"""pytest configuration: the `gpu` marker and shared fixtures.

`-m "not gpu"` (CPU container): oracle vs golden vectors, oracle vs the compiled reference when
oracle/_ref exists, host logic (loader, scenes, file formats), ABI symbol checks.
`-m gpu` (MI355X box): parity of the HIP path against the oracle and the golden vectors, through
the C ABI.
"""
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    import __graft_entry__ as entry
    lib = entry.PKG_DIR / "libptw_hip.so"
    if not lib.exists():
        entry.build()
    return entry.load_package()


@pytest.fixture(scope="session")
def ob(pkg):
    import oracle_binding
    return oracle_binding


@pytest.fixture(scope="session")
def golden_dir():
    return ROOT / "tests" / "golden"
