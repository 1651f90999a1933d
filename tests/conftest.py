"""pytest configuration: marker registration and shared fixtures.

`-m "not gpu"` runs here (no GPU): oracle vs golden vectors, host logic, C-ABI symbol
checks, gloo world_size-2 tests.  `-m gpu` runs on an MI355X box and calls the HIP
path through the C ABI; /root/reference does not exist there.
"""
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


# the 8-bit-digit tests must reach the 8-bit kernels at test sizes (the engine keeps them for >= 2^19 keys by default)
os.environ.setdefault("RSX_RADIX8_MIN_KEYS", "4096")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from _oracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def ref_oracle():
    from _oracle import RefOracle
    if not RefOracle.available():
        pytest.skip("oracle/_ref not built (needs /root/reference at build time)")
    return RefOracle()


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(HERE, "golden", "oracle_golden.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def rsx():
    """The product binding (ctypes over libradixsort_hip.so)."""
    from _loader import load_package
    return load_package()
