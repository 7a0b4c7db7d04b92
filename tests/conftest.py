import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_blobs():
    return [open(os.path.join(GOLDEN, "blobs", f"blob_{i}.bin"), "rb").read() for i in range(10)]


@pytest.fixture(scope="session")
def golden_vectors():
    return json.load(open(os.path.join(GOLDEN, "vectors.json")))["functions"]


@pytest.fixture(scope="session")
def setup_bytes():
    g1 = open(os.path.join(GOLDEN, "trusted_setup_g1.bin"), "rb").read()
    g2 = open(os.path.join(GOLDEN, "trusted_setup_g2.bin"), "rb").read()
    return g1, g2


@pytest.fixture(scope="session")
def oracle():
    from oracle.oracle import Oracle, build
    build()
    return Oracle()


@pytest.fixture(scope="session")
def oracle_settings(oracle, setup_bytes):
    s = oracle.load_trusted_setup(*setup_bytes)
    yield s
    oracle.free_trusted_setup(s)
