import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
DATA = os.path.join(ROOT, "data")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle.pyoracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def ref():
    from oracle.pyoracle import Ref, ref_available, build
    build()
    if not ref_available():
        pytest.skip("oracle/_ref/libacg_ref.so not built (no /root/reference here)")
    return Ref()


@pytest.fixture(scope="session")
def matrices(oracle):
    return {
        "H": oracle.read_pcm(os.path.join(DATA, "H.txt")),
        "H05": oracle.read_pcm(os.path.join(DATA, "H05.txt")),
        "optimalH": oracle.read_pcm(os.path.join(DATA, "optimalH.txt")),
    }
