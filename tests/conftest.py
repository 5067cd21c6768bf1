import os
import sys

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")   # before the HIP runtime initialises: see facialmmt_amd/__init__.py

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by `pytest -m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    from tests.golden_util import Golden
    return Golden()


# development hook: run the suite against another build of the library (same-call A/B of a tagged build, facialmmt_amd/build.py)
if os.environ.get("PROBE_LIB"):
    from facialmmt_amd import _lib as _fmmt_lib
    _fmmt_lib.LIB_PATH = os.environ["PROBE_LIB"]
