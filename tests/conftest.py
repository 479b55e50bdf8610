import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """The built libraries are git-ignored: a fresh checkout has none, and a checkout with edited sources has stale
    ones.  `make` is incremental (a no-op when up to date), so run it every session: hipcc cross-compiles gfx950
    without a GPU; gcc builds the oracle.  Then check that the loaded library is the ABI the binding expects."""
    import subprocess
    for sub in (("fiveeqscm_amd", "csrc"), ("oracle",)):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, *sub)], check=True)
    from fiveeqscm_amd import _capi
    assert _capi.load().fiveeq_abi_version() == _capi.ABI_VERSION


@pytest.fixture(scope="session")
def golden_hfc():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "hfc_conc_golden.json")) as fh:
        return json.load(fh)
