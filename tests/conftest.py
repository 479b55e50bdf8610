import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """The built libraries are git-ignored: a fresh checkout has none.  Build them once (hipcc
    cross-compiles gfx950 without a GPU; gcc for the oracle) so that the suite does not depend on
    someone having called __graft_entry__.build() first."""
    import subprocess
    hip_lib = os.path.join(ROOT, "fiveeqscm_amd", "csrc", "libfiveeq_hip.so")
    if not os.path.exists(hip_lib):
        subprocess.run(["make", "-C", os.path.dirname(hip_lib)], check=True)
    ora_lib = os.path.join(ROOT, "oracle", "libfiveeq_oracle.so")
    if not os.path.exists(ora_lib):
        subprocess.run(["make", "-C", os.path.dirname(ora_lib)], check=True)


@pytest.fixture(scope="session")
def golden_hfc():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "hfc_conc_golden.json")) as fh:
        return json.load(fh)
