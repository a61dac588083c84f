import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure; built with gcc on first use)."""
    sys.path.insert(0, ROOT)
    from oracle import oracle as orc
    orc.lib()
    return orc


@pytest.fixture(scope="session")
def ctx():
    """A context on GPU 0 through the C ABI.  GPU tests must not silently pass without the HIP library."""
    from illuminant_amd import native
    assert native.device_count() > 0, "no HIP device visible: -m gpu tests need the MI355X"
    c = native.Context(0)
    yield c
    c.close()


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """What the float criterion was asked to forgive in this session (tests/util.py CENSUS): printed, and written where ILM_TOLERANCE_CENSUS
    names a file (the GPU box: gpurun_out/..., copied to profiles/ and committed)."""
    from tests import util
    if not util.CENSUS:
        return
    report = util.census_report()
    path = os.environ.get("ILM_TOLERANCE_CENSUS")
    if path:
        with open(path, "w") as f:
            f.write(report + "\n")
    terminalreporter.write_line("")
    for line in report.splitlines()[-1:]:
        terminalreporter.write_line(line)
