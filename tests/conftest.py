import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_sessionstart(session):
    """The shared library is a build artefact (git-ignored): build it once if this checkout does not have it yet.
    (Test infrastructure only - the product itself never builds or falls back at run time: neuma_amd._lib.lib() raises.)"""
    import shutil
    import subprocess
    lib = ROOT / "neuma_amd" / "lib" / "libneuma_hip.so"
    if lib.exists() or os.environ.get("NEUMA_HIP_LIB"):
        return
    if shutil.which("hipcc") is None and not Path("/opt/rocm/bin/hipcc").exists():
        return          # the tests that need the library will fail loudly
    subprocess.run(["make", "-C", str(ROOT / "neuma_amd" / "csrc"), "-j", str(min(8, os.cpu_count() or 2))], check=False)
