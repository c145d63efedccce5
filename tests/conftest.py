import glob
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def pytest_collection_modifyitems(config, items):
    """`gpu` tests are skipped (not errored) where they cannot run: no libkrisp_hip.so in the tree,
    or no AMD GPU device node.  Nothing here initialises HIP."""
    lib = os.environ.get("KRISP_HIP_LIB") or os.path.join(ROOT, "krisp_amd", "libkrisp_hip.so")
    why = None
    if not os.path.exists(lib):
        why = f"{lib} is not built (python -m krisp_amd.build)"
    elif not (os.path.exists("/dev/kfd") or glob.glob("/dev/dri/renderD*") or os.environ.get("KRISP_GPU_TESTS") == "1"):
        why = "no GPU on this machine (no /dev/kfd, no /dev/dri/renderD*; KRISP_GPU_TESTS=1 overrides)"
    if why:
        skip = pytest.mark.skip(reason=why)
        for item in items:
            if "gpu" in item.keywords:
                item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
