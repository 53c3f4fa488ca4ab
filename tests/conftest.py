import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the CPU-side helper libraries (checker + synthetic archive generator) are built on demand;
    # the HIP product library is built by __graft_entry__.build() and travels in-tree.
    for d, target in (("oracle", "liboracle.so"), ("benchdata", "libzpkgen.so")):
        if not os.path.exists(os.path.join(ROOT, d, target)):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, d), target], stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session", autouse=True)
def _torch_hip_first(request):
    """On a GPU box, let torch bring up ITS HIP runtime before the codec library (linked against /opt/rocm) loads: the
    other order leaves torch without a device in the same process ("No HIP GPUs are available"), whatever the test
    selection or file order.  Device memory and streams come from torch in the device-resident tests."""
    if request.config.getoption("-m") and "not gpu" in request.config.getoption("-m"):
        return
    try:
        import torch
        if torch.cuda.device_count() > 0:
            torch.zeros(1, device="cuda:0")
    except Exception:
        pass
