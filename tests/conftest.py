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
