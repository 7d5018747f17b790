import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the in-tree libraries are built by `make` / __graft_entry__.build(); build what is missing
    lib = os.path.join(ROOT, "generalized_rbda_amd", "libgrbda_hip.so")
    ora = os.path.join(ROOT, "oracle", "_build", "libgrbda_oracle.so")
    if not os.path.exists(ora):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True)
    if not os.path.exists(lib):
        subprocess.run(["make", "-C", ROOT, os.path.relpath(lib, ROOT)], check=True)


@pytest.fixture(scope="session")
def gpu():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    return torch.device("cuda:0")
