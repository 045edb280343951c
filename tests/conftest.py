import os
import sys

import pytest
# torch first: its wheel bundles its own ROCm and OpenMP runtimes; the oracle (libgomp) and the HIP library
# are dlopen'ed later and then share them instead of mapping second copies (see mcncrossmodalemotions_amd/_lib.py)
import torch  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    from mcncrossmodalemotions_amd import _lib
    _lib.load()  # fail loudly if the HIP extension is missing
    return torch.device("cuda:0")
