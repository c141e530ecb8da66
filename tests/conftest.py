import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the C-ABI library is a build product (git-ignored): build it once if a fresh checkout has none yet
    # (hipcc cross-compiles gfx950 without a GPU; ~15 s).  The product never falls back to anything else.
    so = os.path.join(ROOT, "tensorflow-yolo_amd", "libyolo_hip.so")
    if not os.path.exists(so):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def has_gpu():
    import torch
    return torch.cuda.is_available()
