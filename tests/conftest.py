import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built_checkers():
    """The oracle libraries are test infrastructure: build them if a run starts from a clean tree."""
    from oracle import oracle as O
    if not all(O.have_port(c) for c in O.COMBOS):
        O.build(ref=True)
    yield


@pytest.fixture(scope="session")
def hip_library():
    """libsdft_hip.so -- the product.  Built in-tree by hipcc when stale (cross-compiles without a GPU)."""
    from sdft_amd import build
    return build.build()


def have_gpu() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
