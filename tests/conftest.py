import os
import subprocess
import sys

import pytest

# The suite's bar is BIT equality with the oracle, which walks every BVH in the reference's order: contexts are created with
# HK_CTX_EXACT_TRAVERSAL (32) unless a test asks for the product default (cases.product_default_traversal()).  Through the
# environment, so that the rank processes the band tests spawn inherit it.
os.environ.setdefault("HIKARI_HIP_DEFAULT_CTX_FLAGS", "32")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Build the product library and the oracle once per session (no-op when up to date)."""
    import __graft_entry__ as g

    g.build(verbose=False)


def has_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def cornell():
    import bevy_hikari_amd as hk

    return hk.load_cornell()
