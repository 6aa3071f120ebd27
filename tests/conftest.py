import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """A clean checkout has no built artefacts (they are git-ignored): build the HIP library, the zpaqv CLI and
    the oracle once, the same way the driver's build() does (hipcc cross-compiles without a GPU)."""
    need = [os.path.join(ROOT, "zpaq-v_amd", "lib", "libzpaq_hip.so"), os.path.join(ROOT, "zpaq-v_amd", "bin", "zpaqv"),
            os.path.join(ROOT, "oracle", "libzpaq_oracle.so")]
    if not all(os.path.exists(p) for p in need):
        import __graft_entry__ as ge
        ge.build()


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this environment")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def zpq():
    """The product package (zpaq-v_amd/) loaded as module zpaq_v_amd."""
    import __graft_entry__ as ge
    return ge.load()


@pytest.fixture(scope="session")
def gpu_ctx(zpq):
    ctx = zpq.Context(0)
    yield ctx
    ctx.close()
