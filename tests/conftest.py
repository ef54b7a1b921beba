import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session", autouse=True)
def _built_libraries():
    """The .so files are built in-tree by __graft_entry__.build(); build here only if they are missing
    and a compiler is present (never on the GPU box, which receives them prebuilt)."""
    lib = os.path.join(ROOT, "pandora_amd", "lib", "libpandora_hip.so")
    orc = os.path.join(ROOT, "oracle", "_build", "libpandora_oracle.so")
    if not os.path.exists(lib) or not os.path.exists(orc):
        sys.path.insert(0, ROOT)
        import __graft_entry__ as g
        g.build()
    yield


@pytest.fixture(scope="session")
def gpu_ctx():
    if not _have_gpu():
        pytest.skip("no GPU visible")
    import pandora_amd as pa
    ctx = pa.accel.Context(0)
    yield ctx
    ctx.close()
