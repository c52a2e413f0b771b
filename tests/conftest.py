import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def built_lib():
    """The HIP library must exist (built by __graft_entry__.build()); build it if missing."""
    from detectorfreesfm_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _lib.lib()


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"))
    return load


@pytest.fixture(autouse=True)
def _workspace_guards():
    """With DFSFM_GUARD=1 (tools/gpu_harden.sh) every test ends with a check of the canary bands around the kernels' workspaces."""
    yield
    if os.environ.get("DFSFM_GUARD", "0") == "1":
        import torch
        if torch.cuda.is_available():
            from detectorfreesfm_amd import ops
            torch.cuda.synchronize()
            ops.check_workspace_guards()
