import os
import sys

import numpy as np
import pytest

# the process-global *_set_* hooks of libddsp_hip.so (kernel-form selection, fault injection) only work in processes that
# opted in BEFORE the library was loaded (include/ddsp_hip.h: ddsp_test_hooks_enabled)
os.environ.setdefault("DDSP_TEST_HOOKS", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def golden():
    return load_golden


@pytest.fixture(scope="session", autouse=True)
def _hooks_are_armed(request):
    """The 'form A == form B' tests switch kernel forms through the process-global *_set_* hooks, which refuse (DDSP_EPERM) unless
    DDSP_TEST_HOOKS=1 was set when the library was loaded; an unarmed run would compare the default kernel with itself and pass
    vacuously.  Checked once per session, on the GPU box only (loading the library needs the built .so, not a GPU)."""
    if "gpu" not in (request.config.getoption("-m") or "") or "not gpu" in (request.config.getoption("-m") or ""):
        return
    import ddsp_pytorch_amd as ddsp
    assert ddsp._lib.lib().ddsp_test_hooks_enabled() == 1, \
        "DDSP_TEST_HOOKS must be 1 before libddsp_hip.so is loaded (an exported DDSP_TEST_HOOKS=0 disables the kernel-form hooks)"
