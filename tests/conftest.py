import os
import sys

import pytest

try:
    # torch bundles its own HIP runtime: it has to be loaded BEFORE libdrs_hip.so brings in the
    # system one, or torch.cuda sees no device in the operator-level GPU tests
    import torch  # noqa: F401
except ImportError:
    pass

os.environ.setdefault("DRS_DISPATCH_LOG", "1")     # engines keep the record drs_last_dispatch returns (off by default)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# DRS_TEST_LAB=1: the GPU suite on the LAB build of the library (libdrs_hip_lab.so: make -C deeprecsys_amd/csrc lab-lib) --
# the same kernels plus the options of the forms that lost their measurement (mlp_stream 0 / 1, mlp_early, launch_thread,
# zero_copy_inputs 0, ...), which the product library refuses; the tests run those entries only then (H.LAB).
if os.environ.get("DRS_TEST_LAB", "0") not in ("", "0"):
    from deeprecsys_amd import _native as _N
    _lab = os.path.join(ROOT, "deeprecsys_amd", "libdrs_hip_lab.so")
    if not os.path.exists(_lab):
        raise RuntimeError("DRS_TEST_LAB=1 needs %s (make -C deeprecsys_amd/csrc lab-lib)" % _lab)
    _N.LIB_PATH = _lab


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        from deeprecsys_amd import _native
        return _native.device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU must fail loudly, not silently skip: only skip
    # gpu tests when they were not explicitly selected.
    if "gpu" in (config.getoption("-m") or ""):
        return
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture
def cpu_abi(monkeypatch):
    """The C ABI of include/drs.h restated on the CPU oracle (oracle/drs_cpu_abi.cpp), bound in
    place of libdrs_hip.so FOR THIS TEST ONLY: the host code above the ABI (ctypes binding,
    model wrappers, engine request loop) runs unchanged, without a GPU.  Test infrastructure:
    nothing in deeprecsys_amd/ can reach this library on its own."""
    import ctypes
    import subprocess
    from deeprecsys_amd import _native
    path = os.path.join(ROOT, "oracle", "_build", "libdrs_cpu.so")
    if not os.path.exists(path):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    L = ctypes.CDLL(path)
    for name, res, args in _native.SYMBOLS:
        fn = getattr(L, name)          # AttributeError = the restatement misses an entry point
        fn.restype = res
        fn.argtypes = args
    monkeypatch.setattr(_native, "_lib", L)
    return L
