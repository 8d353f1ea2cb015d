"""TEST INFRASTRUCTURE: run a repo script (bench.py) with the C ABI of include/drs.h bound to its CPU
restatement (oracle/_build/libdrs_cpu.so) instead of libdrs_hip.so, so the multi-process rank entry
can be exercised on a box without a GPU:

    python tests/cpu_abi_entry.py bench.py --gpus 2 --collective gloo ...

The product binding (deeprecsys_amd/_native.py) refuses any library whose drs_backend() is not
"hip:*" and honours no environment override; the swap below lives in tests/ and nowhere else.
bench.py's ranks are re-launched through this file as well (bench.SPAWN_PREFIX)."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def bind_cpu_abi():
    from deeprecsys_amd import _native
    path = os.path.join(ROOT, "oracle", "_build", "libdrs_cpu.so")
    if not os.path.exists(path):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    L = ctypes.CDLL(path)
    for name, res, args in _native.SYMBOLS:
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    assert L.drs_backend() == b"cpu:oracle"
    _native._lib = L
    return L


if __name__ == "__main__":
    assert sys.argv[1] == "bench.py", "only bench.py's rank entry is driven this way"
    bind_cpu_abi()
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
    import bench
    bench.SPAWN_PREFIX = [os.path.abspath(__file__), "bench.py"]   # ranks come back through this file
    sys.exit(bench.main())
