"""MI355X (gfx950) inference engine for the DeepRecSys hot path.

Host side mirrors the reference's Python surface (DeepRecSys.py / loadGenerator.py /
scheduler.py / accelInferenceEngine.py / models/*_Wrapper); the arithmetic runs in
hand-written HIP kernels behind the C ABI of include/drs.h (libdrs_hip.so, loaded
with ctypes in _native.py).  There is no CPU fallback: without the shared library
or without a GPU every compute entry point raises.
"""
__version__ = "0.1.0"
