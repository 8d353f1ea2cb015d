#!/usr/bin/env python3
"""bench.py on deeprecsys_amd/libdrs_hip_lab.so (make -C deeprecsys_amd/csrc lab-lib): the lab's options
(mlp_cu_mask, gather_priority, ...) for A/B lines; never a published number."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from deeprecsys_amd import _native as N
N.LIB_PATH = os.path.join(os.path.dirname(N.LIB_PATH), "libdrs_hip_lab.so")
import bench
if __name__ == "__main__":
    bench.main()
