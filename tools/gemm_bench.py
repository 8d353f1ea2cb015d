#!/usr/bin/env python3
"""Stand-alone timing of the wide-layer GEMM (drs_fc -> gemm_kernel) on the shapes the models
use, for the MFMA-utilisation figures in DESIGN.md.  Run under rocprofv3 --kernel-trace for
per-launch durations, or alone for torch-event timings:

    python tools/gemm_bench.py [--rows 2048]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from deeprecsys_amd import _native as N

SHAPES = [("RM3 bottom L1", 2560, 1024), ("RM3 bottom L2", 1024, 256), ("W&D top L1", 1376, 1024),
          ("W&D top L2", 1024, 512), ("W&D top L3", 512, 256), ("RM1 top L1", 576, 256)]
PEAK = 157.3e12


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=2048)
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--tile", type=int, default=0, help="mlp_gemm_tile (0 auto | 22 | 12 | 21 | 11 | 214 | 322 | 321 | 312 | 311)")
    ap.add_argument("--gemm32", type=int, default=-1, help="mlp_gemm32 (-1: the engine's default | 0 gemm_kernel only | 1 gemm32_kernel where its 128 x 128 tiles number >= mlp_gemm32_blocks)")
    ap.add_argument("--shapes", default="", help="comma-separated KxN list instead of the models' shapes")
    o = ap.parse_args()
    eng = N.Engine(N.MODEL_DLRM, [16, 16], 8, [4, 8], [24, 4, 1], N.INTERACT_CAT, sigmoid_top=2,
                   max_batch=4, max_lookups=2, num_staged_batches=1, num_slots=1)
    eng.set_option("mlp_gemm_tile", o.tile)
    if o.gemm32 >= 0:
        eng.set_option("mlp_gemm32", o.gemm32)
    shapes = SHAPES if not o.shapes else [("%s" % kn, int(kn.split("x")[0]), int(kn.split("x")[1])) for kn in o.shapes.split(",")]
    dev = torch.device("cuda", 0)
    M = o.rows
    print("%-16s %6s %6s %6s %10s %10s %8s" % ("layer", "M", "K", "N", "us/launch", "TFLOP/s", "of peak"))
    for name, K, Nn in shapes:
        x = torch.rand(M, K, device=dev)
        W = torch.rand(Nn, K, device=dev) - 0.5
        b = torch.rand(Nn, device=dev)
        y = torch.empty(M, Nn, device=dev)
        for _ in range(10):
            eng.fc(x.data_ptr(), M, K, W.data_ptr(), b.data_ptr(), Nn, N.ACT_RELU, y.data_ptr())
        torch.cuda.synchronize()
        # drs_fc launches on the engine's own stream and waits for it: time the whole call
        import time
        t0 = time.perf_counter()
        for _ in range(o.iters):
            eng.fc(x.data_ptr(), M, K, W.data_ptr(), b.data_ptr(), Nn, N.ACT_RELU, y.data_ptr())
        us = (time.perf_counter() - t0) / o.iters * 1e6
        fl = 2.0 * M * K * Nn
        print("%-16s %6d %6d %6d %10.2f %10.1f %8.3f  (host-timed, includes launch + sync)" %
              (name, M, K, Nn, us, fl / us / 1e6, fl / (us * 1e-6) / PEAK))
    eng.close()


if __name__ == "__main__":
    main()
