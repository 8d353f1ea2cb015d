#!/usr/bin/env python3
"""Emit the reference's accelerator characterisation table from MEASURED MI355X runs.

The reference's simulator engine reads `<root>/<hardware>/raw_data/results_<model>.txt`:
six "***" lines per batch size 4**0 .. 4**5 (accelerator/predict_execution.py:10-29,49-62;
produced there by models/*.py --use_accel on a GTX 1080 Ti, generate_data.py:20).  This
tool produces the same files from the HIP engine so `--accel_backend sim` (and the
reference's own accelInferenceEngine) can replay MI355X latencies:

    python tools/characterize.py --model rm1 --out accelerator_mi355x/

"data loading" = host->device input hand-over of the non-staged path (drs_forward_inputs:
int64->int32 narrowing, pinned staging, H2D), "computation" = the forward on resident
inputs, "execution" = their sum -- the split the reference prints
(models/dlrm_s_caffe2.py:645-661).
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench
from deeprecsys_amd import latency_table

MODEL_TO_WORKLOAD = {"rm1": "rmc1_ref", "rm2": "rmc2_ref", "rm3": "rmc3_ref", "rm1_baseline": "rmc1",
                     "wnd": "wnd", "ncf": "ncf", "mtwnd": "mtwnd", "din": "din", "dien": "dien"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="rm1", choices=sorted(MODEL_TO_WORKLOAD))
    ap.add_argument("--out", default="gpurun_out/accelerator_mi355x/")
    ap.add_argument("--iters", type=int, default=200)
    o = ap.parse_args()
    sys.argv = ["bench.py", "--workload", MODEL_TO_WORKLOAD[o.model], "--batch", "1024", "--num_batches", "4",
                "--slots", "1"]
    opt = bench.parse()
    args, net, (lX, lS_l, lS_i) = bench.make_model(opt, 0)
    eng = net.engine
    L = bench.WORKLOADS[opt.workload]["L"]
    rows = []
    print("model %s (%s): batch, load ms/iter, compute ms/iter, total ms/iter" % (o.model, opt.workload))
    for p in range(6):
        bs = 4 ** p
        ids = [i[:bs * L] for i in lS_i[0]]
        lens = [l[:bs] for l in lS_l[0]]
        x = None if bench.WORKLOADS[opt.workload].get("kind") in bench.NO_DENSE else lX[0][:bs]
        for _ in range(20):
            eng.forward(0, bs)
            eng.forward_inputs(x, ids, lens, bs)
        t0 = time.perf_counter()
        for i in range(o.iters):
            eng.forward(i % opt.num_batches, bs)
        comp = (time.perf_counter() - t0) / o.iters * 1e3
        t0 = time.perf_counter()
        for i in range(o.iters):
            eng.forward_inputs(x, ids, lens, bs)
        total = (time.perf_counter() - t0) / o.iters * 1e3
        load = max(total - comp, 0.0)
        rows.append((load * o.iters, load, comp * o.iters, comp, total * o.iters, total))
        print("%5d  %.4f  %.4f  %.4f" % (bs, load, comp, total))
    d = os.path.join(o.out, "amd_mi355x", "raw_data")
    os.makedirs(d, exist_ok=True)
    path = os.path.join(d, "results_%s.txt" % o.model.replace("_baseline", ""))
    latency_table.write_results(path, rows)
    back = latency_table.parse_results(path)
    assert len(back) == 6
    print("wrote", path)
    eng.close()


if __name__ == "__main__":
    main()
