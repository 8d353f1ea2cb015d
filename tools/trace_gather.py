#!/usr/bin/env python3
"""The gather on locality-aware index traces (GPU box).

Index streams for every table are synthesised with the reference's method
(deeprecsys_amd/data_generator/trace_generator.py <- data_generator/trace_generator.py:71-97)
from a stack-distance profile, cut into bags of L consecutive references, staged as the
engine's resident input sets, and the benchmark's timed loop is run on them:

    python tools/trace_gather.py --profile uniform|shipped|hot [--workload rmc1] [--steps 4]

  uniform  bench.py's default generator (sorted unique uniform rows per bag: ~1 % reuse per set)
  shipped  the profile the reference ships (tests/golden/traces.npz: 99.9 % of the references
           touch a new line, the rest re-touch one of the last ~10 k lines)
  hot      a reuse-heavy profile: 55 % of the references re-touch one of the last 100 lines
           (not from the reference: it shows what L2 / Infinity-Cache reuse is worth)
Prints one JSON line: gather GB/s (algorithmic bytes / device-clock duration, like bench.py's
roofline leg), queries/s, distinct rows per launch.  Run under `rocprofv3 --pmc TCC_HIT_sum
TCC_MISS_sum` for the L2 hit rate (tools/capture_profiles.sh does).
"""
import argparse
import json
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench
from deeprecsys_amd import _native as N
from deeprecsys_amd.data_generator import trace_generator as TG

HOT = ([0, 1, 2, 3, 5, 8, 13, 40, 100], [0.45, 0.55, 0.63, 0.7, 0.78, 0.85, 0.9, 0.96, 1.0])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--profile", default="shipped", choices=("uniform", "shipped", "hot"))
    ap.add_argument("--workload", default="rmc1")
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--num_batches", type=int, default=16)
    ap.add_argument("--seed", type=int, default=123)
    o = ap.parse_args()
    sys.argv = ["bench.py", "--workload", o.workload, "--num_batches", str(o.num_batches)]
    opt = bench.parse()
    args, net, (lX, lS_l, lS_i) = bench.make_model(opt, 0)
    eng = net.engine
    w = bench.WORKLOADS[opt.workload]
    T, L, B, nb = w["T"], w["L"], opt.batch, opt.num_batches
    rows = w["rows"] if isinstance(w["rows"], list) else [w["rows"]] * T
    gen_s = 0.0
    if o.profile != "uniform":
        if o.profile == "shipped":
            z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "traces.npz"))
            lsd, csd = z["shipped/list_sd"].tolist(), z["shipped/cumm_sd"].tolist()
        else:
            lsd, csd = HOT
        t0 = time.perf_counter()
        random.seed(o.seed)
        np.random.seed(o.seed)
        per_table = [TG.bags_from_trace(TG.trace_generate_lru(rows[t], lsd, csd, nb * B * L), nb * B, L) for t in range(T)]
        gen_s = time.perf_counter() - t0
        lens = [np.full(B, L, dtype=np.int32) for _ in range(T)]
        for b in range(nb):
            idx = [per_table[t][b * B * L:(b + 1) * B * L] for t in range(T)]
            eng.stage_batch(b, None if w.get("kind") in ("ncf", "din", "dien") else lX[b], idx, lens)
            lS_i[b] = idx
    distinct = float(np.mean([np.unique(np.concatenate([lS_i[b][t] for b in range(min(8, nb))])).size
                              for t in range(T)])) / (min(8, nb) * B * L)
    co, slots, q = opt.coalesce or 8, opt.slots, 8192
    bench.run_queries(eng, q, B, nb, slots, coalesce=co)
    eng.reset_kernel_time()
    eng.set_profiling(1)
    el = bench.run_queries(eng, o.steps * q, B, nb, slots, coalesce=co)
    eng.set_profiling(0)
    ms, n = eng.kernel_time(N.KERNEL_SLS_CLOCK)
    by = eng.kernel_bytes(N.KERNEL_SLS_CLOCK)
    print(json.dumps({"profile": o.profile, "workload": opt.workload, "queries_per_s": round(o.steps * q / el, 1),
                      "gather_GBps": round(by / (ms * 1e-3) / 1e9, 1), "gather_frac_of_8TBps": round(by / (ms * 1e-3) / 8e12, 4),
                      "avg_launch_us": round(ms / n * 1e3, 2), "launches": n,
                      "distinct_rows_share_over_8_sets": round(distinct, 4),
                      "trace_generation_s": round(gen_s, 1), "input_sets": nb}), flush=True)
    eng.close()


if __name__ == "__main__":
    main()
