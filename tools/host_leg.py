#!/usr/bin/env python3
"""Only the per-call host-input loop (profiling target):
    python tools/host_leg.py [--mode 1] [--slots 6] [--n 4000] [--set key=value ...]
prints queries/s and the split between submit (drs_run_queues_async) and wait per query.
--per_set N (> 0): N queries per call through drs_run_queues_multi_async instead."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench

ap = argparse.ArgumentParser()
ap.add_argument("--mode", type=int, default=1)
ap.add_argument("--slots", type=int, default=6)
ap.add_argument("--n", type=int, default=4000)
ap.add_argument("--per_set", type=int, default=0)
ap.add_argument("--set", action="append", default=[])
o = ap.parse_args()
sys.argv = ["bench.py", "--num_batches", "4", "--slots", str(o.slots)]
opt = bench.parse()
args, net, (lX, lS_l, lS_i) = bench.make_model(opt, 0)
eng = net.engine
eng.set_option("zero_copy_inputs", o.mode)
for kv in o.set:
    k, v = kv.split("=")
    eng.set_option(k, int(v))
bs, L = opt.batch, bench.WORKLOADS[opt.workload]["L"]
sets = [(np.stack([np.asarray(t[:bs * L], dtype=np.int64) for t in lS_i[b]]),
         np.stack([np.asarray(t[:bs], dtype=np.int32) for t in lS_l[b]]), np.ascontiguousarray(lX[b][:bs])) for b in range(4)]
if o.per_set > 0:
    for rep in range(2):
        busy = [False] * o.slots
        t_sub = t_wait = 0.0
        n_sets = max(8, o.n // o.per_set)
        t0 = time.perf_counter()
        for i in range(n_sets):
            s = i % o.slots
            if busy[s]:
                ta = time.perf_counter()
                eng.wait(s)
                t_wait += time.perf_counter() - ta
            qs = [(sets[(i + k) % 4][2], sets[(i + k) % 4][0], sets[(i + k) % 4][1], bs) for k in range(o.per_set)]
            ta = time.perf_counter()
            eng.run_queues_multi_async(qs, slot=s)
            t_sub += time.perf_counter() - ta
            busy[s] = True
        eng.sync()
        el = time.perf_counter() - t0
    nq = n_sets * o.per_set
    print("sets of %d, %d in flight %s: %.1f us/query = %.0f queries/s (submit %.1f us, wait %.1f us per SET)"
          % (o.per_set, o.slots, " ".join(o.set), el / nq * 1e6, nq / el, t_sub / n_sets * 1e6, t_wait / n_sets * 1e6))
    eng.close()
    sys.exit(0)
for rep in range(2):
    busy = [False] * o.slots
    t_sub = t_wait = 0.0
    t0 = time.perf_counter()
    for i in range(o.n):
        s = i % o.slots
        if busy[s]:
            ta = time.perf_counter()
            eng.wait(s)
            t_wait += time.perf_counter() - ta
        ids, lens, x = sets[i % 4]
        ta = time.perf_counter()
        eng.forward_inputs_async(x, ids, lens, bs, slot=s)
        t_sub += time.perf_counter() - ta
        busy[s] = True
    eng.sync()
    el = time.perf_counter() - t0
print("mode %d slots %d %s: %.1f us/query = %.0f queries/s (submit %.1f us, wait %.1f us per query)"
      % (o.mode, o.slots, " ".join(o.set), el / o.n * 1e6, o.n / el, t_sub / o.n * 1e6, t_wait / o.n * 1e6))
eng.close()
