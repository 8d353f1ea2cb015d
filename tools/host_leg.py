#!/usr/bin/env python3
"""Only the per-call host-input loop (profiling target): python tools/host_leg.py [mode] [slots] [n]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench

mode = int(sys.argv[1]) if len(sys.argv) > 1 else 3
slots = int(sys.argv[2]) if len(sys.argv) > 2 else 3
n = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
sys.argv = ["bench.py", "--num_batches", "4", "--slots", str(slots)]
opt = bench.parse()
args, net, (lX, lS_l, lS_i) = bench.make_model(opt, 0)
eng = net.engine
eng.set_option("zero_copy_inputs", mode)
bs, L = opt.batch, bench.WORKLOADS[opt.workload]["L"]
sets = [(np.stack([np.asarray(t[:bs * L], dtype=np.int64) for t in lS_i[b]]),
         np.stack([np.asarray(t[:bs], dtype=np.int32) for t in lS_l[b]]), np.ascontiguousarray(lX[b][:bs])) for b in range(4)]
for rep in range(2):
    busy = [False] * slots
    t0 = time.perf_counter()
    for i in range(n):
        s = i % slots
        if busy[s]:
            eng.wait(s)
        ids, lens, x = sets[i % 4]
        eng.forward_inputs_async(x, ids, lens, bs, slot=s)
        busy[s] = True
    eng.sync()
    el = time.perf_counter() - t0
print("mode %d slots %d: %.1f us/query = %.0f queries/s" % (mode, slots, el / n * 1e6, n / el))
eng.close()
