#!/usr/bin/env python3
"""Per-workgroup timeline of the gather launch (GPU box): when do workgroups start,
how long does each run?  Separates 'every wave is slow' from 'a few stragglers'."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench


def main():
    argv = sys.argv[1:]
    sys.argv = ["bench.py", "--num_batches", "8"] + argv
    opt = bench.parse()
    args, net, data = bench.make_model(opt, 0)
    eng = net.engine
    for kv in opt.set:
        k, v = kv.split("=")
        eng.set_option(k, int(v))
    bench.run_queries(eng, 200, opt.batch, opt.num_batches, 1)
    eng.set_profiling(1)
    spans = []
    for i in range(20):
        eng.forward(i % opt.num_batches, opt.batch)
        st = eng.gather_stamps(0).astype(np.int64)
        t0 = st[:, 0].min()
        start = (st[:, 0] - t0) / 100.0          # us (100 MHz clock)
        end = (st[:, 1] - t0) / 100.0
        dur = end - start
        spans.append(end.max())
        if i >= 17:
            q = lambda a: " ".join("%.2f" % np.percentile(a, p) for p in (0, 10, 50, 90, 100))
            print("launch %d: blocks=%d span=%.2f us | start[p0,p10,p50,p90,p100]= %s | dur= %s | end= %s"
                  % (i, len(st), end.max(), q(start), q(dur), q(end)))
    print("mean span %.2f us" % np.mean(spans))
    eng.set_profiling(0)
    eng.close()


if __name__ == "__main__":
    main()
