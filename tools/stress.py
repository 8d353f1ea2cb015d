#!/usr/bin/env python3
"""Race hunt for the pipelined engine (GPU box): random launch sets on random slots for a
fixed time, every result compared BIT FOR BIT with the same query served alone on an idle
engine.  Any stale read (gather -> MLP across streams, slot reuse, completion hand-off,
coalescing offsets) shows up as a mismatch.

    python tools/stress.py --seconds 30 [--max_set k] [--workload rmc1] [--set key=value ...]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench
if os.environ.get("DRS_TEST_LAB", "0") not in ("", "0"):      # the lab build (its options: mlp_early, mlp_stream 0 / 1, ...)
    from deeprecsys_amd import _native as _N
    _N.LIB_PATH = os.path.join(os.path.dirname(_N.LIB_PATH), "libdrs_hip_lab.so")


def main():
    argv = sys.argv[1:]
    seconds = 20.0
    if "--seconds" in argv:
        i = argv.index("--seconds")
        seconds = float(argv[i + 1])
        del argv[i:i + 2]
    max_set = 16                          # --max_set k: launch sets of 1 .. k queries (small k: the small-set launch forms)
    if "--max_set" in argv:
        i = argv.index("--max_set")
        max_set = max(1, min(16, int(argv[i + 1])))
        del argv[i:i + 2]
    sys.argv = ["bench.py", "--num_batches", "8", "--slots", "4"] + argv
    opt = bench.parse()
    args, net, data = bench.make_model(opt, 0)
    eng = net.engine
    for kv in opt.set:
        k, v = kv.split("=")
        eng.set_option(k, int(v))
    nb, B, slots = opt.num_batches, opt.batch, opt.slots
    rng = np.random.RandomState(1)
    sizes = sorted(s for s in {B, 1, 2, 17, 64, 65, 165, B // 2, B - 1} if 1 <= s <= B)
    # ground truth: each (batch, size) alone, single stream, nothing else in flight
    eng.set_option("shared_stream", 1)
    truth = {(b, s): eng.forward(b, s).copy() for b in range(nb) for s in sizes}
    mode = dict(kv.split("=") for kv in opt.set).get("shared_stream", "2")
    eng.set_option("shared_stream", int(mode))
    inflight = [None] * slots
    n_sets = n_q = 0
    t0 = time.time()

    def check(slot):
        jobs = inflight[slot]
        out = eng.wait(slot, sum(s for _, s in jobs))
        o = 0
        for b, s in jobs:
            if not np.array_equal(out[o:o + s], truth[(b, s)]):
                bad = np.argwhere(out[o:o + s] != truth[(b, s)])
                raise SystemExit("MISMATCH slot %d job (%d,%d) first bad row %s after %d sets" % (slot, b, s, bad[0], n_sets))
            o += s
        inflight[slot] = None

    while time.time() - t0 < seconds:
        slot = int(rng.randint(slots))
        if inflight[slot] is not None:
            check(slot)
        k = int(rng.randint(1, max_set + 1))     # 1 .. max_set (default DRS_MAX_COALESCE) queries per launch set
        jobs = [(int(rng.randint(nb)), int(sizes[rng.randint(len(sizes))])) for _ in range(k)]
        eng.forward_multi_async(slot, [b for b, _ in jobs], [s for _, s in jobs])
        inflight[slot] = jobs
        n_sets += 1
        n_q += k
        if rng.rand() < 0.05:                      # now and then drain everything
            for s_ in range(slots):
                if inflight[s_] is not None:
                    check(s_)
    for s_ in range(slots):
        if inflight[s_] is not None:
            check(s_)
    print("stress OK: %d launch sets, %d queries, %.1f s, all bit-identical to the queries served alone"
          % (n_sets, n_q, time.time() - t0))
    eng.close()


if __name__ == "__main__":
    main()
