#!/usr/bin/env python3
"""Where does the host time of one query go?  (GPU box only.)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench


def main():
    print("affinity cpus:", len(os.sched_getaffinity(0)), "cpu_count:", os.cpu_count())
    for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        if os.path.exists(p):
            print(p, open(p).read().strip())
    sys.argv = ["bench.py", "--num_batches", "8", "--slots", "4"]
    opt = bench.parse()
    args, net, data = bench.make_model(opt, 0)
    eng = net.engine
    for kv in os.environ.get("DRS_PROBE_SET", "").split():
        k, v = kv.split("=")
        eng.set_option(k, int(v))
    bs, nb = opt.batch, opt.num_batches
    for slots in (1, 2, 4):
        bench.run_queries(eng, 300, bs, nb, slots)
        t_sub, t_wait = 0.0, 0.0
        n = 2000
        busy = [False] * slots
        t0 = time.perf_counter()
        for i in range(n):
            s = i % slots
            if busy[s]:
                a = time.perf_counter()
                eng.wait(s)
                t_wait += time.perf_counter() - a
            a = time.perf_counter()
            eng.forward_async(s, i % nb, bs)
            t_sub += time.perf_counter() - a
            busy[s] = True
        eng.sync()
        el = time.perf_counter() - t0
        print("slots=%d: %.1f us/query total; submit %.1f us, wait %.1f us" %
              (slots, el / n * 1e6, t_sub / n * 1e6, t_wait / n * 1e6))
    # per-call host inputs (the run_queues signature): where does the time go?
    lX, lS_l, lS_i = data
    L = bench.WORKLOADS[opt.workload]["L"]
    ids = np.stack([np.asarray(t[:bs * L], dtype=np.int64) for t in lS_i[0]])
    lens = np.stack([np.asarray(t[:bs], dtype=np.int32) for t in lS_l[0]])
    x = np.ascontiguousarray(lX[0][:bs])
    for ht in (3, 7):
        eng.set_option("host_threads", ht)
        for zc in (1, 2):
            eng.set_option("zero_copy_inputs", zc)
            for slots in (1, 3, 4):
                t_sub = t_wait = 0.0
                n = 2000
                busy = [False] * slots
                t0 = time.perf_counter()
                for i in range(n):
                    s = i % slots
                    if busy[s]:
                        a = time.perf_counter()
                        eng.wait(s)
                        t_wait += time.perf_counter() - a
                    a = time.perf_counter()
                    eng.forward_inputs_async(x, ids, lens, bs, slot=s)
                    t_sub += time.perf_counter() - a
                    busy[s] = True
                eng.sync()
                el = time.perf_counter() - t0
                print("host inputs, host_threads=%d zero_copy_inputs=%d slots=%d: %.1f us/query; submit %.1f us, wait %.1f us"
                      % (ht, zc, slots, el / n * 1e6, t_sub / n * 1e6, t_wait / n * 1e6))
    eng.set_option("zero_copy_inputs", 1)
    eng.set_option("host_threads", -1)
    # sync forward latency
    t0 = time.perf_counter()
    for i in range(1000):
        eng.forward(i % nb, bs)
    print("sync forward: %.1f us" % ((time.perf_counter() - t0) / 1000 * 1e6))
    # oracle threads scaling
    from oracle import oracle as orc
    from tests import helpers as H
    w = bench.WORKLOADS[opt.workload]
    lX, lS_l, lS_i = data
    lo, hi = -float(np.sqrt(1 / w["rows"])), float(np.sqrt(1 / w["rows"]))
    net.emb_w = [orc.fill_table_uniform(w["rows"], w["D"], t, lo, hi, opt.seed, nthreads=16) for t in range(w["T"])]
    om = H.oracle_model(net)
    for nt in (1, 4, 16, 64):
        om.forward(lX[0], lS_i[0], lS_l[0], bs=bs, nthreads=nt)
        t0 = time.perf_counter()
        for i in range(5):
            om.forward(lX[i % nb], lS_i[i % nb], lS_l[i % nb], bs=bs, nthreads=nt)
        print("oracle threads=%d: %.2f ms/query" % (nt, (time.perf_counter() - t0) / 5 * 1e3))
    eng.close()


if __name__ == "__main__":
    main()
