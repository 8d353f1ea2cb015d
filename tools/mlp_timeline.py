#!/usr/bin/env python3
"""Phase timeline of the MLP kernels (GPU box; needs `make -C deeprecsys_amd/csrc timeline`
; binds deeprecsys_amd/libdrs_hip_tl.so, the same HIP library built with -DDRS_TIMELINE).  Prints, for workgroup 0 / wave 0, the
shader-clock cycles spent in each phase of every K-chunk round of one forward."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench

NAMES = {1: "pass start", 2: "fetch0 issued", 3: "stash0 done", 4: "barrier0", 10: "round start",
         11: "fetch issued", 12: "mfma done", 13: "stash done", 14: "barrier", 15: "seam: piece published", 16: "seam: ticket", 17: "seam: pieces fetched",
         20: "loop end", 21: "signal done"}
# (stream3_kernel: 2 = inputs requested, 3 = ring requested + inputs in LDS, 10 = step start (previous step's
# epilogue / barrier before it), 12 = the step's MFMA stream issued, 14 = epilogue + barrier)


def main():
    argv = sys.argv[1:]
    sys.argv = ["bench.py", "--num_batches", "8"] + argv
    from deeprecsys_amd import _native as N
    N.LIB_PATH = os.path.join(os.path.dirname(N.LIB_PATH), "libdrs_hip_tl.so")   # before anything binds it
    opt = bench.parse()
    args, net, data = bench.make_model(opt, 0)
    eng = net.engine
    L = N.lib()
    L.drs_debug_timeline.argtypes = [C.POINTER(C.c_uint64), C.c_int, C.c_int]
    for kv in opt.set:
        k, v = kv.split("=")
        eng.set_option(k, int(v))
    co = opt.coalesce or 8
    bench.run_queries(eng, 200, opt.batch, opt.num_batches, 1, coalesce=co)
    buf = np.zeros(16384, dtype=np.uint64)
    L.drs_debug_timeline(buf.ctypes.data_as(C.POINTER(C.c_uint64)), 16384, 1)
    bench.run_queries(eng, co, opt.batch, opt.num_batches, 1, coalesce=co)
    n = L.drs_debug_timeline(buf.ctypes.data_as(C.POINTER(C.c_uint64)), 16384, 1)
    tags = (buf[:n] >> np.uint64(48)).astype(int)
    t = (buf[:n] & np.uint64((1 << 48) - 1)).astype(np.int64)
    print("%d stamps, total %.2f us at 100 MHz-equivalent? raw span %d ticks" % (n, 0, t[-1] - t[0]))
    prev = t[0]
    rows = []
    for tag, ti in zip(tags, t):
        rows.append((NAMES.get(tag, str(tag)), ti - prev))
        prev = ti
    # aggregate per phase
    agg = {}
    for name, d in rows:
        a = agg.setdefault(name, [0, 0])
        a[0] += d
        a[1] += 1
    for name, (tot, cnt) in agg.items():
        print("%-14s n=%3d total=%8d ticks avg=%7.1f" % (name, cnt, tot, tot / cnt))
    print("deltas:", [(a[:6], int(b)) for a, b in rows[:int(os.environ.get("TL_ROWS", "60"))]])
    eng.close()


if __name__ == "__main__":
    main()
