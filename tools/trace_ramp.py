#!/usr/bin/env python3
"""Kernel duration against position in the run, from a rocprofv3 --kernel-trace CSV: the launches of every kernel
that makes up >= 5 % of the traced time, in start order, averaged per tenth of the sequence.

A GPU that was idle starts a run at a low shader clock and takes a few hundred launches to reach its sustained one:
a 30-launch trace of an MFMA-bound kernel reads 10-15 % slow (round 4's 119.5 TFLOP/s for gemm32_kernel was such a
trace; tools/clock_trace.py shows the clock itself).  Usage: trace_ramp.py kernel_trace.csv [flop_per_launch]
"""
import collections
import csv
import sys


def main(path, flop=0.0):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        d[name[-70:]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    total = sum(sum(x[1] for x in v) for v in d.values())
    for k, v in sorted(d.items(), key=lambda kv: -sum(x[1] for x in kv[1])):
        if sum(x[1] for x in v) < 0.05 * total or len(v) < 20:
            continue
        v.sort()
        n = len(v)
        print("%s: %d launches, avg %.2f us" % (k, n, sum(x[1] for x in v) / n / 1e3))
        for i in range(10):
            part = v[n * i // 10:n * (i + 1) // 10]
            avg = sum(x[1] for x in part) / len(part) / 1e3
            t0 = (part[0][0] - v[0][0]) / 1e6
            line = "  launches %6d..%6d (from %8.1f ms): avg %9.2f us" % (n * i // 10, n * (i + 1) // 10 - 1, t0, avg)
            if flop:
                line += "  %6.1f TFLOP/s" % (flop / avg / 1e6)
            print(line)


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.0)
