#!/usr/bin/env python3
"""Per-kernel duration statistics from a rocprofv3 --kernel-trace CSV."""
import collections
import csv
import sys


def main(path):
    rows = list(csv.DictReader(open(path)))
    d = collections.defaultdict(list)
    for r in rows:
        key = (r["Kernel_Name"].split("(")[0][-60:], r["Grid_Size_X"], r["Workgroup_Size_X"])
        d[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    print("%-62s %8s %5s %6s %9s %9s %9s %9s" % ("kernel", "grid", "wg", "n", "avg_us", "med_us", "min_us", "max_us"))
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        v = sorted(v)
        print("%-62s %8s %5s %6d %9.2f %9.2f %9.2f %9.2f" % (k[0], k[1], k[2], len(v), sum(v) / len(v) / 1e3,
                                                           v[len(v) // 2] / 1e3, v[0] / 1e3, v[-1] / 1e3))


if __name__ == "__main__":
    main(sys.argv[1])
