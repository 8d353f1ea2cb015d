#!/usr/bin/env python3
"""Per-kernel totals of a rocprofv3 kernel_trace.csv over its second half (steady state): calls,
average duration, sum of durations as a share of the wall span (shares add up to more than
100 % when launches overlap), and the union coverage of the span."""
import collections
import csv
import re
import sys


def short(name):
    n = name.replace("(anonymous namespace)::", "").replace("drs::", "").replace("void ", "")
    return re.sub(r"\(.*", "", n)[:48]


def main(path):
    ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Grid_Size", ""))
          for r in csv.DictReader(open(path))]
    ev.sort()
    ev = ev[len(ev) // 2:]
    span = ev[-1][1] - ev[0][0]
    d = collections.defaultdict(list)
    for s, e, n, g in ev:
        d[(n, g)].append(e - s)
    print("span %.1f us, %d launches" % (span / 1e3, len(ev)))
    for (n, g), v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        print("%-48s grid=%-9s n=%-5d avg=%8.1f us  sum=%5.1f %% of span" % (n, g, len(v), sum(v) / len(v) / 1e3, 100.0 * sum(v) / span))
    cov, cur_s, cur_e = 0, None, None
    for s, e, n, g in ev:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                cov += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    cov += cur_e - cur_s
    print("some kernel running: %.1f %% of the span" % (100.0 * cov / span))


if __name__ == "__main__":
    main(sys.argv[1])
