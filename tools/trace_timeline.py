#!/usr/bin/env python3
"""Timeline of consecutive launch sets from a rocprofv3 kernel_trace.csv: for N launches in the
middle of the trace, start and end of every kernel relative to the first, the idle time since the
previous launch of the same kernel ended, and for each non-gather launch the time since the
latest gather that ended before it started (the cross-stream event latency)."""
import csv
import re
import sys


def short(name):
    n = name.replace("(anonymous namespace)::", "").replace("drs::", "").replace("void ", "")
    return re.sub(r"[<(].*", "", n)[:28]


def main(path, n=40):
    ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]))
          for r in csv.DictReader(open(path))]
    ev.sort()
    mid = ev[len(ev) // 2: len(ev) // 2 + n]
    t0 = mid[0][0]
    last_end = {}
    gather_end = None
    print("%-28s %10s %10s %8s %12s %14s" % ("kernel", "start_us", "end_us", "dur_us", "idle_same_us", "since_gather_us"))
    for s, e, k in mid:
        idle = "%.1f" % ((s - last_end[k]) / 1e3) if k in last_end else "-"
        sg = "-"
        if k.startswith("sls") or k.startswith("din_fused"):
            pass
        elif gather_end is not None:
            sg = "%.1f" % ((s - gather_end) / 1e3)
        print("%-28s %10.1f %10.1f %8.1f %12s %14s" % (k, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, idle, sg))
        last_end[k] = e
        if k.startswith("sls") or k.startswith("din_fused"):
            gather_end = e
    per = {}
    for s, e, k in ev[len(ev) // 2:]:
        per.setdefault(k, []).append((s, e))
    for k, v in per.items():
        if len(v) > 8:
            gaps = [(v[i + 1][0] - v[i][1]) / 1e3 for i in range(len(v) - 1)]
            period = (v[-1][0] - v[0][0]) / 1e3 / (len(v) - 1)
            gaps.sort()
            print("%-28s n=%d period %.2f us, idle between launches: median %.2f, p10 %.2f, p90 %.2f us"
                  % (k, len(v), period, gaps[len(gaps) // 2], gaps[len(gaps) // 10], gaps[9 * len(gaps) // 10]))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
