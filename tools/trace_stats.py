#!/usr/bin/env python3
"""Per-kernel duration statistics from a rocprofv3 --kernel-trace CSV."""
import collections
import csv
import sys


def main(path):
    rows = list(csv.DictReader(open(path)))
    d = collections.defaultdict(list)
    for r in rows:
        gx = r["Grid_Size_X"] if r.get("Grid_Size_Y", "1") in ("1", "") else "%sx%s" % (r["Grid_Size_X"], r["Grid_Size_Y"])
        key = (r["Kernel_Name"].split("(")[0][-60:], gx, r["Workgroup_Size_X"])
        d[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    print("%-62s %10s %5s %6s %9s %9s %9s %9s" % ("kernel", "grid", "wg", "n", "avg_us", "med_us", "min_us", "max_us"))
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        v = sorted(v)
        print("%-62s %10s %5s %6d %9.2f %9.2f %9.2f %9.2f" % (k[0], k[1], k[2], len(v), sum(v) / len(v) / 1e3,
                                                           v[len(v) // 2] / 1e3, v[0] / 1e3, v[-1] / 1e3))
    # idle time between consecutive launches of the same kernel+grid (back-to-back pipelines)
    print()
    print("%-62s %10s %6s %9s %9s   (start[i+1] - end[i], us; median / p10 / p90)" % ("kernel", "grid", "n", "gap_med", "period"))
    t = collections.defaultdict(list)
    for r in rows:
        gx = r["Grid_Size_X"] if r.get("Grid_Size_Y", "1") in ("1", "") else "%sx%s" % (r["Grid_Size_X"], r["Grid_Size_Y"])
        key = (r["Kernel_Name"].split("(")[0][-60:], gx)
        t[key].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    for k, v in sorted(t.items(), key=lambda kv: -len(kv[1])):
        if len(v) < 20:
            continue
        v.sort()
        gaps = sorted(v[i + 1][0] - v[i][1] for i in range(len(v) - 1))
        per = sorted(v[i + 1][0] - v[i][0] for i in range(len(v) - 1))
        print("%-62s %10s %6d %9.2f %9.2f   p10 %.2f p90 %.2f" % (k[0], k[1], len(v), gaps[len(gaps) // 2] / 1e3,
                                                              per[len(per) // 2] / 1e3, gaps[len(gaps) // 10] / 1e3,
                                                              gaps[len(gaps) * 9 // 10] / 1e3))


if __name__ == "__main__":
    main(sys.argv[1])
