#!/usr/bin/env python3
"""Per-kernel averages of a rocprofv3 --pmc counter_collection CSV."""
import collections
import csv
import re
import sys


def short(name):
    m = re.search(r"(\w+)(<[^>]*>)?\(", name.replace("(anonymous namespace)::", ""))
    return (m.group(1) + (m.group(2) or "")) if m else name[:50]


def main(path):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        d[(short(r["Kernel_Name"]), r["Grid_Size"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, v in sorted(d.items()):
        print("%-40s grid=%-8s %-14s n=%-5d avg=%-14.2f min=%-12.2f max=%.2f" %
              (k[0], k[1], k[2], len(v), sum(v) / len(v), min(v), max(v)))


if __name__ == "__main__":
    main(sys.argv[1])
