#!/usr/bin/env python3
"""One line per bench JSON of a capture directory (gpurun_out/<tag> or profiles/ with a prefix)."""
import glob
import json
import os
import sys


def load(f):
    try:
        return json.loads(open(f).read().strip().splitlines()[-1])
    except Exception:
        return None


def main(d, prefix=""):
    for f in sorted(glob.glob(os.path.join(d, prefix + "*.json"))):
        x = load(f)
        if not x or "roofline" not in x:
            continue
        r = x["roofline"]
        s = r.get("single_query_launch") or {}
        h = x.get("host_inputs_leg") or {}
        c = x.get("cpu_baseline") or {}
        print("%-44s %9.1f q/s p99 %.4f ms | gather frac %s (%s us) single %s | host %s | cpu %s" % (
            os.path.basename(f), x["value"], x["latency_ms"]["p99"], r["frac"], r["avg_launch_us"], s.get("frac"),
            h.get("value"), c.get("value")))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
