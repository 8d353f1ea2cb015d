#!/usr/bin/env python3
"""Copy the summaries of one tools/capture_profiles.sh session from gpurun_out/<tag>/ into
profiles/ (tracked) and derive profiles/traffic.json, the per-launch HBM bytes of the gather
kernel that bench.py reports as roofline.traffic.

    python tools/publish_profiles.py r01b r01
"""
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(tag, prefix):
    src = os.path.join(ROOT, "gpurun_out", tag)
    dst = os.path.join(ROOT, "profiles")
    names = {"bench.json": "bench.json", "bench_traced.json": "bench_traced.json",
             "rocprofv3_kernel_stats.csv": "rocprofv3_kernel_stats.csv",
             "kernel_trace_by_grid.txt": "kernel_trace_by_grid.txt", "pmc_summary.txt": "pmc_summary.txt",
             "bench_single_stream.json": "bench_single_stream.json", "bench_coalesce1.json": "bench_coalesce1.json",
             "bench_torchrun_n1.json": "bench_torchrun_n1.json"}
    for w in ("rmc1_ref", "rmc2_ref", "rmc3_ref", "rmc3", "rmc1_dot", "wnd", "ncf"):
        names["bench_%s.json" % w] = "bench_%s.json" % w
    for f in ("serve_rmc1.json", "serve_mix_wnd_ncf.json"):
        names[f] = f
    for a, b in names.items():
        p = os.path.join(src, a)
        if os.path.exists(p) and os.path.getsize(p) > 0:
            shutil.copy(p, os.path.join(dst, "%s_%s" % (prefix, b)))
            print("profiles/%s_%s" % (prefix, b))
    acc = os.path.join(src, "accelerator_mi355x")
    if os.path.isdir(acc):
        out = os.path.join(dst, "accelerator_mi355x")
        shutil.rmtree(out, ignore_errors=True)
        shutil.copytree(acc, out)
        print("profiles/accelerator_mi355x/")
    bench = json.loads(open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1])
    co = bench["config"]["queries_per_launch"]
    # gather launches of the timed region are the biggest sls_kernel grid in the PMC passes
    vals = {}
    for line in open(os.path.join(src, "pmc_summary.txt")):
        m = re.match(r"sls_kernel\S*(?:\s\S+)*?\s+grid=(\d+)\s+(FETCH_SIZE|WRITE_SIZE)\s+n=(\d+)\s+avg=([\d.]+)", line)
        if m:
            g, c, n, avg = int(m.group(1)), m.group(2), int(m.group(3)), float(m.group(4))
            if c not in vals or g > vals[c][0]:
                vals[c] = (g, n, avg)
    fetch_kb, write_kb = vals["FETCH_SIZE"][2], vals["WRITE_SIZE"][2]
    # MI355X_MICROARCH.md, HBM section: both counters are in KB; on gfx950 FETCH_SIZE counts the
    # 128-B requests of 16-B/lane reads at 64 B -> x2 (calibrated against TCC_MISS x 128 B)
    hbm = int(round(fetch_kb * 1024 * 2 + write_kb * 1024))
    tj = {"workload": "rmc1", "batch": 256, "queries_per_launch": co,
          "kernel": "sls_kernel", "grid": vals["FETCH_SIZE"][0], "launches": vals["FETCH_SIZE"][1],
          "FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb,
          "hbm_bytes_per_launch": hbm,
          "algorithmic_bytes_per_launch": bench["roofline"]["bytes_per_launch"],
          "ratio": round(hbm / bench["roofline"]["bytes_per_launch"], 4),
          "source": "profiles/%s_pmc_summary.txt: separate rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes "
                    "over `python bench.py --no_cpu_baseline --steps 1600 --warmup 160`; KB units, FETCH_SIZE x2 on "
                    "gfx950 (MI355X_MICROARCH.md HBM section)" % prefix}
    with open(os.path.join(dst, "traffic.json"), "w") as f:
        json.dump(tj, f, indent=1)
        f.write("\n")
    print(json.dumps(tj))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "r01")
