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
    skip = ("pmc_last.err",)
    for f in sorted(os.listdir(src)):
        p = os.path.join(src, f)
        if not os.path.isfile(p) or f in skip or f.endswith(".err") or os.path.getsize(p) == 0:
            continue
        if os.path.getsize(p) > 2 * 1024 * 1024:        # raw traces stay in gpurun_out/
            continue
        shutil.copy(p, os.path.join(dst, "%s_%s" % (prefix, f)))
        print("profiles/%s_%s" % (prefix, f))
    for sub in ("accelerator_mi355x", "cpu_epyc9575f"):       # latency tables in the reference's "***" format
        acc = os.path.join(src, sub)
        if os.path.isdir(acc):
            out = os.path.join(dst, sub)
            shutil.rmtree(out, ignore_errors=True)
            shutil.copytree(acc, out)
            print("profiles/%s/" % sub)
    # one traffic entry per workload whose PMC passes were taken: (bench line, summary file, its command)
    jobs = [("rmc1", "bench.json", "pmc_summary.txt",
             "python bench.py --no_cpu_baseline --timed_only --steps 2 --warmup 1 --queries_per_step 2048"),
            ("rmc3", "rmc3_bench_single_stream.json", "rmc3_pmc_summary.txt",
             "python bench.py --workload rmc3 --batch 512 --no_cpu_baseline --timed_only --steps 3 --warmup 1 "
             "--queries_per_step 2048 --set shared_stream=1"),
            ("din", "bench_din.json", "din_pmc_summary.txt",
             "python bench.py --workload din --no_cpu_baseline --timed_only --steps 3 --warmup 1 --queries_per_step 2048")]
    by = {}
    for wl, bench_file, pmc_file, cmd in jobs:
        try:
            bench = json.loads(open(os.path.join(src, bench_file)).read().strip().splitlines()[-1])
            lines = open(os.path.join(src, pmc_file)).read().splitlines()
        except (OSError, ValueError, IndexError):
            continue
        co = bench["config"]["queries_per_launch"]
        # the gather launches of the timed region are the biggest grid of a gather kernel in the PMC passes
        vals = {}
        for line in lines:
            m = re.match(r"(sls_\w+kernel|din_fused_kernel)\S*(?:\s\S+)*?\s+grid=(\d+)\s+(FETCH_SIZE|WRITE_SIZE)\s+n=(\d+)\s+avg=([\d.]+)", line)
            if m:
                k, g, c, n, avg = m.group(1), int(m.group(2)), m.group(3), int(m.group(4)), float(m.group(5))
                if c not in vals or g > vals[c][0]:
                    vals[c] = (g, n, avg, k)
        if "FETCH_SIZE" not in vals or "WRITE_SIZE" not in vals or not bench["roofline"].get("bytes_per_launch"):
            continue
        fetch_kb, write_kb = vals["FETCH_SIZE"][2], vals["WRITE_SIZE"][2]
        # MI355X_MICROARCH.md, HBM section: both counters are in KB; on gfx950 FETCH_SIZE counts the
        # 128-B requests of 16-B/lane reads at 64 B -> x2 (calibrated against TCC_MISS x 128 B)
        hbm = int(round(fetch_kb * 1024 * 2 + write_kb * 1024))
        batch = int(re.search(r"batch (\d+)", bench["config"]["workload"]).group(1))
        by[wl] = {"workload": wl, "batch": batch, "queries_per_launch": co,
                  "kernel": vals["FETCH_SIZE"][3] + " (gather)", "grid": vals["FETCH_SIZE"][0], "launches": vals["FETCH_SIZE"][1],
                  "FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb, "hbm_bytes_per_launch": hbm,
                  "algorithmic_bytes_per_launch": bench["roofline"]["bytes_per_launch"],
                  "ratio": round(hbm / bench["roofline"]["bytes_per_launch"], 4),
                  "source": "profiles/%s_%s: separate rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes over `%s`; "
                            "KB units, FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md HBM section)" % (prefix, pmc_file, cmd)}
    tj = {"by_workload": by}
    with open(os.path.join(dst, "traffic.json"), "w") as f:
        json.dump(tj, f, indent=1)
        f.write("\n")
    print(json.dumps(tj))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else sys.argv[1])
