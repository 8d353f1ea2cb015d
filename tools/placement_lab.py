"""GPU-box lab: what property of the table arena's allocation moves the gather?  

Builds ONE model (bench.py's workload shapes, device-filled tables, staged input sets), then makes a
series of copies of the table arena -- plain hipMalloc, and the virtual-memory API with a chosen
address alignment / physical chunk size / chunk order ("table_alloc", "table_vmm_*") -- and times
the model's own full launch sets on each copy with the gather's device-clock stamps (the chip to
itself: "shared_stream" 1).  Every candidate is timed again in reverse order at the end (is the
figure a property of the allocation or of the moment?), then once more after the losing copies were
freed.  Prints one JSON object; run under gpurun and keep the output in profiles/.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

try:                     # (first: torch bundles its own HIP runtime, which must be the one in the process)
    import torch  # noqa: F401
except Exception:  # noqa: BLE001
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B  # noqa: E402
from deeprecsys_amd import _native as N  # noqa: E402

# the lab's options (probes, address moves, pool selection ...) live in the lab build of the library only
_LAB = os.path.join(ROOT, "deeprecsys_amd", "libdrs_hip_lab.so")
if not os.path.exists(_LAB):
    sys.exit("tools/placement_lab.py needs deeprecsys_amd/libdrs_hip_lab.so: make -C deeprecsys_amd/csrc lab-lib")
N.LIB_PATH = _LAB

MB = 1 << 20
RECIPES = [
    # (name, table_alloc, chunk bytes, align bytes, shuffle)
    ("hipMalloc", 0, 0, 0, 0),
    ("hipMalloc", 0, 0, 0, 0),
    ("hipMalloc", 0, 0, 0, 0),
    ("vmm one handle", 1, 0, 0, 0),
    ("vmm one handle, 1 GB aligned", 1, 0, 1024 * MB, 0),
    ("vmm one handle, 4 GB aligned", 1, 0, 4096 * MB, 0),
    ("vmm 2 MB chunks", 1, 2 * MB, 0, 0),
    ("vmm 32 MB chunks", 1, 32 * MB, 0, 0),
    ("vmm 256 MB chunks", 1, 256 * MB, 0, 0),
    ("vmm 1 GB chunks, 1 GB aligned", 1, 1024 * MB, 1024 * MB, 0),
    ("vmm 2 MB chunks, shuffled", 1, 2 * MB, 0, 1),
    ("vmm 256 MB chunks, shuffled", 1, 256 * MB, 0, 1),
]


def gather_alone_us(eng, nb, bs, co, sets):
    prev = eng.get_option("shared_stream")
    eng.set_option("shared_stream", 1)
    try:
        for phase, n_sets in (("warm", max(8, sets // 4)), ("timed", sets)):
            if phase == "timed":
                eng.reset_kernel_time()
                eng.set_profiling(1)
            for g in range(n_sets):
                eng.forward_multi_async(0, [(g * co + k) % nb for k in range(co)], [bs] * co)
                eng.wait(0)
        eng.set_profiling(0)
        ms, n = eng.kernel_time(N.KERNEL_SLS_CLOCK)
        nbytes = eng.kernel_bytes(N.KERNEL_SLS_CLOCK)
    finally:
        eng.set_profiling(0)
        eng.set_option("shared_stream", prev)
    return ms / n * 1e3, nbytes / (ms * 1e-3) / 1e9 / B.HBM_PEAK_GBS


def piped_qps(eng, nb, bs, co, slots, queries):
    B.run_queries(eng, queries // 4, bs, nb, slots, coalesce=co)
    return queries / B.run_queries(eng, queries, bs, nb, slots, coalesce=co)


def h2d_gbs():
    """copy-engine rate as the rest of the process sees it (the round-4 'hipFree slows the copies' report)"""
    try:
        src = torch.empty(64 * MB, dtype=torch.uint8).pin_memory()
        dst = torch.empty(64 * MB, dtype=torch.uint8, device="cuda")
        for _ in range(3):
            dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        return round(10 * 64 * MB / (time.perf_counter() - t0) / 1e9, 2)
    except Exception as ex:  # noqa: BLE001
        return "n/a: %s" % ex


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="rmc1")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--sets", type=int, default=192)
    ap.add_argument("--queries", type=int, default=49152, help="pipelined run per candidate (0: skip)")
    ap.add_argument("--recipes", default="", help="comma-separated indices into RECIPES (default: all)")
    ap.add_argument("--no_free", action="store_true")
    ap.add_argument("--scan", type=int, default=0, help="instead of the recipes: up to this many plain hipMalloc copies, one "
                    "after the other until HBM is full (the allocator hands out memory in address order: a scan of the "
                    "gather's time over the position of the arena in HBM)")
    ap.add_argument("--scan_alloc", default="0,0", help="--scan: table_alloc,table_vmm_chunk[,table_vmm_align] of the copies; several, "
                    "separated by ';', are taken in turn")
    ap.add_argument("--swap", action="store_true", help="--scan: then swap the physical memory of the fastest and the slowest "
                    "virtual-memory-API arena, and map each again at a fresh address")
    ap.add_argument("--variants", default="", help="--scan: engine options (k=v,k=v) under which every arena is timed once more")
    ap.add_argument("--follow", type=int, default=0, help="--scan: then move the fastest and the slowest virtual-memory-API "
                    "arena through this many fresh address ranges each")
    ap.add_argument("--va_search", type=int, default=0, help="instead: one arena of the virtual-memory API mapped at this many "
                    "address ranges in turn, each timed twice")
    ap.add_argument("--va_align", type=int, default=0)
    ap.add_argument("--va_perturb", type=int, default=0, help="--va_search: 4 KiB device allocations made before each new range")
    ap.add_argument("--rows", type=int, default=0, help="override the rows per table (a smaller arena: a finer scan)")
    o = ap.parse_args()
    opt = B.parse(["--workload", o.workload, "--batch", str(o.batch), "--table_placements", "1"])
    opt.slots = 3
    if o.rows:
        B.WORKLOADS[o.workload] = dict(B.WORKLOADS[o.workload], rows=o.rows)
    args, net, data = B.make_model(opt, 0)
    eng = net.engine
    nb, bs = opt.num_batches, o.batch
    co = int(eng.get_option("preferred_coalesce"))
    res = {"workload": o.workload, "batch": bs, "coalesce": co, "table_bytes": eng.get_option("table_bytes"),
           "arena_of_drs_create": {k: eng.get_option(k) for k in ("table_kind", "table_select_pool", "table_select_kept", "table_select_best_ns",
                                                                  "table_select_worst_ns", "table_select_kept_worst_ns", "table_select_ms")},
           "gpu": B.gpu_state(0), "h2d_gbs_before": h2d_gbs(), "candidates": []}
    picks = [int(x) for x in o.recipes.split(",")] if o.recipes else list(range(len(RECIPES)))

    def measure(tag, k):
        eng.set_option("table_placement", k)
        us, frac = gather_alone_us(eng, nb, bs, co, o.sets)
        d = {"when": tag, "gather_alone_us": round(us, 2), "frac": round(frac, 4)}
        if o.queries:
            d["piped_qps"] = round(piped_qps(eng, nb, bs, co, 3, o.queries))
        return d

    if o.va_search:
        # ONE arena (virtual-memory API, no copies), mapped at a series of address ranges: the same memory, the same
        # tables -- only the address (and the page-table blocks behind it) changes
        eng.set_option("table_alloc", 1)
        eng.set_option("table_vmm_chunk", -1)
        eng.set_option("table_vmm_align", o.va_align)
        eng.set_option("table_placement", -3)
        k = eng.get_option("table_placement")
        eng.set_option("table_placement", k)
        eng.set_option("table_placement", -2)          # the hipMalloc arena of drs_create goes
        cand = [dict(measure("first", 0), va=0, address=hex(eng.get_option("table_address")))]
        for i in range(1, o.va_search):
            eng.set_option("table_va_next", o.va_perturb)
            cand.append(dict(measure("first", 0), va=i, address=hex(eng.get_option("table_address"))))
        for c in reversed(cand):
            eng.set_option("table_va_goto", c["va"])
            c["again"] = measure("again", 0)["gather_alone_us"]
        best = int(np.argmin([c["gather_alone_us"] for c in cand]))
        eng.set_option("table_va_select", best)
        res["va_search"] = cand
        res["va_us"] = [c["gather_alone_us"] for c in cand]
        res["va_us_again"] = [c["again"] for c in cand]
        res["kept"] = dict(measure("kept, the other ranges given back", 0), va=best, address=hex(eng.get_option("table_address")),
                           placements=eng.get_option("table_placements"))
        res["h2d_gbs_after"] = h2d_gbs()
        print(json.dumps(res, indent=1))
        eng.close()
        return
    if o.scan:
        cycle = [tuple(int(x) for x in c.split(",")) for c in o.scan_alloc.split(";")]
        scan = [dict(measure("first", 0), k=0, address=hex(eng.get_option("table_address")), alloc=(0, 0))]
        for i in range(o.scan):
            alloc, chunk = cycle[i % len(cycle)][:2]
            eng.set_option("table_alloc", alloc)
            eng.set_option("table_vmm_chunk", chunk)
            eng.set_option("table_vmm_align", cycle[i % len(cycle)][2] if len(cycle[i % len(cycle)]) > 2 else 0)
            try:
                eng.set_option("table_placement", -3)
            except N.DrsError:
                break
            k = eng.get_option("table_placement")
            scan.append(dict(measure("first", k), k=k, address=hex(eng.get_option("table_address")), alloc=cycle[i % len(cycle)]))
        for c in scan:
            eng.set_option("table_probe_gather", c["k"])
            c["probe_gather_us"] = eng.get_option("table_probe_gather_ns") / 1e3
        T = len(net.ln_emb) if hasattr(net, "ln_emb") else eng.T
        for c in scan:
            for name, win, srt in (("probe_gbs", 0, 0), ("probe_win_gbs", T, 0), ("probe_win_sorted_gbs", T, 1)):
                eng.set_option("table_probe_windows", win)
                eng.set_option("table_probe_sorted", srt)
                eng.set_option("table_probe", c["k"])
                c[name] = eng.get_option("table_probe_mbs") / 1e3
        for c in scan[::max(1, len(scan) // 8)]:
            c["again"] = measure("again", c["k"])["gather_alone_us"]
        res["probe_gbs"] = [c["probe_gbs"] for c in scan]
        # the same arenas under the other gather kernels / work orders (does the level belong to the memory or to how
        # one kernel walks it?)
        for key, val in (kv.split("=") for kv in o.variants.split(",") if kv):
            prev = eng.get_option(key)
            eng.set_option(key, int(val))
            for c in scan:
                eng.set_option("table_placement", c["k"])
                c["%s=%s" % (key, val)] = round(gather_alone_us(eng, nb, bs, co, o.sets)[0], 2)
            eng.set_option(key, prev)
            res["%s=%s" % (key, val)] = [c["%s=%s" % (key, val)] for c in scan]
        if o.follow:
            # memory or address, second take: the fastest and the slowest virtual-memory-API arena each move through
            # fresh address ranges (the same memory every time)
            vm = [c for c in scan if c["alloc"][0] == 1]
            if len(vm) >= 2:
                fast, slow = min(vm, key=lambda c: c["gather_alone_us"]), max(vm, key=lambda c: c["gather_alone_us"])
                res["follow"] = {}
                for name, c in (("fast", fast), ("slow", slow)):
                    eng.set_option("table_placement", c["k"])
                    runs = [c["gather_alone_us"]]
                    for _ in range(o.follow):
                        eng.set_option("table_va_next", 0)
                        runs.append(measure("moved", c["k"])["gather_alone_us"])
                    res["follow"][name] = {"k": c["k"], "us_at_successive_addresses": runs}
        res["scan"] = scan
        res["scan_us"] = [c["gather_alone_us"] for c in scan]
        if o.swap:
            # memory or mapping?  the fastest and the slowest virtual-memory-API arena swap their physical handles,
            # then each is mapped once more at a fresh address
            vm = [c for c in scan if c["alloc"][0] == 1]
            if len(vm) >= 2:
                fast, slow = min(vm, key=lambda c: c["gather_alone_us"]), max(vm, key=lambda c: c["gather_alone_us"])
                sw = {"fast": {"k": fast["k"], "address": fast["address"], "before": fast["gather_alone_us"]},
                      "slow": {"k": slow["k"], "address": slow["address"], "before": slow["gather_alone_us"]}}
                eng.set_option("table_vmm_swap", (fast["k"] << 16) | slow["k"])
                sw["fast"]["with_the_slow_arenas_memory"] = measure("swapped", fast["k"])["gather_alone_us"]
                sw["slow"]["with_the_fast_arenas_memory"] = measure("swapped", slow["k"])["gather_alone_us"]
                for c in (fast, slow):
                    eng.set_option("table_placement", c["k"])
                    eng.set_option("table_vmm_remap", c["k"])
                    sw["fast" if c is fast else "slow"]["remapped_at"] = hex(eng.get_option("table_address"))
                    sw["fast" if c is fast else "slow"]["after_remap_at_a_fresh_address"] = measure("remapped", c["k"])["gather_alone_us"]
                res["swap"] = sw
        res["h2d_gbs_with_all_held"] = h2d_gbs()
        eng.set_option("table_placement", int(np.argmin(res["scan_us"])))
        t0 = time.perf_counter()
        eng.set_option("table_placement", -2)
        res["free_losers_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
        res["after_free"] = measure("best, after the others were freed", 0)
        res["h2d_gbs_after_free"] = h2d_gbs()
        # ... and what a fresh allocation gets now that everything else is free again
        eng.set_option("table_alloc", 0)
        eng.set_option("table_placement", -3)
        res["fresh_after_free"] = dict(measure("fresh hipMalloc after the frees", 1), address=hex(eng.get_option("table_address")))
        print(json.dumps(res, indent=1))
        eng.close()
        return
    c0 = {"recipe": "hipMalloc (drs_create)", "address": hex(eng.get_option("table_address")), "runs": [measure("first", 0)]}
    res["candidates"].append(c0)
    for r in picks:
        name, alloc, chunk, align, shuf = RECIPES[r]
        eng.set_option("table_alloc", alloc)
        eng.set_option("table_vmm_chunk", chunk)
        eng.set_option("table_vmm_align", align)
        eng.set_option("table_vmm_shuffle", shuf)
        t0 = time.perf_counter()
        try:
            eng.set_option("table_placement", -1)
        except N.DrsError as ex:
            res["candidates"].append({"recipe": name, "error": str(ex)})
            continue
        k = eng.get_option("table_placement")
        c = {"recipe": name, "k": k, "address": hex(eng.get_option("table_address")),
             "alloc_copy_ms": round((time.perf_counter() - t0) * 1e3, 1), "runs": [measure("first", k)]}
        res["candidates"].append(c)
        print(json.dumps(c), file=sys.stderr, flush=True)
    live = [c for c in res["candidates"] if "k" in c or c is c0]
    for c in reversed(live):
        c["runs"].append(measure("again, reverse order", c.get("k", 0)))
    res["h2d_gbs_with_all_held"] = h2d_gbs()
    best = min(live, key=lambda c: min(r["gather_alone_us"] for r in c["runs"]))
    res["best"] = best["recipe"]
    if not o.no_free:
        eng.set_option("table_placement", best.get("k", 0))
        t0 = time.perf_counter()
        eng.set_option("table_placement", -2)
        res["free_losers_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
        best["runs"].append(measure("after the others were freed", 0))
        res["h2d_gbs_after_free"] = h2d_gbs()
    print(json.dumps(res, indent=1))
    eng.close()


if __name__ == "__main__":
    main()
