#!/usr/bin/env python3
"""bench.py -- queries/sec under a p99 latency SLA, DLRM-RMC1 synthetic, on N MI355X.

    python bench.py --gpus 1 --steps 2000 --warmup 200
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one query: one pass of the hot path (multi-table SparseLengthsSum gather,
bottom MLP, feature interaction, top MLP, sigmoid) over one batch of `--batch`
samples whose inputs are already resident in HBM (the reference engine keeps its
pre-generated input sets in process memory and a request only names
(batch_id, batch_size): inferenceEngine.py:83,200-215).  Queries are submitted
through the C ABI (include/drs.h) with `--slots` in flight; each query's latency is
taken from its submit to the moment its result is observed on the host, and the p99
over the timed steps is checked against the SLA.

Workload at N=1: BASELINE.json configs[1] -- DLRM-RMC1, 8 tables x 1M rows x 64-dim,
80 lookups per bag, bottom MLP 128-64-64, top MLP 576-256-64-1 (cat), batch 256.
Multi-GPU: queries are independent, the model is replicated per GPU, every rank
serves its own K steps (weak scaling); RCCL (torch.distributed "nccl") only
all-reduces the elapsed time and the latency histogram at the end.

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for every field).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
SLA_MS = 25.0              # run_DeepRecSys.sh:42 target_latency

WORKLOADS = {
    # BASELINE.json configs[1]
    "rmc1": dict(rows=1_000_000, T=8, D=64, L=80, bot="128-64-64", top="256-64-1", op="cat"),
    # the reference's own models/configs/dlrm_rm1.json
    "rmc1_ref": dict(rows=4_000_000, T=8, D=32, L=80, bot="128-64-32", top="256-64-1", op="cat"),
    "rmc2_ref": dict(rows=500_000, T=32, D=64, L=120, bot="256-128-64", top="128-64-1", op="cat"),
    "rmc3_ref": dict(rows=2_000_000, T=10, D=32, L=20, bot="2560-1024-256-32", top="512-256-1", op="cat"),
    # BASELINE.json config 3 as written: 12 tables x 10M rows x 32 (15.4 GB), run with --batch 512
    "rmc3": dict(rows=10_000_000, T=12, D=32, L=20, bot="2560-1024-256-32", top="512-256-1", op="cat"),
    "rmc1_dot": dict(rows=1_000_000, T=8, D=64, L=80, bot="128-64-64", top="256-64-1", op="dot"),
    # BASELINE config 4's models, the reference's models/configs/{wide_and_deep,ncf}.json
    "wnd": dict(kind="wnd", rows=1_000_000, T=27, D=32, L=1, bot="512", top="1024-512-256-1", op="cat"),
    "ncf": dict(kind="ncf", rows=[140_000, 140_000, 28_000, 28_000], T=4, D=64, L=1, bot="512",
                top="256-256-128-64-64", op="cat"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16000)
    ap.add_argument("--warmup", type=int, default=1600)
    ap.add_argument("--workload", default="rmc1", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--num_batches", type=int, default=32)
    ap.add_argument("--slots", type=int, default=3)
    ap.add_argument("--coalesce", type=int, default=8,
                    help="queries per launch set (the engine coalesces requests that are already queued)")
    ap.add_argument("--seed", type=int, default=123)
    ap.add_argument("--cpu_seconds", type=float, default=12.0, help="budget of the cpu_baseline sample")
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--timed_only", action="store_true",
                    help="warm-up + timed region only: skip the cross-check / single-query / host-input "
                         "legs and the CPU baseline (profiling runs: every gather launch rocprofv3 sees "
                         "is then an 8-query launch of the benchmark itself)")
    ap.add_argument("--sweep", action="store_true", help="also A/B the gather variants (stderr)")
    ap.add_argument("--set", action="append", default=[], help="engine option key=value")
    return ap.parse_args()


def make_model(opt, device):
    from deeprecsys_amd import dlrm_s_hip as M
    from deeprecsys_amd.data_generator.dlrm_data import generate_fast_input_data
    from deeprecsys_amd.utils.utils import cli
    w = WORKLOADS[opt.workload]
    args = cli([])
    args.arch_sparse_feature_size = w["D"]
    rows = w["rows"] if isinstance(w["rows"], list) else [w["rows"]] * w["T"]
    args.arch_embedding_size = "-".join(str(r) for r in rows)
    args.arch_mlp_bot, args.arch_mlp_top = w["bot"], w["top"]
    args.arch_interaction_op = w["op"]
    args.num_indices_per_lookup = w["L"]
    args.num_batches = opt.num_batches
    args.max_mini_batch_size = args.mini_batch_size = opt.batch
    args.numpy_rand_seed = opt.seed
    args.accel_table_init = "device"          # counter-based fill, bit-identical in oracle/
    args.accel_slots = opt.slots
    kind = w.get("kind", "dlrm")
    args.model_type = args.model_name = kind
    args.num_indices_per_lookup_fixed = True
    args._drs_device = device
    np.random.seed(opt.seed)
    net = {"dlrm": M.DLRM_Net, "wnd": M.Wide_and_Deep, "ncf": M.NCF}[kind](args)
    m_den = int(w["bot"].split("-")[0])
    nb, lX, lS_l, lS_i = generate_fast_input_data(opt.num_batches, opt.batch, m_den, rows, w["L"], opt.seed)
    net.create(lX[0], lS_l[0], lS_i[0], None)
    net.stage_batches(None if kind == "ncf" else lX, lS_l, lS_i)
    return args, net, (lX, lS_l, lS_i)


def run_queries(eng, n, bs, nb, slots, lat=None, start_id=0, coalesce=1):
    """Closed loop: exactly n queries, `coalesce` of them per launch set, `slots` launch
    sets in flight.  Every query's latency runs from the submit of its launch set to the
    moment the set's results are observed on the host.  Returns elapsed seconds."""
    t_submit = [0.0] * slots
    in_slot = [0] * slots
    t0 = time.perf_counter()
    i = g = 0
    while i < n:
        c = min(coalesce, n - i)
        s = g % slots
        if in_slot[s]:
            eng.wait(s)
            if lat is not None:
                lat.extend([time.perf_counter() - t_submit[s]] * in_slot[s])
        t_submit[s] = time.perf_counter()
        if c == 1:
            eng.forward_async(s, (start_id + i) % nb, bs)
        else:
            eng.forward_multi_async(s, [(start_id + i + k) % nb for k in range(c)], [bs] * c)
        in_slot[s] = c
        i += c
        g += 1
    for k in range(slots):
        s = (g + k) % slots
        if in_slot[s]:
            eng.wait(s)
            if lat is not None:
                lat.extend([time.perf_counter() - t_submit[s]] * in_slot[s])
            in_slot[s] = 0
    return time.perf_counter() - t0


def host_cores():
    """CPU cores this process may really use: affinity mask capped by the cgroup quota."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(opt, net, data, budget_s):
    """The CPU oracle (a port of the reference's CPU path, oracle/drs_oracle.c) timed on
    this host's cores on a bounded sample of the same workload."""
    from oracle import oracle as orc
    from tests import helpers as H
    w = WORKLOADS[opt.workload]
    lX, lS_l, lS_i = data
    cores = host_cores()
    lo, hi = -float(np.sqrt(1 / w["rows"])), float(np.sqrt(1 / w["rows"]))
    t0 = time.perf_counter()
    net.emb_w = [orc.fill_table_uniform(w["rows"], w["D"], t, lo, hi, opt.seed, nthreads=cores)
                 for t in range(w["T"])]
    om = H.oracle_model(net)
    fill_s = time.perf_counter() - t0
    om.forward(lX[0], lS_i[0], lS_l[0], bs=opt.batch, nthreads=cores)   # warm
    n, t0 = 0, time.perf_counter()
    while True:
        om.forward(lX[n % len(lX)], lS_i[n % len(lX)], lS_l[n % len(lX)], bs=opt.batch, nthreads=cores)
        n += 1
        el = time.perf_counter() - t0
        if el >= budget_s or n >= 20000:
            break
    return {"value": round(n / el, 2), "unit": "queries/s", "cores": cores, "kind": "port",
            "sample": "%d queries of batch %d (%s) in %.1f s, OpenMP over %d threads; "
                      "tables filled in %.1f s" % (n, opt.batch, opt.workload, el, cores, fill_s)}


def main():
    opt = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    from deeprecsys_amd import _native as N

    args, net, data = make_model(opt, local)
    eng = net.engine
    for kv in opt.set:
        k, v = kv.split("=")
        eng.set_option(k, int(v))
    bs, nb, slots, co = opt.batch, opt.num_batches, opt.slots, opt.coalesce
    stream_mode = {"2": "pipelined (gathers back to back on one stream, MLP launches on a second)",
                   "1": "single stream", "0": "one stream per launch set"}[
        dict(kv.split("=") for kv in opt.set).get("shared_stream", "2")]

    def barrier():
        eng.sync()
        if dist is not None:
            import torch
            dist.barrier()
            torch.cuda.synchronize()

    # warmup
    run_queries(eng, opt.warmup, bs, nb, slots, coalesce=co)
    # timed region: exactly K steps per rank, barrier + sync on both sides.  Profiling level 1
    # (device clock stamps of the gather launch, handed over with the results: no copy, no
    # sync, no extra launch) stays on inside it, so the roofline figure is taken over the
    # timed region itself.
    lat = []
    eng.reset_kernel_time()
    eng.set_profiling(1)
    barrier()
    elapsed = run_queries(eng, opt.steps, bs, nb, slots, lat, coalesce=co)
    barrier()
    eng.set_profiling(0)
    sls_ms, sls_n = eng.kernel_time(N.KERNEL_SLS_CLOCK)
    gbytes = eng.gather_bytes(0, bs) * co          # algorithmic bytes of one gather launch

    extra = not opt.timed_only
    ev_ms = ev_n = mlp_ms = mlp_n = one_ms = one_n = 0
    if extra:
        # cross-check leg (not part of `value`): HIP events recorded around the gather launch on the
        # stream it is launched on; they bracket several us of packet processing as well
        eng.reset_kernel_time()
        eng.set_profiling(2)
        run_queries(eng, min(opt.steps, 1000), bs, nb, slots, coalesce=co)
        eng.set_profiling(0)
        ev_ms, ev_n = eng.kernel_time(N.KERNEL_SLS)
        mlp_ms, mlp_n = eng.kernel_time(N.KERNEL_MLP)
        # reference point: the same gather serving ONE query per launch, nothing else in flight
        eng.reset_kernel_time()
        eng.set_profiling(1)
        run_queries(eng, 300, bs, nb, 1, coalesce=1)
        eng.set_profiling(0)
        one_ms, one_n = eng.kernel_time(N.KERNEL_SLS_CLOCK)
    # PCIe-inclusive leg (never `value`): the same queries handed over as HOST arrays per call --
    # int64 ids / int32 lengths / fp32 dense, the reference's run_queues signature -- through
    # drs_forward_inputs_async with `slots` calls in flight
    lX, lS_l, lS_i = data
    L = WORKLOADS[opt.workload]["L"]
    host_sets = [([np.ascontiguousarray(t[:bs * L]) for t in lS_i[b]], [np.ascontiguousarray(t[:bs]) for t in lS_l[b]],
                  None if WORKLOADS[opt.workload].get("kind") == "ncf" else np.ascontiguousarray(lX[b][:bs]))
                 for b in range(min(nb, 4))]
    def host_leg(n):
        busy = [False] * slots
        t0 = time.perf_counter()
        for i in range(n):
            s_ = i % slots
            if busy[s_]:
                eng.wait(s_, bs)
            ids, lens, x = host_sets[i % len(host_sets)]
            eng.forward_inputs_async(x, ids, lens, bs, slot=s_)
            busy[s_] = True
        for s_ in range(slots):
            if busy[s_]:
                eng.wait(s_, bs)
        return time.perf_counter() - t0
    host_n, host_el = 0, 1.0
    if extra:
        host_leg(100)
        host_n = 1000
        host_el = host_leg(host_n)

    from deeprecsys_amd import stats
    hist = stats.latency_histogram(lat)
    tot_elapsed, tot_queries = elapsed, opt.steps
    if dist is not None:
        import torch
        # the single collective of the run: MAX(elapsed), SUM(count, latency histogram) -- 32 KB
        tot_elapsed, tot_queries, hist = stats.allreduce_run_stats(dist, elapsed, opt.steps, hist,
                                                                   device=torch.device("cuda", local))
    p50, p95, p99 = (stats.percentile_from_histogram(hist, q) for q in (50, 95, 99))

    if rank == 0:
        w = WORKLOADS[opt.workload]
        # HBM bytes per gather launch cannot be counted from inside this process: they come from
        # separate rocprofv3 --pmc passes over this same command (profiles/README.md), committed
        # as profiles/traffic.json and reported here only when they were taken on this workload
        traffic, traffic_src = None, None
        try:
            with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "traffic.json")) as f:
                tj = json.load(f)
            if tj.get("workload") == opt.workload and tj.get("batch") == bs and tj.get("queries_per_launch") == co:
                traffic, traffic_src = tj["hbm_bytes_per_launch"], tj["source"]
        except (OSError, ValueError, KeyError):
            pass
        ach = gbytes / (sls_ms / max(sls_n, 1) * 1e-3) / 1e9 if sls_n else None
        out = {
            "metric": "queries/sec under p99 latency SLA, %s synthetic"
                      % ("DLRM-RMC1" if opt.workload == "rmc1" else opt.workload),
            "value": round(tot_queries / tot_elapsed, 1),
            "unit": "queries/s",
            "n_gpus": world, "steps": opt.steps, "warmup": opt.warmup,
            "ms_per_step": round(tot_elapsed / opt.steps * 1e3, 5),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s-%s: %d tables x %s rows x %d-dim, %d lookups/bag, "
                                   "bot %s, top %s (%s), batch %d, %d resident input sets"
                                   % (w.get("kind", "dlrm").upper(), opt.workload.upper(), w["T"], w["rows"], w["D"], w["L"], w["bot"],
                                      w["top"], w["op"], bs, nb),
                       "parallelism": "dp%d (model replicated, independent queries)" % world,
                       "queries_per_launch": co, "launch_sets_in_flight": slots,
                       "streams": stream_mode,
                       "inputs": "device-resident (pre-staged)"},
            "latency_ms": {"p50": round(p50, 4), "p95": round(p95, 4), "p99": round(p99, 4),
                           "sla": SLA_MS, "sla_met": bool(p99 <= SLA_MS)},
            "roofline": {"bound": "hbm", "kernel": "sls_kernel (multi-table SparseLengthsSum)",
                         "achieved": None if ach is None else round(ach, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": None if ach is None else round(ach / HBM_PEAK_GBS, 4),
                         "traffic": traffic, "traffic_source": traffic_src,
                         "bytes_per_launch": gbytes,
                         "avg_launch_us": None if not sls_n else round(sls_ms / sls_n * 1e3, 3),
                         "launches_timed": sls_n,
                         "timer": "device wall clock stamps of the launch's own workgroups "
                                  "(max end - min start); hip-event bracket for comparison",
                         "hip_event_avg_us": None if not ev_n else round(ev_ms / ev_n * 1e3, 3),
                         "gather_end_to_set_end_event_us": None if not mlp_n else round(mlp_ms / mlp_n * 1e3, 3),
                         # all gather launches of the timed region over its wall time: how busy
                         # the launch structure keeps HBM, gaps and MLP phases included
                         "sustained_over_timed_region": {
                             "GBps": round(gbytes * (opt.steps / co) / elapsed / 1e9, 1),
                             "frac": round(gbytes * (opt.steps / co) / elapsed / 1e9 / HBM_PEAK_GBS, 4)},
                         "single_query_launch": None if not one_n else {
                             "bytes": gbytes // co, "avg_launch_us": round(one_ms / one_n * 1e3, 3),
                             "frac": round(gbytes / co / (one_ms / one_n * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}},
        }
        if extra:
          out["host_inputs_leg"] = {
              "value": round(host_n / host_el, 1), "unit": "queries/s", "queries": host_n,
              "what": "PCIe-inclusive: per-call host arrays (%d KB/query) converted into pinned memory and "
                      "read in place by the kernels, one query per launch set, %d calls in flight"
                      % ((bs * (len(lS_i[0]) * L * 8 + len(lS_i[0]) * 4 + lX[0].shape[1] * 4)) // 1024, slots)}
        if not opt.no_cpu_baseline and not opt.timed_only and world == 1:
            out["cpu_baseline"] = cpu_baseline(opt, net, data, opt.cpu_seconds)
        print(json.dumps(out), flush=True)

    if opt.sweep and rank == 0:
        results = []
        for exact in (1, 0):
            for u in (4, 8, 16, 20):
                eng.set_option("sls_exact", exact)
                eng.set_option("sls_u", u)
                run_queries(eng, 200, bs, nb, slots, coalesce=co)
                eng.reset_kernel_time()
                el = run_queries(eng, 2000, bs, nb, slots, coalesce=co)
                eng.set_profiling(1)
                run_queries(eng, 1000, bs, nb, slots, coalesce=co)
                eng.set_profiling(0)
                ms, n = eng.kernel_time(N.KERNEL_SLS_CLOCK)
                results.append({"exact": exact, "u": u, "qps": round(2000 / el, 1),
                                "sls_us": round(ms / n * 1e3, 3),
                                "GBps": round(gbytes / (ms / n * 1e-3) / 1e9, 1)})
                print("sweep", json.dumps(results[-1]), file=sys.stderr, flush=True)
    eng.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
