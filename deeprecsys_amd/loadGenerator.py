"""Load generator: Poisson arrivals, query-size distributions, CPU/accelerator routing.

Mirror of the reference's loadGenerator.py:14-227 (same function names and
signatures; distributions and partitioning pinned by tests/golden/harness.json).
Extension for this build: `num_accels` accelerator engines, one per
GPU, all pulling from the single accelRequestQueue; with no CPU engines configured
every query goes to the accelerators whatever its size.

Requests for the accelerators that are generated back to back (no sleep between them: the
offered load exceeds what one put per packet can carry, ~40 k/s through a
multiprocessing.Queue) travel as ONE put of a list of up to `--accel_req_batch`
ServiceRequests -- the packets themselves are unchanged (utils/packets.py:6-22), an
engine that finds a list serves its members like requests it found waiting.  The list is
flushed before every sleep, so at low load every request still leaves at once.
"""
import math
import os
import sys
import time

import numpy as np

from .scheduler import Scheduler
from .utils.packets import ServiceRequest
from .utils.utils import debugPrint, mix_models


def model_arrival_times(args):
    # loadGenerator.py:14-17
    return np.random.poisson(lam=args.avg_arrival_rate, size=args.nepochs * args.num_batches)


def model_batch_size_distribution(args):
    """Query sizes for the num_batches query slots, clamped to [1, max_mini_batch_size]
    (loadGenerator.py:20-43)."""
    kind, n = args.batch_size_distribution, args.num_batches
    if kind == "normal":
        sizes = np.random.normal(args.avg_mini_batch_size, args.var_mini_batch_size, n)
    elif kind == "lognormal":
        sizes = np.random.lognormal(args.avg_mini_batch_size, args.var_mini_batch_size, n)
    elif kind == "fixed":
        sizes = np.array([args.avg_mini_batch_size for _ in range(n)])
    elif kind == "file":
        with open(args.batch_dist_file, "r") as f:
            percentiles = [float(line.rstrip()) for line in f.readlines()]
        sizes = [int(percentiles[int(np.random.uniform(0, len(percentiles)))]) for _ in range(n)]
    else:
        raise ValueError("unknown batch_size_distribution " + str(kind))
    for i in range(n):
        sizes[i] = int(max(min(sizes[i], args.max_mini_batch_size), 1))
    return sizes


def partition_requests(args, batch_size):
    """Cut a query into sub_task_batch_size pieces, remainder last (loadGenerator.py:46-54)."""
    full, rest = divmod(int(batch_size), int(args.sub_task_batch_size))
    return [args.sub_task_batch_size] * full + ([rest] if rest > 0 else [])


def loadGenSleep(sleeptime):
    # OS sleep is too coarse below ~5.5 ms: spin instead (loadGenerator.py:57-64)
    if sleeptime > 0.0055:
        time.sleep(sleeptime)
        return
    t0 = time.time()
    while (time.time() - t0) < sleeptime:
        pass


def accel_engine_count(args):
    return int(getattr(args, "num_accels", 1)) if args.model_accel else 0


def loadGenerator(args, requestQueue, loadGeneratorReturnQueue, inferenceEngineReadyQueue, pidQueue,
                  accelRequestQueue):
    for _ in range(args.inference_engines):        # block until every engine built its model
        inferenceEngineReadyQueue.get()

    if getattr(args, "_gen_shard", (0, 1))[1] > 1:
        np.random.seed(args.numpy_rand_seed)       # k generators must draw the SAME query sizes (the reference's one generator draws from an unseeded stream)
    model_arrival_times(args)                      # consumed for RNG-stream parity (unused, as in the reference)
    batch_size_distributions = model_batch_size_distribution(args)
    n_accel = accel_engine_count(args)
    n_cpu = args.inference_engines - n_accel
    # --load_generators k (this build, SURVEY.md 8e at N = 8): generator g of k serves the query slots with
    # batch_id % k == g of every epoch to ITS queue (the accelerator engines e with e % k == g listen there), at
    # 1/k of the arrival rate each -- the same queries, the same aggregate rate, k Python loops instead of one.
    # Query sizes come from the same seeded draw in every generator; the gaps from a per-generator stream.
    shard, n_shards = getattr(args, "_gen_shard", (0, 1))
    if n_shards > 1:
        n_accel = len([e for e in range(n_accel) if e % n_shards == shard])   # the sentinels this generator owes
        np.random.seed(args.numpy_rand_seed + 104729 * (shard + 1))

    cpu_sub_requests = cpu_requests = accel_requests = 0
    tune_batch, tune_accel = args.tune_batch_qps, args.tune_accel_qps
    tuning_batch_qps, tuning_accel_qps = args.tune_batch_qps, False
    if tuning_batch_qps:
        rates = np.logspace(math.log(args.min_arr_range, 10), math.log(args.max_arr_range, 10),
                            num=args.arr_steps)
        print("Arrival rates to try: ", rates)
        sys.stdout.flush()
        batch_configs = np.array([int(x) for x in args.batch_configs.split("-")], dtype=int)
        args.sub_task_batch_size = batch_configs[0]
        args.accel_request_size_thres = 1024     # accelerator sweep starts after the batch sweep
    arrival_rate = args.avg_arrival_rate

    query_scheduler = Scheduler(args, requestQueue, accelRequestQueue, pidQueue, mode="cpu")
    accel_query_scheduler = Scheduler(args, requestQueue, accelRequestQueue, pidQueue, mode="accel")

    # mixed-model stream: each query is for one of the engine's models, drawn from its own
    # seeded stream so the reference's RNG consumption (sizes, arrivals) is untouched
    shares = [share for _a, share in mix_models(args)]
    mix_rng = np.random.RandomState(args.numpy_rand_seed + 7919) if shares else None

    # requests per put: the reference's one packet per put whenever CPU engines are configured (a list would let small
    # CPU-routed queries overtake accelerator requests still waiting in it); at most 8 when several accelerator engines
    # pull from this queue (a whole list lands on ONE engine: 16 would be two launch sets there while its peers idle)
    req_batch = max(1, int(getattr(args, "accel_req_batch", 1)))
    if n_cpu > 0:
        req_batch = 1
    elif n_accel > 1:
        req_batch = min(req_batch, 8)
    pending = []                                   # accelerator requests not yet put

    gstats = {"put": 0.0, "sleep": 0.0, "puts": 0} if os.environ.get("DRS_ENGINE_STATS") else None
    t_gen0 = time.perf_counter()

    def flush():
        if pending:
            t_f = time.perf_counter() if gstats else 0.0
            accelRequestQueue.put(pending[0] if len(pending) == 1 else list(pending))
            del pending[:]
            if gstats:
                gstats["put"] += time.perf_counter() - t_f
                gstats["puts"] += 1

    epoch = exp_epochs = 0
    while tuning_batch_qps or (exp_epochs < args.nepochs):
        # inter-arrival gaps: one draw per query from the same numpy stream as the reference's
        # per-query poisson(size=1) (:198-199) -- a block draw yields the same values; only while the
        # schedulers cannot change the rate under it
        gaps = None if (tuning_batch_qps or tuning_accel_qps) else np.random.poisson(lam=arrival_rate * n_shards, size=args.num_batches)
        for batch_id in range(args.num_batches):
            if tuning_batch_qps and pidQueue.qsize() > 0:
                args, arrival_rate, tuning_batch_qps = query_scheduler.run(pidQueue.get())
                if not tuning_batch_qps:
                    print("Finished batch size scheduler ")
                    if args.model_accel and args.tune_accel_qps:
                        print("Starting accel scheduler")
                        tuning_accel_qps = True
                    continue
            if args.model_accel and tuning_accel_qps and pidQueue.qsize() > 0:
                args, arrival_rate, tuning_accel_qps = accel_query_scheduler.run(pidQueue.get())
                if not tuning_accel_qps:
                    continue

            request_size = int(batch_size_distributions[batch_id])
            model_id = int(mix_rng.choice(len(shares), p=shares)) if shares else 0
            if batch_id % n_shards != shard:
                continue                                       # another generator's query slot
            exploring = bool(tuning_batch_qps or tuning_accel_qps)
            to_accel = n_accel > 0 and (n_cpu == 0 or request_size >= args.accel_request_size_thres)
            if to_accel:
                # whole query to an accelerator (loadGenerator.py:162-177)
                request = ServiceRequest(batch_id=batch_id, epoch=epoch, batch_size=request_size,
                                         sub_id=0, total_sub_batches=1, exp_packet=exploring,
                                         model_id=model_id)
                accel_requests += 1
                request.arrival_time = time.time()
                pending.append(request)
                if len(pending) >= req_batch:
                    flush()
            else:
                pieces = partition_requests(args, request_size)
                for i, piece in enumerate(pieces):
                    request = ServiceRequest(batch_id=batch_id, epoch=epoch, batch_size=piece, sub_id=i,
                                             total_sub_batches=len(pieces), exp_packet=exploring)
                    cpu_sub_requests += 1
                    request.arrival_time = time.time()
                    requestQueue.put(request)
                cpu_requests += 1
            gap = (np.random.poisson(lam=arrival_rate, size=1)[0] if gaps is None else gaps[batch_id]) / 1000.
            if gap > 0:
                flush()                            # nothing waits in the list while this process sleeps
                loadGenSleep(gap)
        epoch += 1
        if not tuning_batch_qps and not tuning_accel_qps:
            exp_epochs += 1

    flush()
    if gstats:
        print("[LoadGen] DRS_ENGINE_STATS %s" % dict(gstats, total=round(time.perf_counter() - t_gen0, 4),
                                                     accel_requests=accel_requests))
        sys.stdout.flush()
    # one shutdown sentinel per engine (loadGenerator.py:208-214)
    for i in range(n_cpu):
        debugPrint(args, "Load Generator", "sending done signal to " + str(i) + " cpu engine")
        requestQueue.put(None)
    for i in range(n_accel):
        debugPrint(args, "Load Generator", "sending done signal to " + str(i) + " accel engine")
        accelRequestQueue.put(None)
    loadGeneratorReturnQueue.put((cpu_sub_requests, cpu_requests, accel_requests))

    if tune_batch and not tune_accel:
        print("Scheduler's Optimal batch_size configuration: ", query_scheduler.args.sub_task_batch_size,
              " @ arrival rate of ", query_scheduler.arrival_rate, "ms")
    elif tune_batch and tune_accel:
        print("Scheduler's Optimal batch_size configuration: ", query_scheduler.args.sub_task_batch_size)
        print("Scheduler's Optimal accel_size configuration: ",
              accel_query_scheduler.args.accel_request_size_thres, " @ arrival rate of",
              accel_query_scheduler.arrival_rate, "ms")
    sys.stdout.flush()
