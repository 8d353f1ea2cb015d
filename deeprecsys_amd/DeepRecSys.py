"""Orchestrator: spawns the load generator and the inference engines, reassembles
responses, feeds the running tail latency back to the scheduler, reports
latency-bounded throughput.

Mirror of the reference's DeepRecSys.py:21-185 (same queues, same process layout, same
printed summary lines) with the extension of SURVEY.md 8(e): `--num_accels k` spawns k
accelerator engines, one per GPU, all consuming the shared accelRequestQueue
(the reference hard-wires one, DeepRecSys.py:38-39,63).

This package ships no CPU engine: the CPU forward lives in oracle/ as test
infrastructure.  Callers that want CPU engines in the mix (bench.py's cpu baseline
does) pass `cpu_engine=<callable with the inferenceEngine signature>`.
"""
import os
import sys
import multiprocessing

import numpy as np

from .accelInferenceEngine import accelInferenceEngine
from .loadGenerator import accel_engine_count, loadGenerator
from .stats import ResponseAggregator
from .utils.packets import ResponseBlock
from .utils.utils import cli, mix_models


def DeepRecSys(args=None, cpu_engine=None, quiet=False):
    say = (lambda *a: None) if quiet else print
    say("Running DeepRecSys")
    if args is None:
        args = cli()
    say("============================================================")
    say("DeepRecSys configuration")
    for key in vars(args):
        say(key, getattr(args, key))
    say("============================================================")
    if not args.queue:
        # reference DeepRecSys.py:184-185: "No queue, run DeepRecSys in standalone mode" -> inferenceEngine(args), whose
        # queue-less branch times nepochs passes over the generated batches (inferenceEngine.py:137-173); here the same
        # loop on the accelerator (this package has no CPU forward)
        from .dlrm_s_hip import standalone
        return standalone(args)

    n_accel = accel_engine_count(args)
    n_cpu = int(args.inference_engines)
    if n_cpu > 0 and cpu_engine is None:
        sys.exit("ERROR: --inference_engines %d CPU engines requested, but deeprecsys_amd has no "
                 "CPU forward (by design); use --inference_engines 0 --model_accel" % n_cpu)
    if n_cpu + n_accel == 0:
        sys.exit("ERROR: no inference engines configured")
    args.inference_engines = n_cpu + n_accel        # reference: += 1 when model_accel (:38-39)
    args.accel_first_engine_id = n_cpu              # engines [n_cpu, n_cpu + n_accel) are accelerators
    say("[DeepRecSys] total inference engine ", args.inference_engines)

    # "spawn", not the reference's implicit fork: a forked child of a process that has
    # already touched the HIP runtime (a test runner, a notebook) hangs in its first HIP call
    ctx = multiprocessing.get_context(getattr(args, "mp_start_method", "spawn"))
    Process, Queue = ctx.Process, ctx.Queue
    requestQueue = Queue(maxsize=1024)
    # --load_generators k: k generator processes, each with its own accelerator queue; accelerator engine e listens
    # to queue e % k (k = 1: the reference's one generator and one shared accelRequestQueue)
    n_gen = max(1, int(getattr(args, "load_generators", 1)))
    if n_gen > 1:
        if n_cpu > 0 or args.tune_batch_qps or args.tune_accel_qps:
            sys.exit("ERROR: --load_generators > 1 serves accelerator engines at a fixed arrival rate "
                     "(no CPU engines, no scheduler sweep)")
        n_gen = min(n_gen, n_accel)
    accelRequestQueues = [Queue(maxsize=32 * max(-(-n_accel // n_gen), 1)) for _ in range(n_gen)]
    accelRequestQueue = accelRequestQueues[0]
    pidQueue = Queue()
    inferenceEngineReadyQueue = Queue()
    loadGeneratorReturnQueue = Queue()
    responseQueues = [Queue() for _ in range(args.inference_engines)]

    load_gens = []
    gen_ready = [inferenceEngineReadyQueue] if n_gen == 1 else [Queue() for _ in range(n_gen)]
    for g in range(n_gen):
        ga = args
        if n_gen > 1:
            import copy
            ga = copy.copy(args)
            ga._gen_shard = (g, n_gen)
        load_gens.append(Process(target=loadGenerator,
                                 args=(ga, requestQueue, loadGeneratorReturnQueue, gen_ready[g],
                                       pidQueue, accelRequestQueues[g])))
    load_gen = load_gens[0]
    engines = []
    for i in range(args.inference_engines):
        if i >= n_cpu:
            p = Process(target=accelInferenceEngine,
                        args=(args, accelRequestQueues[(i - n_cpu) % n_gen], i, responseQueues[i], inferenceEngineReadyQueue))
        else:
            p = Process(target=cpu_engine,
                        args=(args, requestQueue, i, responseQueues[i], inferenceEngineReadyQueue))
        p.daemon = True
        engines.append(p)
    for p in engines:
        p.start()
    for lg in load_gens:
        lg.start()
    if n_gen > 1:
        # every engine announces itself once (the reference's protocol): relay the tokens to each generator
        import threading

        def relay():
            for _ in range(args.inference_engines):
                inferenceEngineReadyQueue.get()
            for q in gen_ready:
                for _ in range(args.inference_engines):
                    q.put(True)
        threading.Thread(target=relay, daemon=True).start()

    agg = ResponseAggregator(args.req_granularity, with_model=bool(mix_models(args)))
    finished = 0
    while finished != args.inference_engines:
        for q in responseQueues:
            if q.qsize():
                item = q.get()
                if item is None:
                    finished += 1
                    say("Joined ", finished, " inference engines")
                    sys.stdout.flush()
                    continue
                if isinstance(item, ResponseBlock):
                    # --accel_response_blocks: many whole-query responses of one engine as columns, booked at once
                    running_p95 = agg.add_block(item)
                    if running_p95 is not None:
                        pidQueue.put(running_p95)
                    continue
                # an accelerator engine returns the responses of one launch set in one put (a list)
                for response in (item if isinstance(item, list) else (item,)):
                    _latency, running_p95 = agg.add(response)
                    if running_p95 is not None:
                        say("Running latency: ", running_p95)
                        sys.stdout.flush()
                        pidQueue.put(running_p95)
    say("Finished runing over the inference engines")

    log_dir = os.path.dirname(args.log_file)
    if log_dir and not os.path.exists(log_dir):
        os.makedirs(log_dir)
    with open(args.log_file, "w") as f:
        for r in agg.responses_list:
            f.write(str(r) + "\n")

    cpu_sub_requests = cpu_requests = accel_requests = 0
    for lg in load_gens:
        a_, b_, c_ = loadGeneratorReturnQueue.get()
        cpu_sub_requests, cpu_requests, accel_requests = cpu_sub_requests + a_, cpu_requests + b_, accel_requests + c_
    for lg in load_gens:
        lg.join()
    agg_requests = cpu_sub_requests + accel_requests
    say("Exiting DeepRecSys after printing ", len(agg.responses_list), "/", agg_requests)
    say("CPU sub requests ", cpu_sub_requests, "/", agg_requests)
    say("CPU requests ", cpu_requests)
    say("Accel requests ", accel_requests, "/", agg_requests)
    s = agg.summary()
    say("Measured QPS: ", s["qps"])
    say("Measured p95 tail-latency: ", s["p95_ms"], " ms")
    say("Measured p99 tail-latency: ", s["p99_ms"], " ms")
    sys.stdout.flush()
    for p in engines:
        p.terminate()
    s.update(cpu_sub_requests=cpu_sub_requests, cpu_requests=cpu_requests, accel_requests=accel_requests)
    return s


if __name__ == "__main__":
    DeepRecSys()
