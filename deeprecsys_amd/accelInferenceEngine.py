"""Accelerator inference engine: one process, one MI355X, real forward passes.

Drop-in for the reference's accelInferenceEngine (accelInferenceEngine.py:18-86): same
signature, same queue protocol --

    put True on inferenceEngineReadyQueue once ready (the load generator blocks on it,
    loadGenerator.py:76-78); loop on requestQueue.get(); None -> put None on the
    responseQueue and return; otherwise answer with one ServiceResponse that echoes the
    request and stamps process_start_time / queue_end_time / inference_end_time.

-- but where the reference looks a latency up in a GTX-1080-Ti table and sleeps
(:63-64), this engine builds the model on its GPU (weights from the same seeded numpy
stream every CPU engine uses, inferenceEngine.py:72-88), keeps all `num_batches` input
sets resident in HBM, runs the query through libdrs_hip.so and stamps
inference_end_time when the result is on the host.  Requests that are already waiting
in the queue when the engine comes back for work (up to --accel_coalesce, max 16; 0 = what the engine prefers for the model) are
served by ONE set of launches (drs_forward_multi_async): the gather then runs long
enough to amortise its start-up and tail, which is worth ~15% HBM efficiency.  Up to
--accel_slots (default: the engine's preference, 3; MLP-bound models 6) such sets are in flight at a time: the library runs the gather
of one beside the MLP of the previous one while this loop is already pulling the next
requests off the queue.  `--accel_backend sim` keeps the reference behaviour
(latency_table.py) for runs without a GPU.

Failure policy: any error is printed, the None sentinel is still sent so the
orchestrator's join loop (DeepRecSys.py:89) cannot hang, and the process exits 1.
"""
import queue as pyqueue
import os
import sys
import time

import numpy as np

from .utils.packets import ResponseBlock, ServiceResponse
from .utils.utils import debugPrint, mix_models

SIM_MODELS = ("wnd", "rm1", "rm2", "rm3", "ncf", "din", "dien", "mtwnd")   # reference omits "ncf"


def _respond(request, engine_id, start_time, end_time, out_batch_size):
    return ServiceResponse(consumer_id=engine_id, epoch=request.epoch, batch_id=request.batch_id,
                           batch_size=request.batch_size, arrival_time=request.arrival_time,
                           process_start_time=start_time, queue_end_time=end_time,
                           inference_end_time=end_time, out_batch_size=out_batch_size,
                           total_sub_batches=request.total_sub_batches,
                           exp_packet=request.exp_packet, sub_id=request.sub_id,
                           model_id=getattr(request, "model_id", 0))


def _build_hip_model(args, engine_id):
    from . import dlrm_s_hip as M
    from .data_generator.dlrm_data import DLRMDataGenerator
    first = getattr(args, "accel_first_engine_id", engine_id if engine_id is not None else 0)
    # engine k of this run -> GPU offset + k, wrapping when there are more engines than GPUs
    # (several engine processes then share a device, e.g. two engines on a one-GPU box)
    from . import _native
    ndev = max(_native.device_count(), 1)
    args._drs_device = (int(getattr(args, "accel_device_offset", 0)) + int((engine_id or 0) - first)) % ndev
    # this engine process onto the cores next to its GPU (utils/affinity.py) before anything is pinned or any worker
    # thread starts; engines that share a GPU share its cores
    from .utils import affinity
    # (more engines than GPUs: engine k drives GPU k % ndev -- two to four engine processes per GPU are what it takes to
    #  keep one MI355X busy through the Python queues, DESIGN.md 8 -- and takes that GPU's cores)
    n_acc = max(1, int(getattr(args, "num_accels", 1)))
    if not getattr(args, "_drs_bound", False):
        args._drs_binding = affinity.bind_rank(args._drs_device, min(n_acc + int(getattr(args, "accel_device_offset", 0)), ndev))
        args._drs_bound = True
    if args.model_type not in M.WRAPPERS:
        raise SystemExit("Model type %r has no accelerator path (%s)" % (args.model_type, " | ".join(sorted(M.WRAPPERS))))
    datagen = DLRMDataGenerator(args)
    nbatches, lX, lS_l, lS_i = datagen.generate_input_data()
    nbatches, lT = datagen.generate_output_data()
    model = M.WRAPPERS[args.model_type](args)
    model.create(lX[0], lS_l[0], lS_i[0], lT[0])
    model.net.stage_batches(lX, lS_l, lS_i)
    # where the table arena lands in HBM moves the gather by a few per cent for the engine's lifetime: try a few places
    # with the model's own launch sets (about 30 ms each) and keep the fastest (--accel_table_placements, 1 = off)
    n_place = int(getattr(args, "accel_table_placements", 12))
    if n_place > 1:
        model.net.tune_table_placement(n_place)
    return model


def accelInferenceEngine(args, requestQueue=None, engine_id=None, responseQueue=None,
                         inferenceEngineReadyQueue=None):
    np.random.seed(args.numpy_rand_seed)
    np.set_printoptions(precision=args.print_precision)
    if requestQueue is None:
        print("If you want to run Accel in isolation please use bench.py / the model classes directly")
        sys.stdout.flush()
        sys.exit()

    backend = getattr(args, "accel_backend", "hip")
    model = accel_data = None
    try:
        if backend == "sim":
            from .latency_table import GPU_Data, predict_time
            if args.model_name not in SIM_MODELS:
                print("Model not found in ones supported")
                raise SystemExit(1)
            accel_data = GPU_Data(root_dir=args.accel_root_dir, hardware="nvidia_gtx_1080_ti")
        else:
            # mixed-model stream: one resident model per --mix_config_files entry, all on this GPU
            mix = [a for a, _share in mix_models(args)]
            models = []
            for a in (mix or [args]):
                a.accel_first_engine_id = getattr(args, "accel_first_engine_id", engine_id or 0)
                np.random.seed(args.numpy_rand_seed)      # every model: the stream a lone engine would see
                models.append(_build_hip_model(a, engine_id))
            model = models[0]
    except BaseException as e:   # incl. SystemExit from the builders' sys.exit checks
        print("[Accel %s] start-up failed: %r" % (engine_id, e))
        sys.stdout.flush()
        inferenceEngineReadyQueue.put(True)      # unblock the load generator ...
        _drain_until_sentinel(requestQueue)      # ... swallow the work routed to us ...
        responseQueue.put(None)                  # ... and let the orchestrator join
        sys.exit(1)

    inferenceEngineReadyQueue.put(True)
    # --accel_coalesce n: up to n (<= 16) queued requests per launch set; 0: what the engine prefers for the
    # model (drs_get_option "preferred_coalesce": 12 for gather-bound DLRM, 16 for the MLP-bound models, whose
    # 16-row MLP workgroups then cover all 256 CUs, 8 otherwise)
    coalesce = 1
    if model is not None:
        want = int(getattr(args, "accel_coalesce", 0))
        coalesce = max(1, min(want, 16)) if want > 0 else min(m.net.engine.get_option("preferred_coalesce") for m in models)
    n_slots = model.net.engine.num_slots if model is not None else 1
    n_models = len(models) if model is not None else 1
    free = [list(range(n_slots)) for _ in range(n_models)]   # per model: slots with nothing in flight
    inflight = []                        # [(model_id, slot, requests, start_time)], oldest first
    backlog = []                         # pulled but not yet submitted (other model's turn / no free slot)
    shutdown = False

    def fail(requests, e):
        print("[Accel %s] request(s) %s failed: %r" % (
            engine_id, [(r.batch_id, r.batch_size) for r in requests], e))
        sys.stdout.flush()
        if not shutdown:
            _drain_until_sentinel(requestQueue)
        responseQueue.put(None)
        sys.exit(1)

    stats = {"pull": 0.0, "submit": 0.0, "collect": 0.0, "respond": 0.0, "sets": 0, "queries": 0} \
        if os.environ.get("DRS_ENGINE_STATS") else None

    block_n = int(getattr(args, "accel_response_blocks", 0)) if model is not None else 0
    blk = [[] for _ in range(8)]

    def flush_block():
        if blk[0]:
            responseQueue.put(ResponseBlock(engine_id, *blk))
            for col in blk:
                del col[:]

    def finish_oldest():
        mid, slot, requests, start_time = inflight.pop(0)
        t_c = time.perf_counter() if stats else 0.0
        try:
            outs = models[mid].net.collect_staged_multi([r.batch_size for r in requests], slot)
        except Exception as e:
            fail(requests, e)
        end_time = time.time()
        if stats:
            stats["collect"] += time.perf_counter() - t_c
            t_c = time.perf_counter()
        free[mid].append(slot)
        # the responses of a launch set leave in one put (a list) unless --accel_req_batch 1 asks for the
        # reference's one packet per put; the packets themselves are the reference's (utils/packets.py:32-59)
        if block_n > 0:
            # --accel_response_blocks: the set's responses join the engine's open block (columns); it leaves when full
            # -- or, at the latest, when the engine finds nothing more to do (flush_block below)
            for r in requests:
                blk[0].append(r.epoch); blk[1].append(r.batch_id); blk[2].append(r.batch_size); blk[3].append(r.arrival_time)
                blk[4].append(start_time); blk[5].append(end_time); blk[6].append(bool(r.exp_packet))
                blk[7].append(getattr(r, "model_id", 0))
            if len(blk[0]) >= block_n:
                flush_block()
            resp = requests
        else:
            resp = [_respond(r, engine_id, start_time, end_time, o.shape[0]) for r, o in zip(requests, outs)]
            if batched and len(resp) > 1:
                responseQueue.put(resp)
            else:
                for x in resp:
                    responseQueue.put(x)
        if stats:
            stats["respond"] += time.perf_counter() - t_c
            stats["sets"] += 1
            stats["queries"] += len(resp)

    def model_of(r):
        mid = int(getattr(r, "model_id", 0) or 0)
        if not 0 <= mid < n_models:
            fail([r], ValueError("request for model %d, engine serves %d model(s)" % (mid, n_models)))
        return mid

    batched = int(getattr(args, "accel_req_batch", 1)) > 1

    def take(item):
        # one put may carry a LIST of requests (loadGenerator, --accel_req_batch): its members are
        # requests that were already waiting
        if isinstance(item, list):
            backlog.extend(item)
        else:
            backlog.append(item)

    while not shutdown or inflight or backlog:
        # 1. pull: block only when the GPU has nothing to do; otherwise take what is already there
        t_p = time.perf_counter() if stats else 0.0
        if not shutdown and len(backlog) < coalesce:
            debugPrint(args, "Accel", "Trying to pull request")
            try:
                if block_n > 0 and not (inflight or backlog):
                    flush_block()            # about to sleep on the queue: what has been answered leaves first
                take(requestQueue.get() if not (inflight or backlog) else requestQueue.get_nowait())
            except pyqueue.Empty:
                pass
            # requests that are ALREADY waiting ride along in the same set of launches
            while backlog and backlog[-1] is not None and len(backlog) < coalesce:
                try:
                    take(requestQueue.get_nowait())
                except pyqueue.Empty:
                    break
            if backlog and backlog[-1] is None:
                shutdown = True
                backlog.pop()
        if stats:
            stats["pull"] += time.perf_counter() - t_p
            t_p = time.perf_counter()
        # 2. submit: the oldest waiting request picks the model; same-model requests behind it join
        submitted = False
        if backlog:
            mid = model_of(backlog[0]) if model is not None else 0
            if model is None:
                # reference behaviour: one request, one table lookup, one sleep
                r = backlog.pop(0)
                start_time = time.time()
                try:
                    time.sleep(predict_time(args.model_name, r.batch_size, accel_data) / 1000.)
                except Exception as e:
                    fail([r], e)
                end_time = time.time()
                responseQueue.put(_respond(r, engine_id, start_time, end_time, r.batch_size))
                submitted = True
            elif free[mid]:
                requests = [r for r in backlog if model_of(r) == mid][:coalesce]
                for r in requests:
                    backlog.remove(r)
                start_time = time.time()
                slot = free[mid].pop()
                try:
                    models[mid].net.submit_staged_multi([r.batch_id for r in requests],
                                                        [r.batch_size for r in requests], slot)
                except Exception as e:
                    fail(requests, e)
                inflight.append((mid, slot, requests, start_time))
                submitted = True
        if stats:
            stats["submit"] += time.perf_counter() - t_p
        # 3. nothing new could be started: retire the oldest set in flight
        if not submitted and inflight:
            finish_oldest()
    if stats:
        print("[Accel %s] DRS_ENGINE_STATS %s" % (engine_id, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in stats.items()}))
        sys.stdout.flush()
    flush_block()
    debugPrint(args, "Accel", "Sending final done signal")
    responseQueue.put(None)
    if model is not None:
        for m in models:
            m.net.engine.close()


def _drain_until_sentinel(q):
    while q.get() is not None:
        pass
