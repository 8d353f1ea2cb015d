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
in the queue when the engine comes back for work (up to --accel_coalesce, max 8) are
served by ONE set of launches (drs_forward_multi_async): the gather then runs long
enough to amortise its start-up and tail, which is worth ~15% HBM efficiency.  Up to
--accel_slots (default 3) such sets are in flight at a time: the library runs the gather
of one beside the MLP of the previous one while this loop is already pulling the next
requests off the queue.  `--accel_backend sim` keeps the reference behaviour
(latency_table.py) for runs without a GPU.

Failure policy: any error is printed, the None sentinel is still sent so the
orchestrator's join loop (DeepRecSys.py:89) cannot hang, and the process exits 1.
"""
import queue as pyqueue
import sys
import time

import numpy as np

from .utils.packets import ServiceResponse
from .utils.utils import debugPrint

SIM_MODELS = ("wnd", "rm1", "rm2", "rm3", "ncf", "din", "dien", "mtwnd")   # reference omits "ncf"


def _respond(request, engine_id, start_time, end_time, out_batch_size):
    return ServiceResponse(consumer_id=engine_id, epoch=request.epoch, batch_id=request.batch_id,
                           batch_size=request.batch_size, arrival_time=request.arrival_time,
                           process_start_time=start_time, queue_end_time=end_time,
                           inference_end_time=end_time, out_batch_size=out_batch_size,
                           total_sub_batches=request.total_sub_batches,
                           exp_packet=request.exp_packet, sub_id=request.sub_id)


def _build_hip_model(args, engine_id):
    from . import dlrm_s_hip as M
    from .data_generator.dlrm_data import DLRMDataGenerator
    first = getattr(args, "accel_first_engine_id", engine_id if engine_id is not None else 0)
    args._drs_device = int(getattr(args, "accel_device_offset", 0)) + int((engine_id or 0) - first)
    if args.model_type not in M.WRAPPERS:
        raise SystemExit("Model type %r has no accelerator path (dlrm | wnd | ncf)" % args.model_type)
    datagen = DLRMDataGenerator(args)
    nbatches, lX, lS_l, lS_i = datagen.generate_input_data()
    nbatches, lT = datagen.generate_output_data()
    model = M.WRAPPERS[args.model_type](args)
    model.create(lX[0], lS_l[0], lS_i[0], lT[0])
    model.net.stage_batches(lX, lS_l, lS_i)
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
            model = _build_hip_model(args, engine_id)
    except BaseException as e:   # incl. SystemExit from the builders' sys.exit checks
        print("[Accel %s] start-up failed: %r" % (engine_id, e))
        sys.stdout.flush()
        inferenceEngineReadyQueue.put(True)      # unblock the load generator ...
        _drain_until_sentinel(requestQueue)      # ... swallow the work routed to us ...
        responseQueue.put(None)                  # ... and let the orchestrator join
        sys.exit(1)

    inferenceEngineReadyQueue.put(True)
    coalesce = max(1, min(int(getattr(args, "accel_coalesce", 8)), 8)) if model is not None else 1
    n_slots = model.net.engine.num_slots if model is not None else 1
    free = list(range(n_slots))          # launch-set slots with nothing in flight
    inflight = []                        # [(slot, requests, start_time)], oldest first
    shutdown = False

    def fail(requests, e):
        print("[Accel %s] request(s) %s failed: %r" % (
            engine_id, [(r.batch_id, r.batch_size) for r in requests], e))
        sys.stdout.flush()
        if not shutdown:
            _drain_until_sentinel(requestQueue)
        responseQueue.put(None)
        sys.exit(1)

    def finish_oldest():
        slot, requests, start_time = inflight.pop(0)
        try:
            outs = model.net.collect_staged_multi([r.batch_size for r in requests], slot)
        except Exception as e:
            fail(requests, e)
        end_time = time.time()
        free.append(slot)
        for r, o in zip(requests, outs):
            responseQueue.put(_respond(r, engine_id, start_time, end_time, o.shape[0]))

    while not shutdown or inflight:
        requests = []
        if not shutdown and free:
            debugPrint(args, "Accel", "Trying to pull request")
            # block only when the GPU has nothing to do; otherwise take what is already there
            try:
                requests.append(requestQueue.get() if not inflight else requestQueue.get_nowait())
            except pyqueue.Empty:
                pass
            # requests that are ALREADY waiting ride along in the same set of launches
            while requests and requests[-1] is not None and len(requests) < coalesce:
                try:
                    requests.append(requestQueue.get_nowait())
                except pyqueue.Empty:
                    break
            if requests and requests[-1] is None:
                shutdown = True
                requests.pop()
        if requests:
            start_time = time.time()
            if model is not None:
                slot = free.pop()
                try:
                    model.net.submit_staged_multi([r.batch_id for r in requests],
                                                  [r.batch_size for r in requests], slot)
                except Exception as e:
                    fail(requests, e)
                inflight.append((slot, requests, start_time))
            else:
                # reference behaviour: one request, one table lookup, one sleep
                try:
                    time.sleep(predict_time(args.model_name, requests[0].batch_size, accel_data) / 1000.)
                except Exception as e:
                    fail(requests, e)
                end_time = time.time()
                responseQueue.put(_respond(requests[0], engine_id, start_time, end_time, requests[0].batch_size))
        elif inflight:
            finish_oldest()              # nothing new (or every slot busy): retire the oldest set
    debugPrint(args, "Accel", "Sending final done signal")
    responseQueue.put(None)
    if model is not None:
        model.net.engine.close()


def _drain_until_sentinel(q):
    while q.get() is not None:
        pass
