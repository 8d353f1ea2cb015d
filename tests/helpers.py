"""Shared test plumbing: golden fixtures, and building the SAME model twice --
once as the product object (deeprecsys_amd, HIP) and once as the oracle (CPU)."""
import hashlib
import json
import os

import numpy as np

from deeprecsys_amd import dlrm_s_hip as M
from deeprecsys_amd.data_generator.dlrm_data import DLRMDataGenerator
from deeprecsys_amd.utils.utils import cli
from oracle import oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MODEL_CASES = ["dlrm_dot_small", "dlrm_dot_itself_small", "dlrm_cat_small", "dlrm_cat_queue_small",
               "dlrm_dot_queue_small", "dlrm_rm1_mini", "dlrm_rm2_mini", "dlrm_rm3_mini", "wnd_mini",
               "ncf_mini", "mtwnd_mini", "din_mini", "dien_mini"]
NET_CLS = {"dlrm": M.DLRM_Net, "wnd": M.Wide_and_Deep, "ncf": M.NCF, "mtwnd": M.MT_Wide_and_Deep,
           "din": M.DIN_Net, "dien": M.DIEN_Net}
NO_DENSE = ("ncf", "din", "dien")     # model types whose queries are sparse features only


LAB = M.N.is_lab()        # the lab build is bound (DRS_TEST_LAB=1, tests/conftest.py): entries that need its options run
# what the lab build accepts beyond the product (docs/OPTIONS.md): whole keys, and values of two product keys
_LAB_KEYS = {"mlp_early", "small_piped", "mlp_layout", "mlp_preload", "mlp_kc", "launch_thread", "mlp_debug", "mlp_small_rows",
             "mlp_fuse_rows", "sls_short_bag", "sls_uniform", "mlp_gemm", "mlp_gemm_2cu", "mlp_gemm32", "mlp_gemm32_small",
             "mlp_gemm32_small_blocks", "mlp_gemm32_blocks", "zero_copy", "table_vmm_chunk", "table_vmm_align"}


def runs_here(opts):
    """does the bound library take this option dict?  (product: no lab keys, mlp_stream in {2, 4}, zero_copy_inputs >= 1)"""
    if LAB:
        return True
    return not (set(opts) & _LAB_KEYS or opts.get("mlp_stream", 2) in (0, 1) or opts.get("zero_copy_inputs", 1) == 0)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def load_fixture(case):
    with open(os.path.join(GOLDEN, case + ".json")) as f:
        meta = json.load(f)
    arrays = np.load(os.path.join(GOLDEN, case + ".npz"))
    return meta, arrays


def golden_output(meta, z):
    """The model's output as the fixtures hold it: `prob_click`, or for MT-WnD the task heads'
    last blobs side by side (the layout the engine returns)."""
    if meta.get("output_blobs"):
        return np.concatenate([z["expected/" + b] for b in meta["output_blobs"]], axis=1)
    return z["expected/prob_click"]


def args_from(meta_args, **override):
    args = cli([])
    for k, v in meta_args.items():
        setattr(args, k, v)
    if meta_args.get("model_type") == "dien":
        args.dien_rnn_init = "fed"      # the fixtures hold the recurrent weights models/dien.py feeds
    for k, v in override.items():
        setattr(args, k, v)
    return args


def materialize(args):
    """Engine start-up order of the reference (inferenceEngine.py:72-88): seed, inputs,
    targets, then the model's weights -- all from one numpy stream."""
    np.random.seed(args.numpy_rand_seed)
    gen = DLRMDataGenerator(args)
    nb, lX, lS_l, lS_i = gen.generate_input_data()
    nb, lT = gen.generate_output_data()
    net = NET_CLS[args.model_type](args)
    lS_l = [[np.asarray(l, dtype=np.int32) for l in per] for per in lS_l]
    lS_i = [[np.asarray(i, dtype=np.int64) for i in per] for per in lS_i]
    return net, lX, lS_l, lS_i, lT


def oracle_model(net):
    """The oracle twin of a host-side net object (same weights)."""
    if net.kind == M.N.MODEL_NCF:
        return orc.Model(orc.MODEL_NCF, net.emb_w, [0], [], net.ln_top[:-1], net.top_w,
                         final=net.final_w[0])
    if net.kind == M.N.MODEL_DIN:
        return orc.Model(orc.MODEL_DIN, net.emb_w, [0], [], net.ln_top, net.top_w, ln_att=net.ln_att,
                         att=net.att_w)
    if net.kind == M.N.MODEL_DIEN:
        rnn = [a for layer in net.rnn_w for a in (layer[0][0], layer[0][1], layer[1][0], layer[1][1])]
        return orc.Model(orc.MODEL_DIEN, net.emb_w, [0], [], net.ln_top, net.top_w, rnn=rnn)
    if net.kind == M.N.MODEL_MTWND:
        return orc.Model(orc.MODEL_MTWND, net.emb_w, net.ln_bot, [], net.ln_top, net.top_w,
                         sigmoid_top=net.sigmoid_top, ln_task=net.ln_task, tasks=net.task_w)
    if net.kind == M.N.MODEL_WND:
        return orc.Model(orc.MODEL_WND, net.emb_w, net.ln_bot, [], net.ln_top, net.top_w,
                         sigmoid_top=net.sigmoid_top)
    op = orc.INTERACT_DOT if net.arch_interaction_op == "dot" else orc.INTERACT_CAT
    return orc.Model(orc.MODEL_DLRM, net.emb_w, net.ln_bot, net.bot_w, net.ln_top, net.top_w,
                     interaction_op=op, itself=net.arch_interaction_itself,
                     sigmoid_top=net.sigmoid_top)


def close(a, b, rtol, atol=0.0, atol_scale=0.0):
    """|a-b| <= atol + atol_scale*max|b| + rtol*|b| elementwise.  atol_scale covers
    near-zero elements produced by cancellation in long fp32 dot products."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    floor = atol + (atol_scale * np.abs(b).max() if b.size else 0.0)
    return bool(np.all(np.abs(a - b) <= floor + rtol * np.abs(b)))


# tolerance BASELINE.json's north_star states for fp32 MLP outputs
RTOL_OUT = 1e-4


def oracle_inference_engine(args, requestQueue=None, engine_id=None, responseQueue=None,
                            inferenceEngineReadyQueue=None):
    """CPU inference engine for harness tests: the reference's inferenceEngine()
    (inferenceEngine.py:62-249) with the CPU oracle in place of Caffe2 -- same start-up
    order (seed, inputs, weights), same ready token, same per-request prefix slicing
    (:200-215), same response stamping, None sentinel in and out.  TEST INFRASTRUCTURE:
    DeepRecSys(cpu_engine=oracle_inference_engine) puts real CPU engines beside the
    accelerator engine, e.g. to run the scheduler against measured latencies."""
    import time

    from deeprecsys_amd.utils.packets import ServiceResponse
    net, lX, lS_l, lS_i, lT = materialize(args)
    om = oracle_model(net)
    inferenceEngineReadyQueue.put(True)
    while True:
        request = requestQueue.get()
        if request is None:
            responseQueue.put(None)
            return
        start = time.time()
        bid, bs = request.batch_id, request.batch_size
        dense = None if args.model_type in NO_DENSE else lX[bid]
        out = om.forward(dense, lS_i[bid], lS_l[bid], bs=bs, nthreads=1)
        end = time.time()
        responseQueue.put(ServiceResponse(consumer_id=engine_id, epoch=request.epoch, batch_id=bid,
                                          batch_size=bs, arrival_time=request.arrival_time,
                                          process_start_time=start, queue_end_time=end,
                                          inference_end_time=end, out_batch_size=out.shape[0],
                                          total_sub_batches=request.total_sub_batches,
                                          exp_packet=request.exp_packet, sub_id=request.sub_id))
