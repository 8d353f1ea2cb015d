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
               "ncf_mini"]
NET_CLS = {"dlrm": M.DLRM_Net, "wnd": M.Wide_and_Deep, "ncf": M.NCF}


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def load_fixture(case):
    with open(os.path.join(GOLDEN, case + ".json")) as f:
        meta = json.load(f)
    arrays = np.load(os.path.join(GOLDEN, case + ".npz"))
    return meta, arrays


def args_from(meta_args, **override):
    args = cli([])
    for k, v in meta_args.items():
        setattr(args, k, v)
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
