"""Command line of the harness: flag-for-flag the reference's cli() (utils/utils.py:15-165).

Same names, types and defaults (pinned by tests/golden/harness.json, captured by
running the reference's cli()), the same "--config_file JSON overrides the CLI"
rule with type coercion (:151-160), the same DIN table expansion (:132-149) and the
same np.random.seed(numpy_rand_seed) side effect (:163).  Flags this build adds are
listed in EXTRA_FLAGS and never collide with a reference name.
"""
import argparse
import json
import sys

import numpy as np

_S, _I, _F, _B = str, int, float, bool
STORE_TRUE = "store_true"

# (name, type | STORE_TRUE, default) in the reference's order of definition
REFERENCE_FLAGS = [
    # model
    ("arch_sparse_feature_size", _I, 2), ("arch_embedding_size", _S, "4-3-2"),
    ("arch_mlp_bot", _S, "4-3-2"), ("arch_mlp_top", _S, "4-2-1"), ("arch_mlp_tasks", _S, "4-2-1"),
    ("num_multi_tasks", _I, 1), ("hidden_size", _I, 64), ("arch_interaction_op", _S, "dot"),
    ("arch_interaction_itself", STORE_TRUE, False), ("inter_op_workers", _I, 1),
    ("sls_workers", _I, 1), ("fc_workers", _I, 1), ("model_type", _S, "dlrm"),
    ("user_behavior_tables", _I, 1000),
    # inference
    ("inference_only", STORE_TRUE, True), ("save_proto_types_shapes", STORE_TRUE, False),
    ("output_log_file", _S, None),
    # dataset
    ("num_batches", _I, 0), ("mini_batch_size", _I, 1), ("max_mini_batch_size", _I, 1),
    ("avg_mini_batch_size", _F, 1), ("var_mini_batch_size", _F, 1),
    ("batch_size_distribution", _S, "fixed"), ("batch_dist_file", _S, "config/batch_distribution.txt"),
    ("sub_task_batch_size", _I, 16), ("data_generation", _S, "random"),
    ("data_trace_file", _S, "./input/dist_emb_j.log"), ("data_set", _S, "kaggle"),
    ("raw_data_file", _S, ""), ("processed_data_file", _S, ""), ("data_randomize", _S, "total"),
    ("data_trace_enable_padding", _B, False), ("num_indices_per_lookup", _I, 10),
    ("num_indices_per_lookup_fixed", _B, False),
    # DeepRecSys
    ("queue", STORE_TRUE, False), ("inference_engines", _I, 1), ("avg_arrival_rate", _F, 10),
    ("target_latency", _F, 10), ("req_granularity", _I, 64),
    ("batch_configs", _S, "32-64-128-256-512-1024"), ("tune_batch_qps", STORE_TRUE, False),
    ("tune_accel_qps", STORE_TRUE, False), ("accel_configs", _S, "128-256-512"),
    ("stable_region", _F, 0.10), ("max_arr_range", _F, 100), ("min_arr_range", _F, 1),
    ("arr_steps", _I, 20), ("sched_timeout", _I, 100),
    # hardware
    ("use_accel", STORE_TRUE, False), ("model_accel", STORE_TRUE, False),
    ("accel_request_size_thres", _I, 1024), ("model_name", _S, ""),
    ("accel_root_dir", _S, "accelerator/"),
    # activations / loss
    ("activation_function", _S, "relu"), ("loss_function", _S, "mse"),
    ("loss_threshold", _F, 0.0), ("round_targets", _B, False),
    # training-era leftovers
    ("nepochs", _I, 1), ("learning_rate", _F, 0.01), ("print_precision", _I, 5),
    ("numpy_rand_seed", _I, 123), ("sync_dense_params", _B, True),
    ("caffe2_net_type", _S, "simple"), ("engine", _S, "TBB"),
    # debugging
    ("print_freq", _I, 1), ("print_time", STORE_TRUE, False), ("debug_mode", STORE_TRUE, False),
    ("enable_profiling", STORE_TRUE, False), ("plot_compute_graph", STORE_TRUE, False),
    ("log_file", _S, "log/output.log"),
    # experiment
    ("config_file", _S, None),
]

# additions of this build (several accelerator engines, one per GPU)
EXTRA_FLAGS = [
    ("num_accels", _I, 1),            # accelerator engine processes (one per GPU)
    ("load_generators", _I, 1),       # load generator processes (k > 1: generator g feeds the accelerator engines e % k == g)
    ("accel_backend", _S, "hip"),     # "hip": real forward on the GPU | "sim": latency table
    ("accel_device_offset", _I, 0),   # first GPU ordinal used by the accel engines
    ("accel_table_init", _S, "numpy"),  # "numpy": reference RNG stream | "device": counter-based fill
    ("accel_table_placements", _I, 12),  # places in HBM tried for the table arena at engine start (1 = wherever hipMalloc put it)
    ("accel_slots", _I, 0),           # launch sets in flight per accel engine; 0 = the engine's preference (3: gather | MLP | enqueue; MLP-bound models 6)
    ("accel_req_batch", _I, 16),      # requests per put on accelRequestQueue / responses per put back (1 = the reference's one packet per put)
    ("accel_response_blocks", _I, 0),  # > 0: an accel engine answers in ResponseBlocks of up to this many responses (columns, one put) instead of ServiceResponse packets
    ("accel_coalesce", _I, 0),        # queued requests an accel engine may serve per launch set (0 = what the engine prefers for the model: 12 | 16 | 8)
    ("mp_start_method", _S, "spawn"),  # engine/loadgen processes: spawn (HIP-safe) | fork
    # mixed-model stream (BASELINE config 4: W&D + NCF on the same accelerators): several model
    # configs served by every accel engine, each query tagged with the model it is for
    ("mix_config_files", _S, ""),     # comma-separated JSON configs (models/configs/*.json format)
    ("mix_weights", _S, ""),          # comma-separated shares of the query stream (default: equal)
]


def debugPrint(args, system_tag, message):
    if args.debug_mode:
        print("[" + str(system_tag) + "] " + str(message))
        sys.stdout.flush()


def build_parser():
    p = argparse.ArgumentParser(description="DeepRecBench")
    for name, kind, default in REFERENCE_FLAGS + EXTRA_FLAGS:
        if kind == STORE_TRUE:
            p.add_argument("--" + name, action="store_true", default=default)
        else:
            p.add_argument("--" + name, type=kind, default=default)
    return p


def _expand_din_tables(args):
    """DIN: replicate the behaviour table user_behavior_tables times (utils/utils.py:132-149)."""
    sizes = [int(x) for x in args.arch_embedding_size.split("-")]
    profile, behaviour, rest = sizes[0], sizes[1], sizes[1:]
    expanded = [profile] + [behaviour] * args.user_behavior_tables + rest
    args.arch_embedding_size = "-".join(str(s) for s in expanded)


def apply_config(args, config):
    """JSON config is the master: it overrides the CLI with the flag's own type."""
    for key, value in config.items():
        caster = type(getattr(args, key))
        setattr(args, key, caster(value))
    return args


def mix_models(args):
    """[(args_for_model_i, share_i)] of a mixed-model run, [] when --mix_config_files is unset.
    Every model's namespace is the run's own with that model's JSON applied on top (the same
    "JSON is the master" rule as --config_file, utils/utils.py:151-160 in the reference)."""
    files = [f for f in str(getattr(args, "mix_config_files", "") or "").split(",") if f]
    if not files:
        return []
    w = [float(x) for x in str(getattr(args, "mix_weights", "") or "").split(",") if x]
    if not w:
        w = [1.0] * len(files)
    if len(w) != len(files) or min(w) < 0 or sum(w) <= 0:
        raise ValueError("--mix_weights needs one non-negative share per --mix_config_files entry")
    out = []
    for f, share in zip(files, w):
        a = argparse.Namespace(**vars(args))
        with open(f, "r") as fh:
            apply_config(a, json.load(fh))
        out.append((a, share / sum(w)))
    return out


def cli(argv=None):
    args = build_parser().parse_args(argv)
    if args.model_type == "din":
        _expand_din_tables(args)
    if args.config_file:
        with open(args.config_file, "r") as f:
            apply_config(args, json.load(f))
    np.random.seed(args.numpy_rand_seed)
    return args
