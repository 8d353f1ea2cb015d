#!/usr/bin/env python3
"""End-to-end serving run through the queue harness (GPU box): load generator ->
accelRequestQueue -> k accelerator engine processes -> orchestrator, i.e. the reference's
run_DeepRecSys.sh flow with real accelerator engines.  Prints the orchestrator summary."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from deeprecsys_amd.DeepRecSys import DeepRecSys
from deeprecsys_amd.utils.utils import cli

if __name__ == "__main__":
    # defaults = run_DeepRecSys.sh's query-size distribution on DLRM-RMC1 (BASELINE shape)
    base = ["--queue", "--model_accel", "--inference_engines", "0", "--num_accels", "1",
            "--model_type", "dlrm", "--arch_sparse_feature_size", "64",
            "--arch_embedding_size", "-".join(["1000000"] * 8), "--arch_mlp_bot", "128-64-64",
            "--arch_mlp_top", "256-64-1", "--arch_interaction_op", "cat",
            "--num_indices_per_lookup", "80", "--num_indices_per_lookup_fixed", "1",
            "--accel_table_init", "device", "--num_batches", "32", "--nepochs", "64",
            "--batch_size_distribution", "normal", "--avg_mini_batch_size", "165",
            "--var_mini_batch_size", "16", "--max_mini_batch_size", "256",
            "--avg_arrival_rate", "0.05", "--req_granularity", "64", "--log_file", "/tmp/drs_log/out.log"]
    argv = sys.argv[1:]
    if "--mix" in argv:
        # BASELINE config 4: W&D + NCF mixed stream (reference JSON shapes) on every accel engine
        argv.remove("--mix")
        import tempfile
        d = tempfile.mkdtemp(prefix="drs_mix_")
        wnd = {"arch_mlp_bot": "512", "arch_mlp_top": "1024-512-256-1",
               "arch_embedding_size": "-".join(["1000000"] * 27), "arch_sparse_feature_size": 32,
               "num_indices_per_lookup_fixed": True, "num_indices_per_lookup": 1,
               "arch_interaction_op": "cat", "model_type": "wnd", "model_name": "wnd"}
        ncf = {"arch_mlp_bot": "512", "arch_mlp_top": "256-256-128-64-64",
               "arch_embedding_size": "140000-140000-28000-28000", "arch_sparse_feature_size": 64,
               "num_indices_per_lookup_fixed": True, "num_indices_per_lookup": 1,
               "arch_interaction_op": "cat", "model_type": "ncf", "model_name": "ncf"}
        files = []
        for name, cfg in (("wide_and_deep", wnd), ("ncf", ncf)):
            files.append(os.path.join(d, name + ".json"))
            json.dump(cfg, open(files[-1], "w"))
        argv = ["--mix_config_files", ",".join(files)] + argv
    args = cli(base + argv)
    s = DeepRecSys(args, quiet=True)
    print(json.dumps(s))
