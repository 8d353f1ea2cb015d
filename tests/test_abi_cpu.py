"""CPU suite for everything ABOVE the C ABI: the ctypes binding, the model wrappers that
mirror the reference's classes, and the accelerator engine's request loop -- driven through
oracle/libdrs_cpu.so, the CPU restatement of include/drs.h (`cpu_abi` fixture, tests only).
The HIP library itself is covered by tests/test_gpu_parity.py on the GPU box."""
import queue
import re

import numpy as np
import pytest

from deeprecsys_amd import _native as N
from tests import helpers as H


def test_cpu_abi_restates_every_entry_point_of_the_header(cpu_abi):
    import os
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "drs.h")).read()
    declared = set(re.findall(r"\b(drs_[a-z0-9_]+)\s*\(", hdr))
    bound = {name for name, _, _ in N.SYMBOLS}
    assert declared == bound, declared ^ bound          # the binding covers the whole header ...
    for name in declared:
        assert hasattr(cpu_abi, name), name              # ... and so does the CPU restatement
    assert N.device_count() == 1 and cpu_abi.drs_abi_version() == 5 and cpu_abi.drs_backend() == b"cpu:oracle"


@pytest.mark.parametrize("case", H.MODEL_CASES)
def test_wrappers_through_the_abi_match_oracle_and_golden(cpu_abi, case):
    """materialize -> create -> stage_batches -> run_staged / run / run_queues: every wrapper
    call lands on the ABI with the right arrays in the right order (tables, bottom, top,
    final; int64 ids; prefix slicing), so the outputs equal the oracle called directly."""
    meta, z = H.load_fixture(case)
    args = H.args_from(meta["args"])
    net, lX, lS_l, lS_i, lT = H.materialize(args)
    om = H.oracle_model(net)
    ncf = args.model_type in H.NO_DENSE
    net.create(lX[0], lS_l[0], lS_i[0], lT[0])
    try:
        net.stage_batches(None if ncf else lX, lS_l, lS_i)
        n = len(lS_l[0][0])
        for bid in range(len(lS_l)):
            for bs in sorted({n, 1, max(1, n // 2)}):
                got = net.run_staged(bid, bs)
                exp, R_exp = om.forward(None if ncf else lX[bid], lS_i[bid], lS_l[bid], bs=bs, want_R=True)
                assert np.array_equal(got, exp), (case, bid, bs)
                assert np.array_equal(net.engine.fetch_interaction(bs), R_exp)
        assert H.close(net.run_staged(0, n), H.golden_output(meta, z), rtol=H.RTOL_OUT)
        # several queries in one call, results back to back
        outs = net.run_staged_multi([0, len(lS_l) - 1, 0], [n, 1, max(1, n // 2)])
        assert [o.shape[0] for o in outs] == [n, 1, max(1, n // 2)]
        assert np.array_equal(outs[1], net.run_staged(len(lS_l) - 1, 1))
        # several waiting requests' arrays as one launch set (drs_run_queues_multi_async)
        L_ = int(args.num_indices_per_lookup)
        reqs = []
        for bid, bs in ((0, n), (len(lS_l) - 1, 1), (0, max(1, n // 2))):
            ids2 = np.stack([np.asarray(i, dtype=np.int64) for i in lS_i[bid]])
            len2 = np.stack([np.asarray(l, dtype=np.int32) for l in lS_l[bid]])
            fc = None if ncf else np.asarray(lX[bid], dtype=np.float32)[:bs]
            reqs.append((ids2[:, :bs * L_], len2[:, :bs], fc, bs))
        outs_q = net.run_queued_multi(reqs)
        assert all(np.array_equal(a_, b_) for a_, b_ in zip(outs_q, outs))
        # the reference's stand-alone run(X, S_lengths, S_indices) signature: per-call host inputs
        net.run(None if ncf else lX[0], lS_l[0], lS_i[0])
        assert np.array_equal(net.fetch_output(), net.run_staged(0, n))
    finally:
        net.engine.close()


def test_abi_status_codes_surface_as_DrsError(cpu_abi):
    e = N.Engine(N.MODEL_DLRM, [50, 60], 8, [4, 8], [24, 4, 1], N.INTERACT_CAT, sigmoid_top=2,
                 max_batch=4, max_lookups=3, num_staged_batches=2, num_slots=2)
    try:
        rng = np.random.RandomState(0)
        idx = [rng.randint(0, 50, 8).astype(np.int64), rng.randint(0, 60, 8).astype(np.int64)]
        lens = [np.full(4, 2, np.int32), np.full(4, 2, np.int32)]
        dense = rng.rand(4, 4).astype(np.float32)
        with pytest.raises(N.DrsError) as ei:           # nothing staged yet
            e.forward(0, 1)
        assert ei.value.code == N.ERR_STATE
        e.stage_batch(0, dense, idx, lens)
        with pytest.raises(N.DrsError) as ei:           # tables / weights missing
            e.forward(0, 1)
        assert ei.value.code == N.ERR_STATE
        bad = [idx[0].copy(), idx[1].copy()]
        bad[1][3] = 60                                  # Caffe2 ENFORCE: index < rows
        with pytest.raises(N.DrsError) as ei:
            e.stage_batch(1, dense, bad, lens)
        assert ei.value.code == N.ERR_INDEX_RANGE
        with pytest.raises(N.DrsError) as ei:           # Caffe2 ENFORCE: sum(lengths) == len(indices)
            e.stage_batch(1, dense, idx, [np.full(4, 2, np.int32), np.array([2, 2, 2, 1], np.int32)])
        assert ei.value.code == N.ERR_LENGTHS_SUM
        for t, r in enumerate((50, 60)):
            e.set_table(t, rng.rand(r, 8).astype(np.float32))
        e.set_fc(N.MLP_BOT, 0, rng.rand(8, 4).astype(np.float32), rng.rand(8).astype(np.float32))
        e.set_fc(N.MLP_TOP, 0, rng.rand(4, 24).astype(np.float32), rng.rand(4).astype(np.float32))
        e.set_fc(N.MLP_TOP, 1, rng.rand(1, 4).astype(np.float32), rng.rand(1).astype(np.float32))
        with pytest.raises(N.DrsError) as ei:           # a layer of the wrong shape
            e.set_fc(N.MLP_TOP, 1, rng.rand(2, 4).astype(np.float32), rng.rand(2).astype(np.float32))
        assert ei.value.code == N.ERR_BAD_ARG
        assert e.forward(0, 4).shape == (4, 1) and e.forward(0, 0).shape == (0, 1)
        with pytest.raises(N.DrsError) as ei:           # more samples than the staged batch holds
            e.forward(0, 5)
        assert ei.value.code == N.ERR_BAD_ARG
        with pytest.raises(N.DrsError):
            e.forward_multi_async(0, [0] * 17, [1] * 17)  # DRS_MAX_COALESCE
        assert e.gather_bytes(0, 4) == 4 * 2 * (2 * 8 * 4 + 2 * 4 + 4 + 8 * 4)
    finally:
        e.close()
    with pytest.raises(N.DrsError) as ei:               # the reference's sys.exit shape checks
        N.Engine(N.MODEL_DLRM, [50], 8, [4, 6], [16, 1], N.INTERACT_CAT, sigmoid_top=1, max_batch=4,
                 max_lookups=3, num_staged_batches=1, num_slots=1)
    assert ei.value.code == N.ERR_BAD_ARG


def _engine_args(tmp_path, **kw):
    from deeprecsys_amd.utils.utils import cli
    a = cli(["--queue", "--model_accel", "--inference_engines", "0", "--num_batches", "4", "--nepochs", "1",
             "--max_mini_batch_size", "16", "--arch_sparse_feature_size", "8", "--arch_embedding_size", "40-50-60",
             "--arch_mlp_bot", "5-16-8", "--arch_mlp_top", "12-1", "--arch_interaction_op", "dot",
             "--num_indices_per_lookup", "4", "--model_type", "dlrm", "--log_file", str(tmp_path / "o.log")])
    for k, v in kw.items():
        setattr(a, k, v)
    return a


@pytest.mark.parametrize("req_batch", [1, 16])
def test_accel_engine_request_loop_in_process(cpu_abi, tmp_path, req_batch):
    """The accelerator engine's loop (ready token, coalescing of queued requests, several launch
    sets in flight, responses in request order per set, None sentinel) against in-process queues,
    with the real model wrappers on the CPU ABI; outputs sizes and response fields as the
    reference's engine stamps them (accelInferenceEngine.py:46-83).  req_batch 1: the reference's
    protocol, one packet per put in both directions; 16: a put may carry a LIST of requests (single
    packets still work beside them) and a launch set's responses come back as one list."""
    from deeprecsys_amd.accelInferenceEngine import accelInferenceEngine
    from deeprecsys_amd.utils.packets import ServiceRequest, ServiceResponse
    a = _engine_args(tmp_path, accel_slots=2, accel_coalesce=3, accel_req_batch=req_batch)
    req, resp, ready = queue.Queue(), queue.Queue(), queue.Queue()
    sizes = [16, 1, 7, 3, 16, 2, 9, 5, 11, 4]
    packets = [ServiceRequest(batch_id=i % a.num_batches, epoch=0, arrival_time=float(i), batch_size=bs,
                              sub_id=0, total_sub_batches=1, exp_packet=False) for i, bs in enumerate(sizes)]
    if req_batch == 1:
        for p in packets:
            req.put(p)
    else:
        req.put(packets[:4])
        req.put(packets[4])
        req.put(packets[5:])
    req.put(None)
    accelInferenceEngine(a, req, 0, resp, ready)
    assert ready.get_nowait() is True
    got, puts = [], 0
    while True:
        r = resp.get_nowait()
        if r is None:
            break
        puts += 1
        assert isinstance(r, ServiceResponse) or (req_batch > 1 and isinstance(r, list) and 1 < len(r) <= 3)
        got.extend(r if isinstance(r, list) else [r])
    assert resp.empty()
    assert puts == len(sizes) if req_batch == 1 else puts < len(sizes)
    assert sorted((r.batch_id, r.batch_size, r.arrival_time) for r in got) == \
        sorted((i % a.num_batches, bs, float(i)) for i, bs in enumerate(sizes))
    assert all(r.out_batch_size == r.batch_size and r.consumer_id == 0 and r.model_id == 0 for r in got)
    assert all(r.queue_start_time <= r.inference_end_time for r in got)


def test_accel_engine_serves_a_mixed_model_stream_in_process(cpu_abi, tmp_path):
    from deeprecsys_amd.accelInferenceEngine import accelInferenceEngine
    from deeprecsys_amd.utils.packets import ServiceRequest
    from tests.test_harness import _write_mix_configs
    a = _engine_args(tmp_path, accel_slots=2, accel_coalesce=4, mix_config_files=_write_mix_configs(tmp_path))
    req, resp, ready = queue.Queue(), queue.Queue(), queue.Queue()
    plan = [(0, 5), (1, 3), (1, 16), (0, 1), (0, 8), (1, 2), (0, 16), (1, 7)]
    for i, (mid, bs) in enumerate(plan):
        req.put(ServiceRequest(batch_id=i % a.num_batches, epoch=0, arrival_time=float(i), batch_size=bs,
                               sub_id=0, total_sub_batches=1, exp_packet=False, model_id=mid))
    req.put(None)
    accelInferenceEngine(a, req, 0, resp, ready)
    got = []
    while True:
        r = resp.get_nowait()
        if r is None:
            break
        got.extend(r if isinstance(r, list) else [r])
    assert sorted((r.model_id, r.batch_size) for r in got) == sorted(plan)
    assert all(r.out_batch_size == r.batch_size for r in got)


def test_hip_library_loads_and_exports_every_symbol_of_the_header():
    """libdrs_hip.so itself (the product): loads without a GPU, exports exactly what
    include/drs.h declares, and -- with no device visible -- refuses to create an engine
    instead of falling back to anything."""
    import ctypes
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(N.LIB_PATH):
        subprocess.check_call(["make", "-C", os.path.join(root, "deeprecsys_amd", "csrc")], stdout=subprocess.DEVNULL)
    L = ctypes.CDLL(N.LIB_PATH)
    declared = set(re.findall(r"\b(drs_[a-z0-9_]+)\s*\(", open(os.path.join(root, "include", "drs.h")).read()))
    assert len(declared) >= 28
    for name in declared:
        assert hasattr(L, name), "libdrs_hip.so does not export %s" % name
    assert N.lib().drs_abi_version() == 5 and N.lib().drs_backend() == b"hip:gfx950"
    if N.device_count() == 0:
        with pytest.raises(N.DrsError) as ei:
            N.Engine(N.MODEL_DLRM, [16, 16], 8, [4, 8], [24, 4, 1], N.INTERACT_CAT, sigmoid_top=2,
                     max_batch=4, max_lookups=2, num_staged_batches=1, num_slots=1)
        assert ei.value.code == N.ERR_HIP and "no CPU fallback" in str(ei.value)


def test_product_binding_refuses_anything_but_the_hip_build(monkeypatch):
    """no environment variable or path puts a served query's arithmetic on the CPU.
    DRS_HIP_LIB is not read any more, and a library that answers drs_backend() with anything but
    "hip:*" -- the CPU restatement exports the same symbol set, drs_abi_version included -- is
    refused by the binding itself."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cpu_so = os.path.join(root, "oracle", "_build", "libdrs_cpu.so")
    assert os.path.exists(cpu_so)
    monkeypatch.setenv("DRS_HIP_LIB", cpu_so)
    import importlib
    fresh = importlib.reload(N)
    try:
        assert fresh.LIB_PATH.endswith(os.path.join("deeprecsys_amd", "libdrs_hip.so"))   # the env var is ignored
        monkeypatch.setattr(fresh, "LIB_PATH", cpu_so)
        monkeypatch.setattr(fresh, "_lib", None)
        with pytest.raises(ImportError) as ei:
            fresh.lib()
        assert "cpu:oracle" in str(ei.value) and fresh._lib is None
    finally:
        monkeypatch.undo()
        importlib.reload(N)


@pytest.mark.parametrize("kind,over", [
    ("din", dict(arch_sparse_feature_size=10, arch_mlp_bot="8-4")),         # a unit with two hidden layers over 40-byte rows
    ("din", dict(arch_sparse_feature_size=16, arch_mlp_bot="100")),         # a hidden layer wider than 64
    ("dien", dict(arch_sparse_feature_size=24, hidden_size=100)),           # neither one of din.hip's instances
    ("dlrm", dict(arch_sparse_feature_size=10, arch_mlp_bot="7-12-10", arch_mlp_top="9-1")),
])
def test_boundary_takes_the_shapes_the_reference_takes(cpu_abi, kind, over):
    """The reference builds its attention units / recurrent layers / MLPs from whatever widths the command line
    names (models/din.py:255-277, models/dien.py:308-380, models/dlrm_s_caffe2.py:435-437); round 6 made the boundary
    total (din_any.hip, sls_any_kernel).  Here: the restated ABI accepts the same shapes (it mirrors drs_create's
    checks) and the host wrappers build, feed and serve them -- outputs equal to the oracle driven directly."""
    rows = [60] + [40] * 3 + [70, 50] if kind != "dlrm" else [60, 40, 50]
    B = 12
    base = dict(arch_embedding_size="-".join(map(str, rows)), arch_mlp_top="24-2", arch_interaction_op="cat",
                num_indices_per_lookup=3, num_batches=1, max_mini_batch_size=B, mini_batch_size=B, numpy_rand_seed=3,
                model_type=kind, accel_slots=2)
    base.update(over)
    args = H.args_from({}, **base)
    net, lX, lS_l, lS_i, lT = H.materialize(args)
    net.create(None if kind != "dlrm" else lX[0], lS_l[0], lS_i[0], None)
    try:
        net.stage_batches(None if kind != "dlrm" else lX, lS_l, lS_i)
        om = H.oracle_model(net)
        for bs in (B, 5, 1):
            exp = om.forward(None if kind != "dlrm" else lX[0], lS_i[0], lS_l[0], bs=bs)
            assert np.array_equal(net.run_staged(0, bs), exp), (kind, bs)
    finally:
        net.engine.close()
    # what still has no form: a sample's activations beyond the 160 KB of LDS the any-shape kernels keep them in
    with pytest.raises(N.DrsError) as ei:
        N.Engine(N.MODEL_DIEN, [10] * 5, 32, [32, 16384], [16384 + 96, 2], max_batch=4, max_lookups=1, num_staged_batches=1, num_slots=1)
    assert ei.value.code == N.ERR_UNSUPPORTED


@pytest.mark.parametrize("case", ["dlrm_rm1_mini", "ncf_mini", "din_mini"])
def test_stand_alone_model_entry_prints_the_reference_table_lines(cpu_abi, case, capsys, tmp_path):
    """`python -m deeprecsys_amd.dlrm_s_hip <reference flags>` (models/dlrm_s_caffe2.py:575-661, models/run.sh):
    generates the inputs, builds the model, runs nepochs x num_batches forwards through run() and prints the
    six `***` lines in the format accelerator/predict_execution.py:10-29 parses; the last forward's output is
    the oracle's for the last input set; --enable_profiling adds the per-operator-type table
    (experiments/operator_breakdown/sweep_p.py:21-28)."""
    import json
    import os
    from deeprecsys_amd import dlrm_s_hip as M, latency_table
    meta, _ = H.load_fixture(case)
    cfg = {k: v for k, v in meta["args"].items() if k.startswith("arch_") or k in
           ("model_type", "model_name", "num_indices_per_lookup", "num_indices_per_lookup_fixed", "user_behavior_tables",
            "hidden_size", "attention_layers")}
    path = str(tmp_path / "cfg.json")
    json.dump(cfg, open(path, "w"))
    M.main(["--inference_only", "--use_accel", "--config_file", path, "--num_batches", "3", "--nepochs", "2",
            "--mini_batch_size", "8", "--max_mini_batch_size", "8", "--caffe2_net_type", "async_dag",
            "--enable_profiling"])
    out = capsys.readouterr().out
    f = str(tmp_path / "results.txt")
    open(f, "w").write(out)
    rows = latency_table.parse_results(f)
    assert len(rows) == 1 and len(rows[0]) == 6
    load, load_it, comp, comp_it, tot, tot_it = rows[0]
    assert load_it == pytest.approx(load / 6) and comp_it == pytest.approx(comp / 6) and tot_it == pytest.approx(tot / 6)
    assert tot == pytest.approx(load + comp) and load > 0 and comp > 0
    assert out.count("Time per operator type:") == 6 and "SparseLengthsSum" in out and " FC" in out
    assert "Created network" in out and "Running networks" in out


def test_cpu_abi_answers_the_engines_launch_set_preference(cpu_abi):
    """the CPU restatement used to answer 8 for every model, so the host code ran with other
    launch-set sizes off-GPU than on it.  It now restates the engine's rule (csrc/engine_options.hip): 12 for
    gather-bound DLRM, 16 where MLP launches overlap each other, 8 otherwise."""
    def pref(kind, rows, D, bot, top, L, key="preferred_coalesce", **kw):
        e = N.Engine(kind, rows, D, bot, top, max_batch=16, max_lookups=L, num_staged_batches=1, num_slots=3, **kw)
        try:
            return e.get_option(key)
        finally:
            e.close()
    # ... and launch sets in flight: 3 for the gather-bound models, 6 for the MLP-bound ones (round 6)
    assert pref(N.MODEL_DLRM, [1000] * 12, 32, [2560, 1024, 256, 32], [416, 512, 256, 1], 20, key="preferred_slots", sigmoid_top=3) == 6
    assert pref(N.MODEL_WND, [1000] * 27, 32, [512], [1376, 1024, 512, 256, 1], 1, key="preferred_slots", sigmoid_top=4) == 4   # (two at a time)
    assert pref(N.MODEL_DIN, [1000] * 254, 32, [96, 1, 32], [128, 200, 80, 2], 3, key="preferred_slots") == 3
    assert pref(N.MODEL_DIEN, [1000] * 43, 32, [32, 64], [160, 200, 80, 2], 1, key="preferred_slots") == 4     # (two at a time)
    assert pref(N.MODEL_DLRM, [1000] * 8, 64, [128, 64, 64], [576, 256, 64, 1], 80, key="preferred_slots", sigmoid_top=3) == 3
    assert pref(N.MODEL_NCF, [1000, 1000, 200, 200], 64, [512], [128, 256, 128, 64, 64], 1, key="preferred_slots") == 6
    assert pref(N.MODEL_DLRM, [1000] * 8, 64, [128, 64, 64], [576, 256, 64, 1], 80, sigmoid_top=3) == 12        # RMC1
    assert pref(N.MODEL_DLRM, [1000] * 12, 32, [2560, 1024, 256, 32], [416, 512, 256, 1], 20, sigmoid_top=3) == 16   # RM3
    assert pref(N.MODEL_DLRM, [1000] * 32, 64, [256, 128, 64], [2112, 128, 64, 1], 120, sigmoid_top=3) == 16     # RM2: a wide layer between two chains
    assert pref(N.MODEL_WND, [1000] * 27, 32, [512], [1376, 1024, 512, 256, 1], 1, sigmoid_top=4) == 16
    assert pref(N.MODEL_DIN, [1000] * 254, 32, [96, 1, 32], [128, 200, 80, 2], 3) == 8
