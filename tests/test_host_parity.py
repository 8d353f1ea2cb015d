"""CPU suite for the host side: the mirrors of the reference's Python surface against
values captured by running the reference itself (tests/golden/harness.json and the
per-model fixtures; tools/gen_golden.py)."""
import json
import os
import types
from unittest import mock

import numpy as np
import pytest

from deeprecsys_amd import latency_table, loadGenerator, scheduler, stats
from deeprecsys_amd.utils import packets
from deeprecsys_amd.utils.utils import EXTRA_FLAGS, cli
from tests import helpers as H

with open(os.path.join(H.GOLDEN, "harness.json")) as _f:
    HARNESS = json.load(_f)


# ---- cli ------------------------------------------------------------------------------
def _strip_extra(d):
    return {k: v for k, v in d.items() if k not in {n for n, _, _ in EXTRA_FLAGS}}


def test_cli_defaults_match_reference():
    assert _strip_extra(vars(cli([]))) == HARNESS["cli"]["defaults"]


@pytest.mark.parametrize("cfg", [k for k in HARNESS["cli"] if k.endswith(".json")])
def test_cli_config_file_overrides_like_reference(cfg, tmp_path):
    p = tmp_path / cfg
    p.write_text(json.dumps(HARNESS["cli"][cfg + ":json"]))
    got = _strip_extra(vars(cli(["--config_file", str(p)])))
    exp = dict(HARNESS["cli"][cfg])
    got.pop("config_file"), exp.pop("config_file")
    assert got == exp


def test_cli_run_deeprecsys_bundle(tmp_path):
    p = tmp_path / "dlrm_rm1.json"
    p.write_text(json.dumps(HARNESS["cli"]["dlrm_rm1.json:json"]))
    got = _strip_extra(vars(cli(HARNESS["cli"]["run_DeepRecSys.sh:argv"] + ["--config_file", str(p)])))
    exp = dict(HARNESS["cli"]["run_DeepRecSys.sh"])
    got.pop("config_file"), exp.pop("config_file")
    assert got == exp


def test_cli_config_beats_command_line(tmp_path):
    p = tmp_path / "c.json"
    p.write_text(json.dumps({"arch_embedding_size": "7-7", "num_indices_per_lookup_fixed": True}))
    a = cli(["--arch_embedding_size", "9-9-9", "--config_file", str(p)])
    assert a.arch_embedding_size == "7-7" and a.num_indices_per_lookup_fixed is True


# ---- load generator --------------------------------------------------------------------
@pytest.mark.parametrize("rec", HARNESS["partition_requests"])
def test_partition_requests(rec):
    a = types.SimpleNamespace(sub_task_batch_size=rec["sub_task_batch_size"])
    assert [int(x) for x in loadGenerator.partition_requests(a, rec["batch_size"])] == rec["chunks"]


@pytest.mark.parametrize("rec", HARNESS["batch_size_distribution"], ids=lambda r: "%s-%s" % (r["kind"], r["avg"]))
def test_batch_size_distribution_and_arrivals(rec):
    a = types.SimpleNamespace(batch_size_distribution=rec["kind"], avg_mini_batch_size=rec["avg"],
                              var_mini_batch_size=rec["var"], num_batches=32, max_mini_batch_size=1024,
                              nepochs=2, avg_arrival_rate=10)
    np.random.seed(rec["seed"])
    assert [int(x) for x in loadGenerator.model_batch_size_distribution(a)] == rec["sizes"]
    np.random.seed(rec["seed"])
    assert [int(x) for x in loadGenerator.model_arrival_times(a)] == rec["arrival_delays"]


# ---- scheduler ---------------------------------------------------------------------------
class _FakeQ(object):
    def __init__(self, n=0):
        self.n = n

    def qsize(self):
        return self.n

    def get(self, *a):
        self.n -= 1
        return 0


@pytest.mark.parametrize("rec", HARNESS["scheduler"], ids=lambda r: "%s-%s" % (r["mode"], r["script"]))
def test_scheduler_trajectories_match_reference(rec):
    a = types.SimpleNamespace(min_arr_range=1, max_arr_range=20, arr_steps=50, avg_arrival_rate=10.0,
                              batch_configs="512-256-128", accel_configs="96-128-192-256-384-512",
                              target_latency=25.0, stable_region=0.10, sched_timeout=16,
                              sub_task_batch_size=512, accel_request_size_thres=1024)
    with mock.patch("builtins.print"), mock.patch("time.sleep"):
        s = scheduler.Scheduler(a, _FakeQ(3), _FakeQ(2), _FakeQ(1), mode=rec["mode"])
        assert [float(v) for v in s.possible_arrival_rates] == rec["possible_arrival_rates"]
        for lat, exp in zip(rec["latencies"], rec["steps"]):
            args_o, rate, tuning = s.run(lat)
            got = {"arr_id": int(s.arr_id), "arrival_rate": float(rate), "tuning": bool(tuning),
                   "sub_task_batch_size": int(args_o.sub_task_batch_size),
                   "accel_request_size_thres": int(args_o.accel_request_size_thres)}
            assert got == exp


# ---- latency table (accelerator/) ------------------------------------------------------------
def test_predict_time_and_table_format(tmp_path):
    rec = HARNESS["predict_time"]
    gd = types.SimpleNamespace()
    for m in latency_table.MODELS:
        setattr(gd, m + "_exec_time", np.array(rec["table"]) * (1 + 0.1 * len(m)))
    for pt in rec["points"]:
        assert float(latency_table.predict_time(pt["model"], pt["batch_size"], gd)) == pytest.approx(pt["ms"], rel=1e-12)
    p = tmp_path / "results_probe.txt"
    p.write_text("\n".join(HARNESS["parse_gpu"]["lines"]) + "\n")
    assert latency_table.parse_results(str(p)) == HARNESS["parse_gpu"]["tuples"]
    # what we write, the reference's parser (restated) reads back
    q = tmp_path / "out.txt"
    latency_table.write_results(str(q), HARNESS["parse_gpu"]["tuples"])
    assert latency_table.parse_results(str(q)) == HARNESS["parse_gpu"]["tuples"]


# ---- packets -------------------------------------------------------------------------------
def test_packet_fields_are_the_wire_contract():
    r = packets.ServiceRequest(batch_id=3, epoch=1, arrival_time=2.5, batch_size=7, sub_id=0,
                               total_sub_batches=2, exp_packet=False)
    assert (r.batch_id, r.batch_size, r.epoch, r.arrival_time, r.total_sub_batches, r.sub_id, r.exp_packet) \
        == (3, 7, 1, 2.5, 2, 0, False)
    s = packets.ServiceResponse(consumer_id=4, epoch=1, batch_id=3, batch_size=7, arrival_time=2.5,
                                process_start_time=3.0, queue_end_time=3.5, inference_end_time=4.0,
                                out_batch_size=7, sub_id=0, total_sub_batches=2, exp_packet=False)
    assert s.queue_start_time == 3.0          # reference stores process_start_time under this name
    assert set(s.as_dict()) == {"consumer_id", "epoch", "batch_id", "batch_size", "arrival_time",
                                "queue_start_time", "queue_end_time", "inference_end_time",
                                "out_batch_size", "total_sub_batches", "exp_packet", "sub_id"}
    import pickle
    assert pickle.loads(pickle.dumps(s)).as_dict() == s.as_dict()
    str(r), str(s)


# ---- inputs + weights: RNG consumption order ---------------------------------------------------
@pytest.mark.parametrize("case", H.MODEL_CASES)
def test_inputs_and_weights_bitexact_vs_reference(case):
    meta, z = H.load_fixture(case)
    dg = meta["digests"]
    args = H.args_from(meta["args"])
    net, lX, lS_l, lS_i, lT = H.materialize(args)
    for j in range(meta["nbatches"]):
        assert H.sha(lX[j]) == dg["lX/%d" % j]["sha256"]
        assert H.sha(lT[j]) == dg["lT/%d" % j]["sha256"]
        assert H.sha(np.array(lS_l[j], dtype=np.int32)) == dg["lS_l/%d" % j]["sha256"]
        for t in range(len(lS_i[j])):
            assert H.sha(lS_i[j][t]) == dg["lS_i/%d/%d" % (j, t)]["sha256"]
    if args.model_type == "ncf":
        tabs = ["blob/emb:::mf_sls0_w", "blob/emb:::mf_sls1_w", "blob/emb:::mlp_sls0_w", "blob/emb:::mlp_sls1_w"]
    else:
        tabs = ["blob/emb:::sls%d_w" % t for t in range(len(net.emb_w))]
    for t, k in enumerate(tabs):
        assert H.sha(net.emb_w[t]) == dg[k]["sha256"]
        assert dg[k]["dtype"] == "float32"
    if args.model_type == "dlrm":
        for i, (W, b) in enumerate(net.bot_w):
            assert H.sha(W) == dg["blob/bot:::fc%d_w" % (i + 1)]["sha256"]
            assert H.sha(b) == dg["blob/bot:::fc%d_b" % (i + 1)]["sha256"]
        assert np.array_equal(net.tril_indices(), z["blob/interaction_tril_indices"])
        assert list(net.ln_top) == [z["blob/top:::fc1_w"].shape[1] if "blob/top:::fc1_w" in z.files
                                    else dg["blob/top:::fc1_w"]["shape"][1]] + \
            [int(x) for x in args.arch_mlp_top.split("-")]
    if args.model_type == "din":
        # one MLP per attention unit, created in table order (models/din.py:262-277)
        for u, unit in enumerate(net.att_w):
            for i, (W, b) in enumerate(unit):
                assert H.sha(W) == dg["blob/atten:::_fc_%d:::fc%d_w" % (u, i + 1)]["sha256"]
                assert H.sha(b) == dg["blob/atten:::_fc_%d:::fc%d_b" % (u, i + 1)]["sha256"]
    if args.model_type == "dien":
        # the values models/dien.py feeds for the two BasicRNN layers (:318-331,350-363)
        for l in (0, 1):
            (iw, ib), (gw, gb) = net.rnn_w[l]
            for nm, a in (("i2h_w", iw), ("i2h_b", ib), ("gates_t_w", gw), ("gates_t_b", gb)):
                assert H.sha(a) == dg["blob/rnn_%d/%s" % (l, nm)]["sha256"], (l, nm)
    pre = "mlpfc" if args.model_type == "ncf" else "top"
    for i, (W, b) in enumerate(net.top_w):
        assert H.sha(W) == dg["blob/%s:::fc%d_w" % (pre, i + 1)]["sha256"]
        assert H.sha(b) == dg["blob/%s:::fc%d_b" % (pre, i + 1)]["sha256"]
    if args.model_type == "ncf":
        assert H.sha(net.final_w[0][0]) == dg["blob/final:::fc1_w"]["sha256"]
        assert H.sha(net.final_w[0][1]) == dg["blob/final:::fc1_b"]["sha256"]


def test_reference_graph_is_what_the_engine_wires():
    """The op list recorded from the reference builder is exactly the fused pipeline:
    T x SparseLengthsSum -> bottom FC/Relu chain -> (dot ops | Concat) -> top chain -> Sigmoid."""
    meta, _ = H.load_fixture("dlrm_dot_small")
    kinds = [op["type"] for op in meta["ops"]]
    assert kinds == ["SparseLengthsSum"] * 3 + ["FC", "Relu"] * 2 + \
        ["Concat", "BatchMatMul", "Flatten", "BatchGather", "Concat"] + ["FC", "Relu"] * 2 + ["FC", "Sigmoid"]
    meta, _ = H.load_fixture("dlrm_cat_queue_small")
    kinds = [op["type"] for op in meta["ops"]]
    assert kinds[:4] == ["DequeueBlobs", "Cast", "DequeueBlobs", "SparseLengthsSum"]
    assert meta["feed_dtypes"]["emb:::sls0_i"] == "int64" and meta["feed_dtypes"]["emb:::sls0_l"] == "int32"
    assert meta["ops"][-1]["outputs"] == ["prob_click"]


def test_model_shape_checks_exit_like_reference():
    a = H.args_from({}, arch_sparse_feature_size=8, arch_embedding_size="10-10", arch_mlp_bot="4-6",
                    arch_mlp_top="4-1", arch_interaction_op="dot")
    with pytest.raises(SystemExit):
        H.M.DLRM_Net(a)          # m_spa != ln_bot[-1]
    a.arch_mlp_bot = "4-8"
    a.arch_interaction_op = "sum"
    with pytest.raises(SystemExit):
        H.M.DLRM_Net(a)          # unknown interaction op


# ---- response reassembly + QPS / tail latency formulae -------------------------------------------
def test_response_aggregator_matches_reference_formulae():
    rng = np.random.RandomState(0)
    agg = stats.ResponseAggregator(request_granularity=8)
    sent = []
    t = 100.0
    expect_lat, expect_final = [], []
    pid = []
    for q in range(40):
        exp_packet = q < 10
        pieces = int(rng.randint(1, 4))
        arr = t
        ends = []
        for sub in range(pieces):
            end = arr + 0.001 * rng.randint(1, 30)
            ends.append(end)
            r = packets.ServiceResponse(consumer_id=sub, epoch=0, batch_id=q, batch_size=8, arrival_time=arr + 1e-6 * sub,
                                        process_start_time=arr, queue_end_time=end, inference_end_time=end,
                                        out_batch_size=8, sub_id=sub, total_sub_batches=pieces, exp_packet=exp_packet)
            sent.append(r)
        t += 0.01
    order = rng.permutation(len(sent))
    # a query's pieces may arrive interleaved with other queries'
    for i in order:
        lat, running = agg.add(sent[i])
        if running is not None:
            pid.append(running)
    # restate: per key, latency = max(end) - min(arrival)
    by = {}
    for r in sent:
        k = (r.epoch, r.batch_id, r.exp_packet)
        a0, e0 = by.get(k, (np.inf, -np.inf))
        by[k] = (min(a0, r.arrival_time), max(e0, r.inference_end_time))
    assert sorted(agg.response_latencies) == pytest.approx(sorted(e - a for a, e in by.values()))
    finals = [e - a for (ep, b, x), (a, e) in by.items() if not x]
    assert sorted(agg.final_response_latencies) == pytest.approx(sorted(finals))
    assert len(pid) == 40 // 8
    s = agg.summary()
    meas = [r for r in agg.responses_list if not r["exp_packet"] and r["sub_id"] == 0]
    assert s["qps"] == pytest.approx(len(meas) / (meas[-1]["inference_end_time"] - meas[0]["inference_end_time"]))
    assert s["p99_ms"] == pytest.approx(np.percentile(finals, 99) * 1000.)
    # histogram route used by the multi-GPU bench agrees with the exact percentile
    h = stats.latency_histogram(finals)
    # (a histogram percentile is the upper edge of the bin holding the "higher" order statistic)
    assert stats.percentile_from_histogram(h, 99) == pytest.approx(
        np.percentile(finals, 99, method="higher") * 1e3, rel=0.01)


# ---- locality-aware index traces: data_generator/trace_generator.py, trace_profile.py ----
def _traces():
    import json
    with open(os.path.join(H.GOLDEN, "traces.json")) as f:
        meta = json.load(f)
    return meta, np.load(os.path.join(H.GOLDEN, "traces.npz"))


def test_trace_generate_lru_equals_reference_for_the_same_seeds():
    """The LRU-stack walk (deque + cursor here, one big list there) gives the reference's trace
    element for element: shipped profile (99.9 % new references) and a reuse-heavy one, table
    exhaustion (64 lines, 400 references) and the padding mode."""
    import random
    from deeprecsys_amd.data_generator import trace_generator as TG
    meta, z = _traces()
    for c in meta["lru_cases"]:
        lsd, csd = z[c["profile"] + "/list_sd"].tolist(), z[c["profile"] + "/cumm_sd"].tolist()
        random.seed(c["seed"])
        np.random.seed(c["seed"])
        got = TG.trace_generate_lru(c["table_size"], lsd, csd, c["len"], c["padding"])
        assert np.array_equal(np.array(got, dtype=np.uint64), z[c["key"]]), c
        assert all(isinstance(v, np.uint64) for v in got[:3])
    hot = z["lru/hot/500_1500_7_0"]
    assert len(np.unique(hot)) < 0.7 * hot.size          # the reuse-heavy profile really reuses lines


def test_trace_profile_and_distribution_equal_reference(tmp_path):
    from deeprecsys_amd.data_generator import trace_generator as TG
    meta, z = _traces()
    tr = z["profile/trace"]
    for max_sd in meta["profile_max_sd"]:
        sds, lines = TG.trace_profile(tr, max_sd)
        assert np.array_equal(sds, z["profile/%d/stack_distances" % max_sd])
        assert np.array_equal(lines, z["profile/%d/line_accesses" % max_sd])
    # distribution -> file -> back, in the reference's two-line format
    list_sd, prob_sd, cumm_sd = TG.stack_distance_distribution(z["profile/1000/stack_distances"].tolist())
    assert list_sd[0] == 0 and abs(cumm_sd[-1] - 1.0) < 1e-12 and abs(sum(prob_sd) - 1.0) < 1e-12
    p = str(tmp_path / "sd_cumm")
    TG.write_dist_to_file(p, list_sd, cumm_sd)
    back = TG.read_dist_from_file(p)
    assert back[0] == list_sd and np.allclose(back[1], cumm_sd, rtol=0, atol=0)
    # a trace synthesised from a profile has (statistically) that profile's share of new lines
    import random
    random.seed(3)
    np.random.seed(3)
    syn = TG.trace_generate_lru(5000, z["hot/list_sd"].tolist(), z["hot/cumm_sd"].tolist(), 4000)
    sds, _ = TG.trace_profile(np.array(syn, dtype=np.int64), 1000)
    new_share = float(np.mean(np.array(sds) == 0))
    assert abs(new_share - 0.45) < 0.05
    bags = TG.bags_from_trace(syn, 50, 80)
    assert bags.dtype == np.int64 and bags.size == 4000 and bags.max() < 5000


def _write_profile(tmp_path, name="hot"):
    from deeprecsys_amd.data_generator import trace_generator as TG
    _, z = _traces()
    path = str(tmp_path / ("dist_emb_%s.log" % name))
    TG.write_dist_to_file(path, z[name + "/list_sd"].tolist(), z[name + "/cumm_sd"].tolist())
    return path


def test_data_generation_synthetic_goes_through_the_trace_generator(tmp_path):
    """`--data_generation synthetic` (data_generator/dlrm_data_caffe2.py:34-60,152-222): the engine's
    start-up sequence yields per-table index streams synthesised from the profile file -- a bag is
    np.unique of its references (sorted, lengths reset to what is left), one seed fixes everything,
    the profile's reuse shows up ACROSS bags, and indices stay inside their tables; `dataset` and
    unknown modes exit like the reference."""
    path = _write_profile(tmp_path, "hot")
    rows = "3000-5000-4000"
    a = H.args_from({}, arch_sparse_feature_size=16, arch_embedding_size=rows, arch_mlp_bot="8-16", arch_mlp_top="8-1",
                    arch_interaction_op="dot", num_indices_per_lookup=20, num_batches=3, max_mini_batch_size=32,
                    mini_batch_size=32, numpy_rand_seed=9, model_type="dlrm", data_generation="synthetic",
                    data_trace_file=path)
    net, lX, lS_l, lS_i, lT = H.materialize(a)
    net2, lX2, lS_l2, lS_i2, _ = H.materialize(a)
    assert len(lX) == 3 and lX[0].shape == (32, 8) and lX[0].dtype == np.float32
    for b in range(3):
        assert np.array_equal(lX[b], lX2[b])
        for t, size in enumerate((3000, 5000, 4000)):
            ln, ix = lS_l[b][t], lS_i[b][t]
            assert np.array_equal(ln, lS_l2[b][t]) and np.array_equal(ix, lS_i2[b][t])          # one seed fixes everything
            assert ln.dtype == np.int32 and ln.shape == (32,) and int(ln.sum()) == ix.size
            assert 1 <= ln.min() and ln.max() <= 20 and ix.min() >= 0 and ix.max() < size
            o = 0
            for n in ln:                                                                             # np.unique per bag
                assert np.all(np.diff(ix[o:o + n]) > 0)
                o += n
    assert any(int(l.min()) < 20 for per in lS_l for l in per)                                      # the hot profile repeats lines within a bag
    # reuse across bags: far fewer distinct rows than references (the uniform generator: ~all distinct)
    allrefs = np.concatenate([lS_i[b][1] for b in range(3)])
    assert np.unique(allrefs).size < 0.75 * allrefs.size
    # the model weights follow the inputs in the same numpy stream, like the random mode
    assert net.emb_w[0].shape == (3000, 16)
    # fixed-length bags (what bench.py --trace times): duplicates and reference order kept
    from deeprecsys_amd.data_generator.dlrm_data import DLRMDataGenerator
    np.random.seed(9)
    _, _, fl, fi = DLRMDataGenerator(a).generate_synthetic_input_data(2, 16, False, 20, True, 8, np.array([3000, 5000]), path,
                                                                      False, unique=False)
    assert all(int(np.min(l)) == int(np.max(l)) == 20 for per in fl for l in per) and len(fi[0][0]) == 16 * 20
    for mode in ("dataset", "nonsense"):
        a.data_generation = mode
        with pytest.raises(SystemExit):
            H.materialize(a)


def test_measured_mi355x_tables_load_like_the_reference_tables():
    """profiles/accelerator_mi355x/ is a drop-in for the reference's accelerator/ directory:
    GPU_Data reads a results_<model>.txt for EVERY model name it knows (wnd, rm1-3, ncf, mtwnd,
    din, dien; accelerator/predict_execution.py:49-62) and predict_time interpolates in it."""
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles",
                        "accelerator_mi355x") + os.sep
    gd = latency_table.GPU_Data(root_dir=root, hardware="amd_mi355x")
    for m in latency_table.MODELS:
        t = getattr(gd, m + "_exec_time")
        assert t.shape == (6,) and np.all(t > 0), m
        mid = float(latency_table.predict_time(m, 32, gd))
        assert min(t[2], t[3]) <= mid <= max(t[2], t[3]), m     # 16 < 32 < 64


# ------------------------------------------------------------------------------------
# N = 8 host readiness (VERDICT r4 #6a): every rank / accelerator engine process is bound to the cores of
# its GPU's NUMA node before it pins memory or starts workers (deeprecsys_amd/utils/affinity.py)
def test_rank_core_masks_for_1_2_4_8_ranks():
    from deeprecsys_amd.utils import affinity as A
    assert A.parse_cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11}
    assert A.format_cpulist({0, 1, 2, 3, 8, 10, 11}) == "0-3,8,10-11"
    # the GPU box of this pool: two sockets, 128 hardware threads each, GPUs 0-2 and 7 on node 0, 3-6 on node 1
    # (gpurun_out host probe: /sys/class/drm/card*/device/numa_node = 0 0 0 1 1 1 1 0)
    nodes = [0, 0, 0, 1, 1, 1, 1, 0]
    cpus = {0: set(range(0, 64)) | set(range(128, 192)), 1: set(range(64, 128)) | set(range(192, 256))}
    allowed = set(range(256))
    for n in (1, 2, 4, 8):
        plan = A.plan(nodes, cpus, allowed, n)
        assert len(plan) == n and all(src == "numa" for _, src in plan)
        for r, (cores, _) in enumerate(plan):
            assert cores and set(cores) <= cpus[nodes[r]], (n, r)          # on the GPU's own node
        for a in range(n):                                                   # ranks never share a core
            for b in range(a + 1, n):
                assert not set(plan[a][0]) & set(plan[b][0]), (n, a, b)
        # the ranks of one node split its cores evenly
        for node in (0, 1):
            on = [len(plan[r][0]) for r in range(n) if nodes[r] == node]
            assert not on or (max(on) - min(on) <= 1 and sum(on) == 128), (n, node, on)
    assert len(A.plan(nodes, cpus, allowed, 8)[0][0]) == 32 and len(A.plan(nodes, cpus, allowed, 1)[0][0]) == 128
    # a cgroup that leaves 16 cores, all on node 0: the ranks of node 1 fall back to an even deal of what is allowed
    small = set(range(16))
    plan = A.plan(nodes, cpus, small, 8)
    assert [src for _, src in plan] == ["numa"] * 3 + ["even"] * 4 + ["numa"]
    assert all(cores and set(cores) <= small for cores, _ in plan)
    assert sorted(c for r in (0, 1, 2, 7) for c in plan[r][0]) == list(range(16))
    # no topology at all (containers: numa_node = -1): an even deal, every core used once
    plan = A.plan([-1] * 8, {}, allowed, 8)
    assert all(src == "even" and len(cores) == 32 for cores, src in plan)
    assert sorted(c for cores, _ in plan for c in cores) == list(range(256))
    # more ranks than cores: ranks share, nobody is left without a core
    plan = A.plan([-1] * 8, {}, {5, 6}, 8)
    assert all(len(cores) == 1 and cores[0] in (5, 6) for cores, _ in plan)


def test_bind_rank_applies_the_mask_in_a_child_process():
    """bind_rank on this host (whatever its topology): the mask it reports is the mask the process has."""
    import multiprocessing as mp
    import os
    from deeprecsys_amd.utils import affinity as A

    def child(q, r, n):
        info = A.bind_rank(r, n)
        q.put((info, sorted(os.sched_getaffinity(0))))
    ctx = mp.get_context("fork")
    before = sorted(os.sched_getaffinity(0))
    seen = []
    for r in range(2):
        q = ctx.Queue()
        p = ctx.Process(target=child, args=(q, r, 2))
        p.start()
        info, mask = q.get(timeout=30)
        p.join()
        assert A.parse_cpulist(info["cpus"]) == set(mask) and info["n_cpus"] == len(mask)
        assert set(mask) <= set(before)
        seen.append(set(mask))
    if len(before) >= 2:
        assert not seen[0] & seen[1]
    assert sorted(os.sched_getaffinity(0)) == before          # the parent is untouched
    os.environ["DRS_NO_AFFINITY"] = "1"
    try:
        assert A.bind_rank(0, 2)["source"] == "unchanged"
    finally:
        del os.environ["DRS_NO_AFFINITY"]


def test_tune_table_placement_search_logic_on_a_scripted_engine():
    """DLRM_Net.tune_table_placement against an engine whose gather time is scripted per
    (arena, load policy): it times both policies on every arena, stops as soon as one arena is a level (8 %)
    faster than another, keeps the best (arena, policy), releases everything else, and leaves the
    allocation mode as it found it."""
    from deeprecsys_amd import dlrm_s_hip as M

    class Eng:
        max_batch = 256

        def __init__(self, us):
            self.us = us                       # us[arena][policy index: nt, plain]
            self.opt = {"mlp_streams": 1, "gather_bound": 1, "preferred_coalesce": 12, "preferred_slots": 3, "shared_stream": 2, "sls_nt": 1, "table_bytes": 2 << 30,
                        "table_alloc": 0, "table_vmm_chunk": -1}
            self.arenas, self.cur, self.log, self.busy = 1, 0, [], {}
            self.user_options = set()

        def get_option(self, k):
            return self.opt[k]

        def set_option(self, k, v, user=True):
            self.log.append((k, v))
            if user:
                self.user_options.add(k)
            if k == "table_placement":
                if v == -1:
                    if self.arenas >= len(self.us):
                        raise M.N.DrsError(-2, "drs_set_option", "no room")
                    self.arenas += 1
                    self.cur = self.arenas - 1
                elif v == -2:
                    self.kept, self.arenas, self.cur = self.cur, 1, 0
                else:
                    self.cur = v
            else:
                self.opt[k] = v

        num_slots = 3

        def forward_multi_async(self, slot, *a):
            assert 0 <= slot < 3 and not self.busy.get(slot), "a slot is waited for before it is reused"
            self.busy[slot] = True

        def wait(self, slot, *a):
            assert self.busy.get(slot)
            self.busy[slot] = False

        def reset_kernel_time(self): pass
        def set_profiling(self, *a): pass

        def kernel_time(self, _k):
            return self.us[self.cur][0 if self.opt["sls_nt"] else 1] * 1e-3, 1

    def run(us, candidates=6, prepare=None, **kw):
        net = M.DLRM_Net.__new__(M.DLRM_Net)
        net.engine, net._n_staged = Eng(us), 4
        if prepare:
            prepare(net.engine)
        return net.tune_table_placement(candidates, sets=4, **kw), net.engine

    # the third arena is a level faster under plain loads: the search stops there and keeps (arena 2, plain)
    res, eng = run([[86.5, 87.5], [86.6, 87.9], [85.0, 79.0], [78.0, 78.0]])
    assert res["kept"] == 2 and res["sls_nt"] == 0 and res["candidates"] == 3 and res["losers"] == "freed"
    assert eng.kept == 2 and eng.arenas == 1 and eng.opt["sls_nt"] == 0 and eng.opt["table_alloc"] == 0
    assert ("table_spacer", 2 << 30) in eng.log                      # a spacer of the arena's size between candidates
    # no spread anywhere: every candidate is tried, and the first arena stays (another must be 1 % faster to replace it)
    res, eng = run([[86.0, 87.0]] * 3 + [[85.8, 87.0]] + [[86.0, 87.0]] * 2)
    assert res["candidates"] == 6 and res["kept"] == 0 and res["sls_nt"] == 1 and eng.kept == 0
    res, eng = run([[86.0, 87.0]] * 3 + [[83.8, 87.0]] + [[86.0, 87.0]] * 2)
    assert res["candidates"] == 6 and res["kept"] == 3 and res["sls_nt"] == 1 and eng.kept == 3
    assert not eng.user_options                                      # (the search's own settings are not the caller's)
    # a load policy the caller pinned is kept: only the arenas are searched -- and the caller's allocation mode comes back
    def pin(e):
        e.set_option("sls_nt", 1)
        e.set_option("table_alloc", 1)
    res, eng = run([[86.5, 70.0], [86.6, 70.0], [79.0, 70.0]], prepare=pin)
    assert res["kept"] == 2 and res["sls_nt"] == 1 and res["policies"] == ["nt"] and eng.opt["sls_nt"] == 1 and eng.opt["table_alloc"] == 1
    # the transient footprint is bounded: 2 GB arenas + 2 GB spacers, 9 GB for this engine -> two candidates; a GPU shared
    # by four engine processes -> none beside the first arena
    res, eng = run([[86.0, 87.0]] * 6, max_extra_gb=9)
    assert res["candidates"] == 2
    res, eng = run([[86.0, 87.0]] * 6, max_extra_gb=9, sharers=4)
    assert res["candidates"] == 1 and res["kept"] == 0
    # no room for a second arena: the policies of the first one still compete
    res, eng = run([[86.0, 84.0]])
    assert res["candidates"] == 1 and res["kept"] == 0 and res["sls_nt"] == 0 and eng.arenas == 1
    # MLP-bound models are left alone
    net = M.DLRM_Net.__new__(M.DLRM_Net)
    net.engine, net._n_staged = Eng([[1, 1]]), 4
    net.engine.opt["mlp_streams"], net.engine.opt["gather_bound"] = 4, 0
    assert net.tune_table_placement(6) is None and net.engine.log == []
