"""CPU suite: the oracle (oracle/drs_oracle.c) against the golden fixtures captured
from the reference and against torch-CPU ops (the Caffe2 kernels' lineal
descendants).  No GPU."""
import numpy as np
import pytest
import torch

from oracle import c2ops
from oracle import oracle as orc
from tests import helpers as H


@pytest.mark.parametrize("case", H.MODEL_CASES)
def test_oracle_forward_matches_golden(case):
    """Graph, weights and inputs come from the reference's own builders; the expected
    prob_click is the recorded op list run by oracle/c2ops.py (fp64 contractions).
    The C oracle (fp32 k-ordered fma chains) must agree to fp32 round-off."""
    meta, z = H.load_fixture(case)
    args = H.args_from(meta["args"])
    net, lX, lS_l, lS_i, lT = H.materialize(args)
    om = H.oracle_model(net)
    dense = None if args.model_type in H.NO_DENSE else lX[0]
    out, R = om.forward(dense, lS_i[0], lS_l[0], want_R=True)
    exp = H.golden_output(meta, z)
    assert out.shape == exp.shape
    # fp32 k-ordered chains vs fp64-accumulated truth: well inside the 1e-4 north-star bar
    assert H.close(out, exp, rtol=2e-5, atol=1e-6), np.abs(out - exp).max()
    # interaction tensor (input of the top MLP) where the fixture has it; RM3's bottom
    # MLP has K=2560 all-positive inputs, so allow cancellation noise relative to max|R|
    key = {"dlrm": "expected/interaction", "wnd": "expected/interaction", "mtwnd": "expected/interaction",
           "ncf": "expected/feat_int", "din": "expected/top_fc_in",
           "dien": "expected/gru:::concat"}[args.model_type]
    if key in z.files:
        assert H.close(R, z[key], rtol=2e-5, atol_scale=2e-6)


@pytest.mark.parametrize("case", ["dlrm_cat_queue_small", "dlrm_dot_queue_small"])
def test_oracle_matches_golden_queue_requests(case):
    """Per-request inputs are the arrays the reference engine itself enqueued
    (inferenceEngine.py:200-215): a query is a prefix of a pre-generated batch."""
    meta, z = H.load_fixture(case)
    args = H.args_from(meta["args"])
    net, lX, lS_l, lS_i, lT = H.materialize(args)
    om = H.oracle_model(net)
    T = len(net.emb_w)
    for r, (bid, bs) in enumerate(meta["queue_requests"]):
        ids = [z["req/%d/id_inputs_%d" % (r, t)] for t in range(T)]
        lens = [z["req/%d/len_inputs_%d" % (r, t)] for t in range(T)]
        fc = z["req/%d/fc_inputs" % r]
        # the reference's slicing == prefix of the staged batch
        for t in range(T):
            assert np.array_equal(ids[t], lS_i[bid][t][:bs * args.num_indices_per_lookup])
            assert np.array_equal(lens[t], lS_l[bid][t][:bs])
            assert ids[t].dtype == np.int64 and lens[t].dtype == np.int32
        assert np.array_equal(fc, lX[bid][:bs])
        out = om.forward(fc, ids, lens)
        out2 = om.forward(lX[bid], lS_i[bid], lS_l[bid], bs=bs)
        assert np.array_equal(out, out2)
        exp = z["req/%d/expected/prob_click" % r]
        assert H.close(out, exp, rtol=2e-5, atol=1e-6)


@pytest.mark.parametrize("D", [4, 8, 32, 48, 64, 128])
@pytest.mark.parametrize("L", [0, 1, 7, 80, 130])
def test_oracle_sls_bitexact_vs_torch_embedding_bag(D, L):
    """torch-CPU embedding_bag(sum) descends from the Caffe2 EmbeddingLookup perfkernel
    (sequential fp32 sum per column): the oracle must be bit-identical to it."""
    rng = np.random.RandomState(D * 1000 + L)
    rows, bags = 997, 33
    W = rng.uniform(-1, 1, size=(rows, D)).astype(np.float32)
    lengths = rng.randint(0, L + 1, size=bags).astype(np.int32)
    lengths[::5] = L
    idx = rng.randint(0, rows, size=int(lengths.sum())).astype(np.int64)
    out = orc.sls(W, idx, lengths)
    offsets = np.concatenate([[0], np.cumsum(lengths)[:-1]]).astype(np.int64)
    ref = torch.nn.functional.embedding_bag(torch.from_numpy(idx), torch.from_numpy(W),
                                            torch.from_numpy(offsets), mode="sum").numpy()
    assert np.array_equal(out, ref)
    assert np.array_equal(out, c2ops.sparse_lengths_sum(W, idx, lengths))
    out32 = orc.sls(W, idx.astype(np.int32), lengths)
    assert np.array_equal(out, out32)


def test_oracle_sls_enforces_like_caffe2():
    W = np.ones((10, 4), np.float32)
    with pytest.raises(orc.OracleError) as e:
        orc.sls(W, np.array([0, 10], np.int64), np.array([2], np.int32))
    assert e.value.code == orc.ERR_INDEX_RANGE
    with pytest.raises(orc.OracleError) as e:
        orc.sls(W, np.array([0, -1], np.int64), np.array([2], np.int32))
    assert e.value.code == orc.ERR_INDEX_RANGE
    with pytest.raises(orc.OracleError) as e:
        orc.sls(W, np.array([0, 1, 2], np.int64), np.array([2], np.int32))
    assert e.value.code == orc.ERR_LENGTHS_SUM
    assert np.array_equal(orc.sls(W, np.zeros(0, np.int64), np.zeros(3, np.int32)), np.zeros((3, 4)))


@pytest.mark.parametrize("M,K,N", [(1, 3, 1), (5, 6, 12), (16, 128, 64), (33, 576, 256), (7, 100, 5)])
@pytest.mark.parametrize("act", [orc.ACT_NONE, orc.ACT_RELU, orc.ACT_SIGMOID])
def test_oracle_fc_vs_torch_addmm(M, K, N, act):
    rng = np.random.RandomState(M + K + N)
    x = rng.uniform(-1, 1, (M, K)).astype(np.float32)
    W = rng.normal(0, 0.1, (N, K)).astype(np.float32)
    b = rng.normal(0, 0.1, N).astype(np.float32)
    y = orc.fc(x, W, b, act)
    ref = torch.addmm(torch.from_numpy(b), torch.from_numpy(x), torch.from_numpy(W).t())
    if act == orc.ACT_RELU:
        ref = torch.relu(ref)
    elif act == orc.ACT_SIGMOID:
        ref = torch.sigmoid(ref)
    # two fp32 summation orders (torch sgemm vs k-ordered chain): round-off relative to max|y|
    assert H.close(y, ref.numpy(), rtol=1e-5, atol_scale=2e-6)
    # and against an explicit k-ordered fp32 fma chain in numpy (the oracle's definition)
    acc = np.zeros((M, N), np.float32)
    for k in range(K):
        acc = (x[:, k:k + 1].astype(np.float64) * W[:, k][None, :].astype(np.float64)
               + acc.astype(np.float64)).astype(np.float32)   # one rounding per fma
    z = acc + b
    if act == orc.ACT_RELU:
        z = np.maximum(z, 0)
    if act != orc.ACT_SIGMOID:
        assert np.array_equal(y, z.astype(np.float32))


@pytest.mark.parametrize("F,D,itself", [(4, 8, False), (4, 8, True), (9, 32, False), (33, 64, False)])
def test_oracle_interact_dot_vs_recorded_op_chain(F, D, itself):
    """Concat(add_axis) + BatchMatMul(trans_b) + Flatten + BatchGather(tril) + Concat
    (models/dlrm_s_caffe2.py:334-354) evaluated op by op in numpy vs the fused oracle."""
    rng = np.random.RandomState(F * D)
    B = 5
    T = rng.uniform(-1, 1, (B, F, D)).astype(np.float32)
    R = orc.interact_dot(T, itself)
    Z = c2ops.batch_matmul(T, T, trans_b=1)
    off = 1 if itself else 0
    tril = np.array([j + i * F for i in range(F) for j in range(i + off)])
    ref = c2ops.concat([T[:, 0, :], c2ops.batch_gather(c2ops.flatten(Z, 1), tril)], axis=1)
    assert R.shape == ref.shape
    assert H.close(R, ref, rtol=1e-5, atol=1e-6)
    tb = torch.from_numpy(T)
    assert H.close(c2ops.batch_matmul(T, T, 1), torch.bmm(tb, tb.transpose(1, 2)).numpy(), 1e-5, 1e-6)


def test_fill_value_is_deterministic_and_in_range():
    W = orc.fill_table_uniform(1000, 32, 3, -0.5, 0.25, 12345)
    assert W.min() >= -0.5 and W.max() <= 0.25
    assert abs(float(W.mean()) + 0.125) < 0.01
    W2 = orc.fill_table_uniform(1000, 32, 3, -0.5, 0.25, 12345, nthreads=4)
    assert np.array_equal(W, W2)
    assert not np.array_equal(W, orc.fill_table_uniform(1000, 32, 4, -0.5, 0.25, 12345))


def test_oracle_dien_matches_torch_rnn_fp64():
    """Independent opinion on the DIEN restatement: torch.nn.RNN (tanh; h_t = tanh(W_ih x_t + b_ih +
    W_hh h_{t-1} + b_hh), the published BasicRNN recurrence) in float64 over the embeddings in the
    reference's Reshape order (a row-major reinterpretation of [bs, U*D] as [U, bs, D],
    models/dien.py:316-320), then the top MLP.  Xavier-scale recurrent weights as in a live run."""
    import torch
    D, Hs, B = 16, 32, 5
    rows = [40, 30, 35, 25, 45, 50, 20, 60]           # 5 behaviour tables
    T, U = len(rows), len(rows) - 3
    args = H.args_from({}, arch_sparse_feature_size=D, arch_embedding_size="-".join(map(str, rows)),
                       arch_mlp_top="12-3", hidden_size=Hs, num_indices_per_lookup=2, model_type="dien",
                       numpy_rand_seed=7)
    np.random.seed(7)
    net = H.M.DIEN_Net(args)
    om = H.oracle_model(net)
    rng = np.random.RandomState(1)
    lens = [np.full(B, 2, dtype=np.int32) for _ in range(T)]
    idx = [rng.randint(0, rows[t], size=2 * B).astype(np.int64) for t in range(T)]
    out, R = om.forward(None, idx, lens, want_R=True)
    emb = [torch.from_numpy(net.emb_w[t].astype(np.float64))[torch.from_numpy(idx[t]).view(B, 2)].sum(1) for t in range(T)]
    X = torch.stack(emb[1:T - 2], 1).reshape(U, B, D)
    rnn = torch.nn.RNN(D, Hs, num_layers=2, nonlinearity="tanh").double()
    with torch.no_grad():
        for l in (0, 1):
            (iw, ib), (gw, gb) = net.rnn_w[l]
            getattr(rnn, "weight_ih_l%d" % l).copy_(torch.from_numpy(iw.astype(np.float64)))
            getattr(rnn, "bias_ih_l%d" % l).copy_(torch.from_numpy(ib.astype(np.float64)))
            getattr(rnn, "weight_hh_l%d" % l).copy_(torch.from_numpy(gw.astype(np.float64)))
            getattr(rnn, "bias_hh_l%d" % l).copy_(torch.from_numpy(gb.astype(np.float64)))
        _, hn = rnn(X)
        x = torch.cat([hn[1], emb[0], emb[T - 2], emb[T - 1]], 1)
        assert H.close(R, x.numpy(), rtol=1e-5, atol=1e-6), np.abs(R - x.numpy()).max()
        for W, b in net.top_w:
            x = torch.relu(x @ torch.from_numpy(W.astype(np.float64)).t() + torch.from_numpy(b.astype(np.float64)))
    assert H.close(out, x.numpy(), rtol=1e-5, atol=1e-6), np.abs(out - x.numpy()).max()


def _torch_forward(net, model_type, dense, idx, lens):
    """The whole forward on torch-CPU operators only -- embedding_bag(sum), addmm, bmm + tril gather,
    relu / sigmoid / tanh -- written against the reference's graph builders (file:line below), NOT
    against the oracle: a second, independent restatement of the same graph on the same weights."""
    import torch
    import torch.nn.functional as F
    T = len(net.emb_w)
    D = int(net.m_spa)
    tables = [torch.from_numpy(np.ascontiguousarray(w, dtype=np.float32)) for w in net.emb_w]
    emb = []
    for t in range(T):                                   # SparseLengthsSum per table (models/dlrm_s_caffe2.py:281-329)
        le = torch.from_numpy(np.asarray(lens[t], dtype=np.int64))
        offs = torch.cumsum(le, 0) - le
        emb.append(F.embedding_bag(torch.from_numpy(np.asarray(idx[t], dtype=np.int64)), tables[t], offs, mode="sum"))

    def mlp(x, layers, sigmoid_layer=-1):                # create_mlp: FC + Relu, Sigmoid at layer index sigmoid_layer (:223-279)
        for i, (W, b) in enumerate(layers):
            x = torch.addmm(torch.from_numpy(b), x, torch.from_numpy(W).t())
            x = torch.sigmoid(x) if i + 1 == sigmoid_layer else torch.relu(x)
        return x

    x = None if dense is None else torch.from_numpy(np.ascontiguousarray(dense, dtype=np.float32))
    if model_type == "ncf":                              # models/ncf.py:301-346: Sum | Concat -> MLP | Concat -> FC + Relu
        mf = emb[0] + emb[1]
        z = mlp(torch.cat([emb[2], emb[3]], 1), net.top_w)
        return mlp(torch.cat([mf, z], 1), net.final_w).numpy()
    if model_type == "din":                              # models/din.py:246-330
        ad, z = emb[T - 2], None
        for u, unit in enumerate(net.att_w):
            y = mlp(torch.cat([emb[1 + u], ad, emb[1 + u] + ad], 1), unit)
            z = y if z is None else z + y
        return mlp(torch.cat([emb[0], z, ad, emb[T - 1]], 1), net.top_w).numpy()
    if model_type == "dien":                             # models/dien.py:308-432 (Reshape = reinterpretation, two BasicRNN layers)
        U, Hs = T - 3, int(net.hidden_size)
        X = torch.stack(emb[1:T - 2], 1).reshape(U, -1, D)
        rnn = torch.nn.RNN(D, Hs, num_layers=2, nonlinearity="tanh")
        with torch.no_grad():
            for l in (0, 1):
                (iw, ib), (gw, gb) = net.rnn_w[l]
                getattr(rnn, "weight_ih_l%d" % l).copy_(torch.from_numpy(iw))
                getattr(rnn, "bias_ih_l%d" % l).copy_(torch.from_numpy(ib))
                getattr(rnn, "weight_hh_l%d" % l).copy_(torch.from_numpy(gw))
                getattr(rnn, "bias_hh_l%d" % l).copy_(torch.from_numpy(gb))
            _, hn = rnn(X)
        return mlp(torch.cat([hn[1], emb[0], emb[T - 2], emb[T - 1]], 1), net.top_w).numpy()
    if model_type == "wnd":                              # models/wide_and_deep.py:271-305
        return mlp(torch.cat([x] + emb, 1), net.top_w, net.sigmoid_top).numpy()
    if model_type == "mtwnd":                            # models/multi_task_wnd.py:286-316: all-ReLU trunk, Sigmoid in the heads
        shared = mlp(torch.cat([x] + emb, 1), net.top_w)
        return torch.cat([mlp(shared, head, net.sigmoid_top) for head in net.task_w], 1).numpy()
    d = mlp(x, net.bot_w, net.sigmoid_bot)               # models/dlrm_s_caffe2.py:331-386
    if net.arch_interaction_op == "cat":
        R = torch.cat([d] + emb, 1)
    else:
        Tt = torch.stack([d] + emb, 1)
        Z = torch.bmm(Tt, Tt.transpose(1, 2))
        Fn = T + 1
        off = 0 if net.arch_interaction_itself else -1
        li, lj = torch.tril_indices(Fn, Fn, off)
        R = torch.cat([d, Z[:, li, lj]], 1)
    return mlp(R, net.top_w, net.sigmoid_top).numpy()


@pytest.mark.parametrize("case", H.MODEL_CASES)
def test_torch_cpu_forward_second_opinion(case):
    """the fixtures' expected outputs are the recorded op list run by oracle/c2ops.py; here the
    same weights and inputs go through torch-CPU operators end to end (embedding_bag(sum) + addmm + bmm /
    tril + sigmoid / relu, nn.RNN for DIEN) -- a restatement that shares no code with c2ops.py or the C
    oracle.  It must agree with the golden output AND with the C oracle: the pin is two independent
    restatements deep on OUTPUTS, not only op by op."""
    import torch
    meta, z = H.load_fixture(case)
    args = H.args_from(meta["args"])
    net, lX, lS_l, lS_i, lT = H.materialize(args)
    dense = None if args.model_type in H.NO_DENSE else lX[0]
    with torch.no_grad():
        got = _torch_forward(net, args.model_type, dense, lS_i[0], lS_l[0])
    exp = H.golden_output(meta, z)
    assert got.shape == exp.shape
    assert H.close(got, exp, rtol=2e-5, atol=1e-6), np.abs(got - exp).max()
    out = H.oracle_model(net).forward(dense, lS_i[0], lS_l[0])
    assert H.close(got, out, rtol=2e-5, atol=1e-6), np.abs(got - out).max()


@pytest.mark.parametrize("kind,over", [
    ("din", dict(arch_sparse_feature_size=10, arch_mlp_bot="8-4")),          # attention unit with two hidden layers, 40-byte rows
    ("din", dict(arch_sparse_feature_size=16, arch_mlp_bot="100")),          # a hidden layer wider than 64
    ("din", dict(arch_sparse_feature_size=6, arch_mlp_bot="2-300-1")),
    ("dien", dict(arch_sparse_feature_size=24, hidden_size=100)),
    ("dien", dict(arch_sparse_feature_size=10, hidden_size=7)),
    ("dlrm", dict(arch_sparse_feature_size=10, arch_mlp_bot="7-12-10", arch_mlp_top="9-1", arch_interaction_op="dot")),
    ("wnd", dict(arch_sparse_feature_size=12, arch_mlp_bot="13", arch_mlp_top="20-6-1", num_indices_per_lookup=1)),
])
def test_oracle_agrees_with_torch_cpu_on_the_shapes_only_the_generic_kernels_serve(kind, over):
    """The reference's fixtures hold the shipped shapes only; the shapes the generic kernels admit (any embedding
    width, attention units of any depth, any hidden size -- din_any.hip, sls_any_kernel) are checked on the GPU
    against the C oracle, so the oracle itself gets its torch-CPU second opinion on them here."""
    import torch
    rows = [60] + [40] * 3 + [70, 50] if kind in ("din", "dien") else [60, 40, 50]
    B = 9
    base = dict(arch_embedding_size="-".join(map(str, rows)), arch_mlp_top="24-2", arch_interaction_op="cat",
                num_indices_per_lookup=3, num_batches=1, max_mini_batch_size=B, mini_batch_size=B, numpy_rand_seed=3,
                model_type=kind)
    base.update(over)
    args = H.args_from({}, **base)
    net, lX, lS_l, lS_i, lT = H.materialize(args)
    dense = None if kind in H.NO_DENSE else lX[0]
    with torch.no_grad():
        got = _torch_forward(net, kind, dense, lS_i[0], lS_l[0])
    out = H.oracle_model(net).forward(dense, lS_i[0], lS_l[0])
    assert got.shape == out.shape
    assert H.close(got, out, rtol=2e-5, atol=1e-6), np.abs(got - out).max()
