"""numpy restatement of the Caffe2 operators the hot path emits (TEST INFRASTRUCTURE).

A tiny interpreter that executes an op list *recorded from the reference's own
graph builders* (tools/gen_golden.py imports models/dlrm_s_caffe2.py etc. under a
recording stand-in for caffe2.python and stores `(op, inputs, outputs, kwargs)`).
Caffe2 itself is not in the reference tree (pinned as torch==1.4.0+cu92,
build/pip_requirements.txt:28) and cannot be installed here, so the operator
semantics are restated from Caffe2's published operator schemas:

  SparseLengthsSum(DATA, INDICES, LENGTHS) : segment sums of DATA rows, fp32,
      accumulated sequentially in index order; ENFORCE index range / length sum
  FC(X, W, b)             : X . W^T + b        (W is [N, K])
  Relu, Sigmoid           : elementwise
  Concat(axis=1)          : join along dim 1;  add_axis=1 stacks as a new dim 1
  BatchMatMul(A,B,trans_b): per-batch A[b] . B[b]^T
  Flatten(axis=1)         : [B, ...] -> [B, prod(...)]
  BatchGather(DATA, IDX)  : DATA[:, IDX]
  Cast(to=INT32)          : value preserving narrowing
  Sum(xs...)              : elementwise sum, left to right (one input: a copy)
  Reshape(X, shape)       : row-major reinterpretation (one -1 allowed)
  Softmax(axis)           : over the dims from `axis` on, flattened
  FC(axis=k)              : leading k dims are the batch
  BasicRNN                : caffe2.python.rnn_cell.BasicRNN (BasicRNNCell, forward only):
      i2h = FC(X [T,B,Din], i2h_w, i2h_b, axis=2); per step gates = FC(h_prev, gates_t_w,
      gates_t_b, axis=2); gates = Sum(gates, i2h[t]); h = tanh(gates); with seq_lengths,
      rows whose length <= t keep h_prev.  Outputs: every step's h [T,B,H], the last h [1,B,H]
An op whose parameter blob does not exist (a brew.fc weight that only Caffe2's param_init_net
would create) yields an UNAVAILABLE output, which propagates; models/dien.py's FC + Softmax over
the first RNN's states is such a chain, and it is dead: the Sum that follows overwrites its output
blob with a copy of the RNN states (models/dien.py:345-348).
  DequeueBlobs            : pops the blob most recently enqueued on that queue

Used only by tools/gen_golden.py (fixture generation) and tests/.  Floating-point
contractions (FC, BatchMatMul) are evaluated in float64 and rounded once to
float32: that is the "truth within fp32 rounding" every fp32 summation order
(MKL sgemm in the reference, k-ordered fma chains in oracle/drs_oracle.c and on
the GPU) must agree with to ~1e-6 relative.
"""
import numpy as np

INT32 = 2  # caffe2.proto TensorProto.DataType.INT32


def sparse_lengths_sum(data, indices, lengths):
    data = np.asarray(data, dtype=np.float32)
    indices = np.asarray(indices)
    lengths = np.asarray(lengths)
    if lengths.sum() != indices.size:
        raise ValueError("SparseLengthsSum: sum(LENGTHS) != len(INDICES)")
    if indices.size and (indices.min() < 0 or indices.max() >= data.shape[0]):
        raise IndexError("SparseLengthsSum: index out of range")
    out = np.zeros((lengths.size, data.shape[1]), dtype=np.float32)
    pos = 0
    for b, n in enumerate(lengths):
        acc = np.zeros(data.shape[1], dtype=np.float32)
        for j in range(pos, pos + int(n)):
            acc = (acc + data[int(indices[j])]).astype(np.float32)  # sequential fp32 adds
        out[b] = acc
        pos += int(n)
    return out


def fc(x, w, b):
    y = np.asarray(x, np.float64) @ np.asarray(w, np.float64).T + np.asarray(b, np.float64)
    return y.astype(np.float32)


def relu(x):
    return np.maximum(np.asarray(x, np.float32), np.float32(0))


def sigmoid(x):
    return (1.0 / (1.0 + np.exp(-np.asarray(x, np.float64)))).astype(np.float32)


def concat(xs, axis=1, add_axis=0):
    xs = [np.asarray(x) for x in xs]
    if add_axis:
        return np.stack(xs, axis=axis)
    return np.concatenate(xs, axis=axis)


def batch_matmul(a, b, trans_b=0):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    if trans_b:
        b = np.swapaxes(b, -1, -2)
    return np.matmul(a, b).astype(np.float32)


def flatten(x, axis=1):
    x = np.asarray(x)
    lead = int(np.prod(x.shape[:axis])) if axis else 1
    return x.reshape(lead, -1)


def batch_gather(data, idx):
    return np.asarray(data)[:, np.asarray(idx, dtype=np.int64)]


def reshape(x, shape):
    return np.asarray(x).reshape(tuple(int(v) for v in shape))


def softmax(x, axis=1):
    x = np.asarray(x, np.float64)
    flat = x.reshape(int(np.prod(x.shape[:axis])), -1)
    e = np.exp(flat - flat.max(axis=1, keepdims=True))
    return (e / e.sum(axis=1, keepdims=True)).reshape(x.shape).astype(np.float32)


def basic_rnn(x, seq_lengths, h0, i2h_w, i2h_b, gates_w, gates_b, activation="tanh"):
    assert activation == "tanh"
    x = np.asarray(x, np.float32)
    steps, B, _ = x.shape
    H = np.asarray(gates_w).shape[0]
    i2h = fc(x.reshape(steps * B, -1), i2h_w, i2h_b).reshape(steps, B, H)
    h = np.asarray(h0, np.float32).reshape(B, H)
    seq = np.asarray(seq_lengths).reshape(B)
    every = np.zeros((steps, B, H), np.float32)
    for t in range(steps):
        gates = (fc(h, gates_w, gates_b) + i2h[t]).astype(np.float32)
        hn = np.tanh(gates.astype(np.float64)).astype(np.float32)
        h = np.where((seq > t)[:, None], hn, h)
        every[t] = h
    return every, h.reshape(1, B, H)


class _Unavailable(object):
    pass


UNAVAILABLE = _Unavailable()


def run_ops(ops, blobs, queues=None):
    """Execute recorded ops in order.  `blobs`: name -> ndarray (weights + fed inputs).
    `queues`: queue blob name -> list of enqueued arrays (queue mode)."""
    ws = dict(blobs)
    queues = {k: list(v) for k, v in (queues or {}).items()}
    for op in ops:
        kind, ins, outs, kw = op["type"], op["inputs"], op["outputs"], op.get("kwargs", {})
        if kind != "DequeueBlobs" and any(i not in ws or ws[i] is UNAVAILABLE for i in ins):
            for o in outs:
                ws[o] = UNAVAILABLE
            continue
        if kind == "DequeueBlobs":
            ws[outs[0]] = queues[ins[0]].pop(0)
        elif kind == "Cast":
            assert kw.get("to") == INT32
            src = np.asarray(ws[ins[0]])
            assert src.size == 0 or (src.min() >= -2**31 and src.max() < 2**31)
            ws[outs[0]] = src.astype(np.int32)
        elif kind == "SparseLengthsSum":
            ws[outs[0]] = sparse_lengths_sum(ws[ins[0]], ws[ins[1]], ws[ins[2]])
        elif kind == "FC":
            x = np.asarray(ws[ins[0]])
            ax = int(kw.get("axis", 1))
            y = fc(x.reshape(int(np.prod(x.shape[:ax])), -1), ws[ins[1]], ws[ins[2]])
            ws[outs[0]] = y.reshape(x.shape[:ax] + (y.shape[1],))
        elif kind == "Reshape":
            import ast
            ws[outs[1]] = np.array(np.asarray(ws[ins[0]]).shape, dtype=np.int64)
            ws[outs[0]] = reshape(ws[ins[0]], ast.literal_eval(kw["shape"]))
        elif kind == "Softmax":
            ws[outs[0]] = softmax(ws[ins[0]], axis=int(kw.get("axis", 1)))
        elif kind == "BasicRNN":
            ws[outs[0]], ws[outs[1]] = basic_rnn(*[ws[i] for i in ins], activation=kw.get("activation"))
        elif kind == "Relu":
            ws[outs[0]] = relu(ws[ins[0]])
        elif kind == "Sigmoid":
            ws[outs[0]] = sigmoid(ws[ins[0]])
        elif kind == "Concat":
            ws[outs[0]] = concat([ws[i] for i in ins], axis=kw.get("axis", 1),
                                 add_axis=kw.get("add_axis", 0))
        elif kind == "BatchMatMul":
            ws[outs[0]] = batch_matmul(ws[ins[0]], ws[ins[1]], trans_b=kw.get("trans_b", 0))
        elif kind == "Flatten":
            ws[outs[0]] = flatten(ws[ins[0]], axis=kw.get("axis", 1))
        elif kind == "BatchGather":
            ws[outs[0]] = batch_gather(ws[ins[0]], ws[ins[1]])
        elif kind == "Sum":
            acc = np.asarray(ws[ins[0]], np.float32)
            for i in ins[1:]:
                acc = (acc + np.asarray(ws[i], np.float32)).astype(np.float32)
            ws[outs[0]] = acc
        else:
            raise NotImplementedError("op %s is not on the hot path" % kind)
    return ws
