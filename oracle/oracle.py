"""ctypes binding of oracle/drs_oracle.c (TEST INFRASTRUCTURE, see that file's header).

The C library restates, on the CPU, the operator graph the reference emits in
models/dlrm_s_caffe2.py:367-389 (and wide_and_deep.py:282-305, ncf.py:317-346)
together with the Caffe2 operator arithmetic it invokes.  Nothing under
deeprecsys_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libdrs_oracle.so")

MODEL_DLRM, MODEL_WND, MODEL_NCF, MODEL_MTWND, MODEL_DIN, MODEL_DIEN = 0, 1, 2, 3, 4, 5
INTERACT_DOT, INTERACT_CAT = 0, 1
ACT_NONE, ACT_RELU, ACT_SIGMOID = 0, 1, 2

ERR_INDEX_RANGE = -4
ERR_LENGTHS_SUM = -5


def build(force=False):
    """Compile the oracle with gcc (Makefile in this directory)."""
    src = os.path.join(_HERE, "drs_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


_lib = None

_f32p = C.POINTER(C.c_float)
_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)


class _Model(C.Structure):
    _fields_ = [
        ("model_kind", C.c_int32), ("T", C.c_int32), ("D", C.c_int32),
        ("rows", _i64p),
        ("tables", C.POINTER(_f32p)),
        ("n_bot", C.c_int32), ("ln_bot", _i32p),
        ("bot_W", C.POINTER(_f32p)), ("bot_b", C.POINTER(_f32p)),
        ("n_top", C.c_int32), ("ln_top", _i32p),
        ("top_W", C.POINTER(_f32p)), ("top_b", C.POINTER(_f32p)),
        ("final_W", _f32p), ("final_b", _f32p), ("final_m", C.c_int32),
        ("interaction_op", C.c_int32), ("itself", C.c_int32), ("sigmoid_top", C.c_int32),
        ("bot_Wt", C.POINTER(_f32p)), ("top_Wt", C.POINTER(_f32p)), ("final_Wt", _f32p),
        ("n_task", C.c_int32), ("ln_task", _i32p), ("num_tasks", C.c_int32), ("task_sigmoid", C.c_int32),
        ("task_W", C.POINTER(_f32p)), ("task_b", C.POINTER(_f32p)), ("task_Wt", C.POINTER(_f32p)),
        ("n_att", C.c_int32), ("ln_att", _i32p), ("att_W", C.POINTER(_f32p)), ("att_b", C.POINTER(_f32p)),
        ("rnn_hidden", C.c_int32), ("rnn_w", _f32p * 8),
    ]


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        L.orc_max_threads.restype = C.c_int32
        L.orc_fill_value.restype = C.c_float
        L.orc_fill_value.argtypes = [C.c_uint64, C.c_int32, C.c_uint64, C.c_float, C.c_float]
        L.orc_fill_table_uniform.argtypes = [_f32p, C.c_int64, C.c_int32, C.c_int32, C.c_float,
                                             C.c_float, C.c_uint64, C.c_int32]
        L.orc_sls_i64.argtypes = [_f32p, C.c_int64, C.c_int32, _i64p, _i32p, C.c_int64, C.c_int64,
                                  _f32p, C.c_int32]
        L.orc_sls_i32.argtypes = [_f32p, C.c_int64, C.c_int32, _i32p, _i32p, C.c_int64, C.c_int64,
                                  _f32p, C.c_int32]
        L.orc_fc.argtypes = [_f32p, C.c_int64, C.c_int32, _f32p, _f32p, C.c_int32, C.c_int32,
                             _f32p, C.c_int32]
        L.orc_interact_dot.argtypes = [_f32p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, _f32p,
                                       C.c_int32]
        L.orc_forward.argtypes = [C.POINTER(_Model), C.c_int32, _f32p, C.POINTER(_i64p), _i64p,
                                  C.POINTER(_i32p), _f32p, _f32p, C.c_int32]
        _lib = L
    return _lib


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_f32p)


def max_threads():
    return int(lib().orc_max_threads())


class OracleError(RuntimeError):
    def __init__(self, code, what):
        super().__init__("%s failed with status %d" % (what, code))
        self.code = code


def fill_table_uniform(rows, D, t, lo, hi, seed, nthreads=0):
    W = np.empty((rows, D), dtype=np.float32)
    rc = lib().orc_fill_table_uniform(W.ctypes.data_as(_f32p), rows, D, t, lo, hi, seed, nthreads)
    if rc:
        raise OracleError(rc, "orc_fill_table_uniform")
    return W


def sls(W, idx, lengths, nthreads=1):
    """SparseLengthsSum(W, idx, lengths) -> [n_bags, D]; idx may be int32 or int64."""
    W, Wp = _f32(W)
    lengths = np.ascontiguousarray(lengths, dtype=np.int32)
    idx = np.ascontiguousarray(idx)
    out = np.empty((lengths.size, W.shape[1]), dtype=np.float32)
    if idx.dtype == np.int32:
        fn, ip = lib().orc_sls_i32, idx.ctypes.data_as(_i32p)
    else:
        idx = idx.astype(np.int64, copy=False)
        fn, ip = lib().orc_sls_i64, idx.ctypes.data_as(_i64p)
    rc = fn(Wp, W.shape[0], W.shape[1], ip, lengths.ctypes.data_as(_i32p), lengths.size, idx.size,
            out.ctypes.data_as(_f32p), nthreads)
    if rc:
        raise OracleError(rc, "orc_sls")
    return out


def fc(x, W, b, act=ACT_NONE, nthreads=1):
    x, xp = _f32(x)
    W, Wp = _f32(W)
    b, bp = _f32(b)
    M, K = x.shape
    N = W.shape[0]
    assert W.shape[1] == K and b.shape[0] == N
    y = np.empty((M, N), dtype=np.float32)
    rc = lib().orc_fc(xp, M, K, Wp, bp, N, act, y.ctypes.data_as(_f32p), nthreads)
    if rc:
        raise OracleError(rc, "orc_fc")
    return y


def interact_dot(T, itself=False, nthreads=1):
    T, Tp = _f32(T)
    B, F, D = T.shape
    P = F * (F - 1) // 2 + (F if itself else 0)
    R = np.empty((B, D + P), dtype=np.float32)
    rc = lib().orc_interact_dot(Tp, B, F, D, int(bool(itself)), R.ctypes.data_as(_f32p), nthreads)
    if rc:
        raise OracleError(rc, "orc_interact_dot")
    return R


class Model(object):
    """Weights + wiring of one model, laid out for orc_forward.

    kind        : MODEL_DLRM | MODEL_WND | MODEL_NCF
    tables      : list of [rows, D] float32
    ln_bot/top  : width lists (ln_top[0] == num_int; for NCF the MLP-branch widths)
    bot/top     : lists of (W [m,n], b [m])
    final       : (W, b) of NCF's predictor or None
    """

    def __init__(self, kind, tables, ln_bot, bot, ln_top, top, interaction_op=INTERACT_CAT,
                 itself=False, sigmoid_top=-1, final=None, ln_task=None, tasks=None, ln_att=None, att=None,
                 rnn=None):
        """MT-WnD: ln_task = head widths, tasks = list (one per head) of lists of (W, b).
        DIN: ln_att = attention-unit widths, att = list (one per behaviour table) of lists of (W, b).
        DIEN: rnn = [i2h_w, i2h_b, gates_t_w, gates_t_b] of layer 1 then of layer 2 (8 arrays)."""
        self.kind = kind
        self.tables = [np.ascontiguousarray(t, dtype=np.float32) for t in tables]
        self.D = int(self.tables[0].shape[1])
        self.ln_bot = np.ascontiguousarray(ln_bot, dtype=np.int32)
        self.ln_top = np.ascontiguousarray(ln_top, dtype=np.int32)
        self.bot = [(np.ascontiguousarray(W, np.float32), np.ascontiguousarray(b, np.float32))
                    for W, b in bot]
        self.top = [(np.ascontiguousarray(W, np.float32), np.ascontiguousarray(b, np.float32))
                    for W, b in top]
        self.final = None if final is None else (
            np.ascontiguousarray(final[0], np.float32), np.ascontiguousarray(final[1], np.float32))
        self.interaction_op = interaction_op
        self.itself = bool(itself)
        self.sigmoid_top = int(sigmoid_top)
        self.rows = np.array([t.shape[0] for t in self.tables], dtype=np.int64)
        # keep ctypes arrays alive
        T = len(self.tables)
        self._tabp = (_f32p * T)(*[t.ctypes.data_as(_f32p) for t in self.tables])
        nb, nt = max(len(self.bot), 1), max(len(self.top), 1)
        self._bW = (_f32p * nb)(*[W.ctypes.data_as(_f32p) for W, _ in self.bot])
        self._bb = (_f32p * nb)(*[b.ctypes.data_as(_f32p) for _, b in self.bot])
        self._tW = (_f32p * nt)(*[W.ctypes.data_as(_f32p) for W, _ in self.top])
        self._tb = (_f32p * nt)(*[b.ctypes.data_as(_f32p) for _, b in self.top])
        m = _Model()
        m.model_kind, m.T, m.D = kind, T, self.D
        m.rows = self.rows.ctypes.data_as(_i64p)
        m.tables = self._tabp
        m.n_bot, m.ln_bot = self.ln_bot.size, self.ln_bot.ctypes.data_as(_i32p)
        m.bot_W, m.bot_b = self._bW, self._bb
        m.n_top, m.ln_top = self.ln_top.size, self.ln_top.ctypes.data_as(_i32p)
        m.top_W, m.top_b = self._tW, self._tb
        if self.final is not None:
            m.final_W = self.final[0].ctypes.data_as(_f32p)
            m.final_b = self.final[1].ctypes.data_as(_f32p)
            m.final_m = self.final[0].shape[0]
        m.interaction_op, m.itself, m.sigmoid_top = interaction_op, int(self.itself), self.sigmoid_top
        # weights transposed ONCE at model build ([K][N], what fc_impl's inner loop walks): a CPU
        # engine does not re-lay-out its weights per query
        self._botT = [np.ascontiguousarray(W.T) for W, _ in self.bot]
        self._topT = [np.ascontiguousarray(W.T) for W, _ in self.top]
        self._bWt = (_f32p * nb)(*[W.ctypes.data_as(_f32p) for W in self._botT])
        self._tWt = (_f32p * nt)(*[W.ctypes.data_as(_f32p) for W in self._topT])
        m.bot_Wt, m.top_Wt = self._bWt, self._tWt
        if self.final is not None:
            self._finT = np.ascontiguousarray(self.final[0].T)
            m.final_Wt = self._finT.ctypes.data_as(_f32p)
        self.tasks = None
        if tasks is not None:
            self.ln_task = np.ascontiguousarray(ln_task, dtype=np.int32)
            self.tasks = [[(np.ascontiguousarray(W, np.float32), np.ascontiguousarray(b, np.float32)) for W, b in head]
                          for head in tasks]
            flat = [wb for head in self.tasks for wb in head]
            self._kT = [np.ascontiguousarray(W.T) for W, _ in flat]
            nk = len(flat)
            self._kW = (_f32p * nk)(*[W.ctypes.data_as(_f32p) for W, _ in flat])
            self._kb = (_f32p * nk)(*[b.ctypes.data_as(_f32p) for _, b in flat])
            self._kWt = (_f32p * nk)(*[W.ctypes.data_as(_f32p) for W in self._kT])
            m.n_task, m.ln_task = self.ln_task.size, self.ln_task.ctypes.data_as(_i32p)
            m.num_tasks, m.task_sigmoid = len(self.tasks), self.sigmoid_top
            m.task_W, m.task_b, m.task_Wt = self._kW, self._kb, self._kWt
        self.att = None
        if att is not None:
            self.ln_att = np.ascontiguousarray(ln_att, dtype=np.int32)
            self.att = [[(np.ascontiguousarray(W, np.float32), np.ascontiguousarray(b, np.float32)) for W, b in unit]
                        for unit in att]
            flat = [wb for unit in self.att for wb in unit]
            na = len(flat)
            self._aW = (_f32p * na)(*[W.ctypes.data_as(_f32p) for W, _ in flat])
            self._ab = (_f32p * na)(*[b.ctypes.data_as(_f32p) for _, b in flat])
            m.n_att, m.ln_att = self.ln_att.size, self.ln_att.ctypes.data_as(_i32p)
            m.att_W, m.att_b = self._aW, self._ab
        self.rnn = None
        if rnn is not None:
            self.rnn = [np.ascontiguousarray(w, np.float32) for w in rnn]
            assert len(self.rnn) == 8
            m.rnn_hidden = int(self.rnn[2].shape[0])
            for i, w in enumerate(self.rnn):
                m.rnn_w[i] = w.ctypes.data_as(_f32p)
        self._c = m

    @property
    def n_out(self):
        if self.tasks is not None:
            return len(self.tasks) * int(self.ln_task[-1])
        return int(self.final[0].shape[0]) if self.final is not None else int(self.ln_top[-1])

    @property
    def num_int(self):
        if self.kind == MODEL_NCF:
            return self.D + int(self.ln_top[-1])
        if self.kind == MODEL_DIN:
            return 4 * self.D
        return int(self.ln_top[0])

    def forward(self, dense, idx, lengths, bs=None, nthreads=1, want_R=False):
        """dense [n, m_den] (None for NCF); idx: list of T int arrays (concatenated bags);
        lengths: list of T int32 arrays [n].  Returns out [bs, n_out] (and R)."""
        T = len(self.tables)
        lengths = [np.ascontiguousarray(l, dtype=np.int32) for l in lengths]
        idx = [np.ascontiguousarray(i, dtype=np.int64) for i in idx]
        n = lengths[0].size
        if bs is None:
            bs = n
        if bs < n:  # query = prefix of the staged batch (inferenceEngine.py:200-206)
            cut = [int(l[:bs].sum()) for l in lengths]
            idx = [np.ascontiguousarray(i[:c]) for i, c in zip(idx, cut)]
            lengths = [np.ascontiguousarray(l[:bs]) for l in lengths]
        if dense is not None:
            dense = np.ascontiguousarray(np.asarray(dense, dtype=np.float32)[:bs])
            dp = dense.ctypes.data_as(_f32p)
        else:
            dp = None
        n_idx = np.array([i.size for i in idx], dtype=np.int64)
        ip = (_i64p * T)(*[i.ctypes.data_as(_i64p) for i in idx])
        lp = (_i32p * T)(*[l.ctypes.data_as(_i32p) for l in lengths])
        out = np.empty((bs, self.n_out), dtype=np.float32)
        R = np.empty((bs, self.num_int), dtype=np.float32) if want_R else None
        rc = lib().orc_forward(C.byref(self._c), bs, dp, ip, n_idx.ctypes.data_as(_i64p), lp,
                               out.ctypes.data_as(_f32p),
                               R.ctypes.data_as(_f32p) if want_R else None, nthreads)
        if rc:
            raise OracleError(rc, "orc_forward")
        return (out, R) if want_R else out
