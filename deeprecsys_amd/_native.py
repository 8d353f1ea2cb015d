"""ctypes binding of include/drs.h (libdrs_hip.so) -- the only way into the kernels.

Fails loudly: a missing library raises at first use, and drs_create fails when no
HIP device is visible.  Nothing here (or anywhere in this package) falls back to a
CPU implementation.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdrs_hip.so")   # the one library this package binds (no override)

# status codes (include/drs.h)
OK, ERR_BAD_ARG, ERR_OOM, ERR_HIP, ERR_INDEX_RANGE, ERR_LENGTHS_SUM, ERR_STATE, ERR_UNSUPPORTED = \
    0, -1, -2, -3, -4, -5, -6, -7
MODEL_DLRM, MODEL_WND, MODEL_NCF, MODEL_MTWND, MODEL_DIN, MODEL_DIEN = 0, 1, 2, 3, 4, 5
INTERACT_DOT, INTERACT_CAT = 0, 1
ACT_NONE, ACT_RELU, ACT_SIGMOID = 0, 1, 2
MLP_BOT, MLP_TOP, MLP_FINAL = 0, 1, 2
MLP_TASK0 = 16     # + k: task head k (MT-WnD)
MLP_ATT0 = 1024    # + i: attention unit i (DIN)
MLP_RNN0, MLP_RNN1 = 32, 33   # the two BasicRNN layers (DIEN): layer 0 = i2h, layer 1 = gates_t
KERNEL_SLS, KERNEL_MLP, KERNEL_SLS_CLOCK = 0, 1, 2

_f32p = C.POINTER(C.c_float)
_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)

# every symbol include/drs.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("drs_abi_version", C.c_int32, []),
    ("drs_backend", C.c_char_p, []),
    ("drs_device_count", C.c_int32, [_i32p]),
    ("drs_last_error", C.c_char_p, [C.c_void_p]),
    ("drs_create", C.c_int32, [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p)]),
    ("drs_destroy", C.c_int32, [C.c_void_p]),
    ("drs_set_table", C.c_int32, [C.c_void_p, C.c_int32, _f32p, C.c_int64]),
    ("drs_fill_table_uniform", C.c_int32, [C.c_void_p, C.c_int32, C.c_float, C.c_float, C.c_uint64]),
    ("drs_set_fc", C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, _f32p, _f32p, C.c_int32, C.c_int32]),
    ("drs_stage_batch", C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, _f32p, C.POINTER(_i64p), _i64p,
                                    C.POINTER(_i32p)]),
    ("drs_forward", C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, _f32p]),
    ("drs_forward_async", C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]),
    ("drs_forward_multi_async", C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    ("drs_wait", C.c_int32, [C.c_void_p, C.c_int32, _f32p, C.c_int64]),
    ("drs_sync", C.c_int32, [C.c_void_p]),
    ("drs_forward_inputs_async", C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, _f32p, C.POINTER(_i64p), _i64p,
                                             C.POINTER(_i32p)]),
    ("drs_forward_inputs", C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, _f32p, C.POINTER(_i64p), _i64p,
                                       C.POINTER(_i32p), _f32p]),
    ("drs_run_queues_async", C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64,
                                         C.c_int64, C.c_void_p, C.c_int64]),
    ("drs_run_queues_multi_async", C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, _i32p, C.POINTER(C.c_void_p),
                                               C.POINTER(C.c_void_p), _i64p, _i64p, C.POINTER(C.c_void_p), _i64p]),
    ("drs_fetch_interaction", C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, _f32p]),
    ("drs_out_width", C.c_int32, [C.c_void_p, _i32p]),
    ("drs_interaction_width", C.c_int32, [C.c_void_p, _i32p]),
    ("drs_sls", C.c_int32, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p,
                            C.c_int64, C.c_int64, C.c_void_p, C.c_int32]),
    ("drs_fc", C.c_int32, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p,
                           C.c_int32, C.c_int32, C.c_void_p]),
    ("drs_interact_dot", C.c_int32, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32,
                                     C.c_int32, C.c_void_p]),
    ("drs_set_option", C.c_int32, [C.c_void_p, C.c_char_p, C.c_int64]),
    ("drs_get_option", C.c_int32, [C.c_void_p, C.c_char_p, _i64p]),
    ("drs_set_profiling", C.c_int32, [C.c_void_p, C.c_int32]),
    ("drs_kernel_time", C.c_int32, [C.c_void_p, C.c_int32, C.POINTER(C.c_double), _i64p]),
    ("drs_reset_kernel_time", C.c_int32, [C.c_void_p]),
    ("drs_kernel_bytes", C.c_int32, [C.c_void_p, C.c_int32, _i64p]),
    ("drs_debug_gather_stamps", C.c_int32, [C.c_void_p, C.c_int32, C.POINTER(C.c_uint64), C.c_int64, _i64p]),
    ("drs_gather_bytes", C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, _i64p]),
    ("drs_last_dispatch", C.c_int32, [C.c_void_p, C.c_int32, C.c_char_p, C.c_int64]),
    ("drs_comm_unique_id", C.c_int32, [C.POINTER(C.c_uint8)]),
    ("drs_comm_create", C.c_int32, [C.POINTER(C.c_uint8), C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    ("drs_comm_destroy", C.c_int32, [C.c_void_p]),
    ("drs_comm_barrier", C.c_int32, [C.c_void_p]),
    ("drs_stats_allreduce", C.c_int32, [C.c_void_p, _i64p, C.c_int32, C.POINTER(C.c_double)]),
    ("drs_comm_last_error", C.c_char_p, []),
]
COMM_ID_BYTES = 128


class ModelCfg(C.Structure):
    _fields_ = [
        ("model_kind", C.c_int32), ("num_tables", C.c_int32), ("table_rows", _i64p),
        ("sparse_dim", C.c_int32), ("n_bot", C.c_int32), ("ln_bot", _i32p),
        ("n_top", C.c_int32), ("ln_top", _i32p),
        ("interaction_op", C.c_int32), ("interaction_itself", C.c_int32),
        ("sigmoid_top", C.c_int32), ("max_batch", C.c_int32), ("max_lookups", C.c_int32),
        ("num_staged_batches", C.c_int32), ("num_slots", C.c_int32),
        ("n_task", C.c_int32), ("ln_task", _i32p), ("num_tasks", C.c_int32),
    ]


class DrsError(RuntimeError):
    def __init__(self, code, what, detail=""):
        super().__init__("%s failed with status %d%s" % (what, code, (": " + detail) if detail else ""))
        self.code = code
        self.detail = detail


_lib = None


def is_lab():
    """the lab build of the library (make -C deeprecsys_amd/csrc lab-lib; -DDRS_LAB) is bound: it also takes the lab's
    instruments and the options of the forms that lost their measurement (docs/OPTIONS.md, last section)"""
    return os.path.basename(LIB_PATH).startswith("libdrs_hip_lab")


def lib():
    """Load libdrs_hip.so (built by __graft_entry__.build() / csrc/Makefile)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "%s is missing: build it with `make -C deeprecsys_amd/csrc` (hipcc, gfx950). "
                "deeprecsys_amd has no CPU fallback." % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        backend = L.drs_backend() or b""
        if not backend.startswith(b"hip:"):
            raise ImportError("%s identifies itself as %r: deeprecsys_amd binds the HIP build only "
                              "(the CPU restatement of the ABI is test infrastructure)" % (LIB_PATH, backend))
        _lib = L
    return _lib


def device_count():
    n = C.c_int32(0)
    rc = lib().drs_device_count(C.byref(n))
    return int(n.value) if rc == OK else 0


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class Engine(object):
    """Thin object wrapper over a drs_handle (one GPU, one process)."""

    def __init__(self, kind, table_rows, sparse_dim, ln_bot, ln_top, interaction_op=INTERACT_CAT,
                 interaction_itself=False, sigmoid_top=-1, max_batch=1, max_lookups=1,
                 num_staged_batches=1, num_slots=1, device=0, ln_task=None, num_tasks=0):
        L = lib()
        self._rows = np.ascontiguousarray(table_rows, dtype=np.int64)
        self._ln_bot = np.ascontiguousarray(ln_bot, dtype=np.int32)
        self._ln_top = np.ascontiguousarray(ln_top, dtype=np.int32)
        cfg = ModelCfg(kind, self._rows.size, self._rows.ctypes.data_as(_i64p), int(sparse_dim),
                       self._ln_bot.size, self._ln_bot.ctypes.data_as(_i32p),
                       self._ln_top.size, self._ln_top.ctypes.data_as(_i32p),
                       int(interaction_op), int(bool(interaction_itself)), int(sigmoid_top),
                       int(max_batch), int(max_lookups), int(num_staged_batches), int(num_slots))
        if ln_task is not None:
            self._ln_task = np.ascontiguousarray(ln_task, dtype=np.int32)
            cfg.n_task, cfg.ln_task, cfg.num_tasks = self._ln_task.size, self._ln_task.ctypes.data_as(_i32p), int(num_tasks)
        h = C.c_void_p()
        rc = L.drs_create(C.byref(cfg), int(device), C.byref(h))
        if rc != OK:
            raise DrsError(rc, "drs_create", (L.drs_last_error(None) or b"").decode())
        self._h = h
        self.kind = kind
        self.T = int(self._rows.size)
        self.D = int(sparse_dim)
        self.max_batch = int(max_batch)
        # width of a query's dense rows (drs_create's shape algebra): DLRM / W&D / MT-WnD feed ln_bot[0]
        self.m_den = int(self._ln_bot[0]) if kind in (MODEL_DLRM, MODEL_WND, MODEL_MTWND) else 0
        self.num_slots = int(num_slots)
        self.device = int(device)
        self._marsh = ((C.c_int32 * 64)(), (C.c_int32 * 64)())
        self.user_options = set()        # option keys the caller set (set_option)
        # the per-slot record of the kernel forms a launch set took (last_dispatch) costs the host ~1 us per set: off
        # unless asked for -- DRS_DISPATCH_LOG=1 (tests/conftest.py sets it) or set_option("dispatch_log", 1)
        if os.environ.get("DRS_DISPATCH_LOG", "0") not in ("", "0"):
            self.set_option("dispatch_log", 1)

    # -- helpers ------------------------------------------------------------------
    def _check(self, rc, what):
        if rc != OK:
            raise DrsError(rc, what, (lib().drs_last_error(self._h) or b"").decode())

    def close(self):
        if getattr(self, "_h", None):
            lib().drs_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def n_out(self):
        n = C.c_int32(0)
        lib().drs_out_width(self._h, C.byref(n))
        return int(n.value)

    @property
    def num_int(self):
        n = C.c_int32(0)
        lib().drs_interaction_width(self._h, C.byref(n))
        return int(n.value)

    # -- parameters ---------------------------------------------------------------
    def set_table(self, t, W):
        W = _f32(W)
        self._check(lib().drs_set_table(self._h, t, W.ctypes.data_as(_f32p), W.shape[0]), "drs_set_table")

    def fill_table_uniform(self, t, lo, hi, seed):
        self._check(lib().drs_fill_table_uniform(self._h, t, lo, hi, seed), "drs_fill_table_uniform")

    def set_fc(self, mlp, layer, W, b):
        W, b = _f32(W), _f32(b)
        self._check(lib().drs_set_fc(self._h, mlp, layer, W.ctypes.data_as(_f32p),
                                     b.ctypes.data_as(_f32p), W.shape[0], W.shape[1]), "drs_set_fc")

    # -- inputs -------------------------------------------------------------------
    @staticmethod
    def _pack_sparse(idx, lengths):
        """-> (idx, lengths, n_idx, T pointers, T pointers).  Fast path: the reference's feeder
        hands run_queues 2-D arrays (ids [T, bs*L] int64, lengths [T, bs] int32, rows possibly
        strided slices of the pre-generated sets, inferenceEngine.py:200-206); their row pointers
        are base + t * stride -- no per-table Python objects on the per-query path."""
        if (isinstance(idx, np.ndarray) and idx.ndim == 2 and idx.dtype == np.int64 and idx.strides[1] == 8 and
                isinstance(lengths, np.ndarray) and lengths.ndim == 2 and lengths.dtype == np.int32 and
                lengths.strides[1] == 4 and lengths.shape[0] == idx.shape[0]):
            T = idx.shape[0]
            steps = np.arange(T, dtype=np.int64)
            ia = (steps * idx.strides[0] + idx.ctypes.data).astype(np.uint64)
            la = (steps * lengths.strides[0] + lengths.ctypes.data).astype(np.uint64)
            n_idx = np.full(T, idx.shape[1], dtype=np.int64)
            # (idx, ia) / (lengths, la): the caller keeps these alive across the C call
            return (idx, ia), (lengths, la), n_idx, ia.ctypes.data_as(C.POINTER(_i64p)), la.ctypes.data_as(C.POINTER(_i32p))
        idx = [np.ascontiguousarray(i, dtype=np.int64) for i in idx]
        lengths = [np.ascontiguousarray(l, dtype=np.int32) for l in lengths]
        T = len(idx)
        n_idx = np.array([i.size for i in idx], dtype=np.int64)
        ip = (_i64p * T)(*[i.ctypes.data_as(_i64p) for i in idx])
        lp = (_i32p * T)(*[l.ctypes.data_as(_i32p) for l in lengths])
        return idx, lengths, n_idx, ip, lp

    def stage_batch(self, batch_id, dense, idx, lengths):
        n = int(np.asarray(lengths[0]).size)
        idx, lengths, n_idx, ip, lp = self._pack_sparse(idx, lengths)
        dp = None
        if dense is not None:
            dense = _f32(dense)
            dp = dense.ctypes.data_as(_f32p)
        self._check(lib().drs_stage_batch(self._h, batch_id, n, dp, ip, n_idx.ctypes.data_as(_i64p), lp),
                    "drs_stage_batch")

    # -- hot path -----------------------------------------------------------------
    def forward(self, batch_id, bs):
        out = np.empty((bs, self.n_out), dtype=np.float32)
        self._check(lib().drs_forward(self._h, batch_id, bs, out.ctypes.data_as(_f32p)), "drs_forward")
        return out

    def forward_async(self, slot, batch_id, bs):
        self._check(lib().drs_forward_async(self._h, slot, batch_id, bs), "drs_forward_async")

    def forward_multi_async(self, slot, batch_ids, bss):
        """Coalesce several queries into one set of launches; wait(slot, sum(bss)) returns
        their outputs back to back."""
        # (two preallocated ctypes arrays: 0.7 us to fill against 6.5 us for two numpy conversions + pointer casts,
        #  on a call an engine process makes ~15 000 times a second)
        n = len(batch_ids)
        if n > 64 or n != len(bss):          # (the library enforces DRS_MAX_COALESCE itself)
            raise ValueError("one size per query, at most 64 entries")
        a = self._marsh
        a[0][:n] = batch_ids
        a[1][:n] = bss
        self._check(lib().drs_forward_multi_async(self._h, slot, n, a[0], a[1]), "drs_forward_multi_async")

    def wait(self, slot, bs=None):
        """bs = total samples submitted on the slot (sum over coalesced queries); the library
        checks the buffer against what is really in flight there."""
        if bs is None:
            self._check(lib().drs_wait(self._h, slot, None, 0), "drs_wait")
            return None
        out = np.empty((bs, self.n_out), dtype=np.float32)
        self._check(lib().drs_wait(self._h, slot, out.ctypes.data_as(_f32p), out.size), "drs_wait")
        return out

    def sync(self):
        self._check(lib().drs_sync(self._h), "drs_sync")

    def forward_inputs(self, dense, idx, lengths, bs, slot=0):
        if type(idx) is np.ndarray and idx.ndim == 2:
            self.forward_inputs_async(dense, idx, lengths, bs, slot=slot)
            return self.wait(slot, bs)
        idx, lengths, n_idx, ip, lp = self._pack_sparse(idx, lengths)
        dp = None
        if dense is not None:
            dense = _f32(dense)
            dp = dense.ctypes.data_as(_f32p)
        out = np.empty((bs, self.n_out), dtype=np.float32)
        self._check(lib().drs_forward_inputs(self._h, slot, bs, dp, ip, n_idx.ctypes.data_as(_i64p), lp,
                                             out.ctypes.data_as(_f32p)), "drs_forward_inputs")
        return out

    def forward_inputs_async(self, dense, idx, lengths, bs, slot=0):
        """Enqueue only; wait(slot, bs) returns the result.  The arrays are consumed (converted
        into the slot's pinned block) before this returns."""
        if (type(idx) is np.ndarray and idx.ndim == 2 and idx.dtype == np.int64 and idx.strides[1] == 8 and
                type(lengths) is np.ndarray and lengths.ndim == 2 and lengths.dtype == np.int32 and
                lengths.strides[1] == 4 and lengths.shape[0] == idx.shape[0] == self.T and
                (dense is None or (type(dense) is np.ndarray and dense.dtype == np.float32 and dense.flags.c_contiguous))):
            # the reference feeder's arrays as they are: two base pointers and two row strides
            # (the C side reads lengths[t][0 .. bs) and bs dense rows: check them here, a short array
            # must be a Python error, not a host out-of-bounds read)
            if bs < 0 or bs > lengths.shape[1]:
                raise ValueError("bs=%d but lengths has %d columns" % (bs, lengths.shape[1]))
            if dense is not None and (dense.ndim != 2 or dense.shape[0] < bs or dense.shape[1] != self.m_den):
                raise ValueError("bs=%d, dense width %d, but dense has shape %r" % (bs, self.m_den, dense.shape))
            self._check(lib().drs_run_queues_async(self._h, slot, bs, None if dense is None else dense.ctypes.data,
                                                   idx.ctypes.data, idx.strides[0] // 8, idx.shape[1],
                                                   lengths.ctypes.data, lengths.strides[0] // 4),
                        "drs_run_queues_async")
            return
        idx, lengths, n_idx, ip, lp = self._pack_sparse(idx, lengths)
        dp = None
        if dense is not None:
            dense = _f32(dense)
            dp = dense.ctypes.data_as(_f32p)
        self._check(lib().drs_forward_inputs_async(self._h, slot, bs, dp, ip, n_idx.ctypes.data_as(_i64p), lp),
                    "drs_forward_inputs_async")

    def run_queues_multi_async(self, queries, slot=0):
        """Several waiting requests as ONE launch set: `queries` is a list of (dense, ids, lengths, bs)
        with the arrays as the reference's feeder holds them (ids [T, n] int64, lengths [T, >= bs]
        int32, dense [>= bs, m_den] float32 or None; row slices of bigger arrays are fine).
        wait(slot, sum(bs)) returns the outputs back to back.  The arrays are consumed before this
        returns."""
        n = len(queries)
        bsa = np.empty(n, dtype=np.int32)
        ids_stride = np.empty(n, dtype=np.int64)
        len_stride = np.empty(n, dtype=np.int64)
        n_idx = np.empty(n, dtype=np.int64)
        dp = (C.c_void_p * n)()
        ip = (C.c_void_p * n)()
        lp = (C.c_void_p * n)()
        keep = []
        for i, (dense, idx, lengths, bs) in enumerate(queries):
            if not (type(idx) is np.ndarray and idx.ndim == 2 and idx.dtype == np.int64 and idx.strides[1] == 8):
                idx = np.ascontiguousarray(idx, dtype=np.int64)
            if not (type(lengths) is np.ndarray and lengths.ndim == 2 and lengths.dtype == np.int32 and lengths.strides[1] == 4):
                lengths = np.ascontiguousarray(lengths, dtype=np.int32)
            if idx.ndim != 2 or lengths.ndim != 2 or idx.shape[0] != self.T or lengths.shape[0] != self.T:
                raise ValueError("query %d: ids / lengths must be [T=%d, ...] arrays" % (i, self.T))
            if bs < 0 or bs > lengths.shape[1]:
                raise ValueError("query %d: bs=%d but lengths has %d columns" % (i, bs, lengths.shape[1]))
            if dense is not None:
                if not (type(dense) is np.ndarray and dense.dtype == np.float32 and dense.flags.c_contiguous):
                    dense = np.ascontiguousarray(dense, dtype=np.float32)
                if dense.ndim != 2 or dense.shape[0] < bs or dense.shape[1] != self.m_den:
                    raise ValueError("query %d: bs=%d, dense width %d, but dense has shape %r" % (i, bs, self.m_den, dense.shape))
            keep.append((dense, idx, lengths))
            bsa[i] = bs
            ids_stride[i] = idx.strides[0] // 8
            len_stride[i] = lengths.strides[0] // 4
            n_idx[i] = idx.shape[1]
            dp[i] = None if dense is None else dense.ctypes.data
            ip[i] = idx.ctypes.data
            lp[i] = lengths.ctypes.data
        self._check(lib().drs_run_queues_multi_async(self._h, slot, n, bsa.ctypes.data_as(_i32p), dp, ip,
                                                     ids_stride.ctypes.data_as(_i64p), n_idx.ctypes.data_as(_i64p),
                                                     lp, len_stride.ctypes.data_as(_i64p)),
                    "drs_run_queues_multi_async")

    def fetch_interaction(self, bs, slot=0):
        R = np.empty((bs, self.num_int), dtype=np.float32)
        self._check(lib().drs_fetch_interaction(self._h, slot, bs, R.ctypes.data_as(_f32p)),
                    "drs_fetch_interaction")
        return R

    # -- operator level (device pointers, e.g. torch.Tensor.data_ptr()) -------------
    def sls(self, d_W, rows, D, d_idx, d_len, n_bags, n_idx, d_out, exact_order=True):
        self._check(lib().drs_sls(self._h, d_W, rows, D, d_idx, d_len, n_bags, n_idx, d_out,
                                  int(bool(exact_order))), "drs_sls")

    def fc(self, d_x, M, K, d_W, d_b, N, act, d_y):
        self._check(lib().drs_fc(self._h, d_x, M, K, d_W, d_b, N, act, d_y), "drs_fc")

    def interact_dot(self, d_T, B, F, D, itself, d_R):
        self._check(lib().drs_interact_dot(self._h, d_T, B, F, D, int(bool(itself)), d_R),
                    "drs_interact_dot")

    # -- tuning / measurement -------------------------------------------------------
    def set_option(self, key, value, user=True):
        """user=False: a setting the host code makes on its own behalf (DLRM_Net.tune_table_placement) -- keys set with
        user=True are remembered in `user_options`, and the search leaves those alone."""
        self._check(lib().drs_set_option(self._h, key.encode(), int(value)), "drs_set_option")
        if user:
            self.user_options.add(key)

    def get_option(self, key):
        v = C.c_int64(0)
        self._check(lib().drs_get_option(self._h, key.encode(), C.byref(v)), "drs_get_option")
        return int(v.value)

    def kernel_bytes(self, kernel):
        """Algorithmic gather bytes of exactly the launches kernel_time(kernel) has timed."""
        b = C.c_int64(0)
        self._check(lib().drs_kernel_bytes(self._h, kernel, C.byref(b)), "drs_kernel_bytes")
        return int(b.value)

    def set_profiling(self, level):
        """0 off | 1 device clock stamps of the gather launch (free) | 2 also HIP events."""
        self._check(lib().drs_set_profiling(self._h, int(level)), "drs_set_profiling")

    def kernel_time(self, kernel):
        ms, n = C.c_double(0), C.c_int64(0)
        self._check(lib().drs_kernel_time(self._h, kernel, C.byref(ms), C.byref(n)), "drs_kernel_time")
        return float(ms.value), int(n.value)

    def reset_kernel_time(self):
        self._check(lib().drs_reset_kernel_time(self._h), "drs_reset_kernel_time")

    def gather_stamps(self, slot=0, cap=1 << 16):
        buf = np.zeros(2 * cap, dtype=np.uint64)
        n = C.c_int64(0)
        self._check(lib().drs_debug_gather_stamps(self._h, slot, buf.ctypes.data_as(C.POINTER(C.c_uint64)),
                                                  2 * cap, C.byref(n)), "drs_debug_gather_stamps")
        return buf[:2 * n.value].reshape(-1, 2)

    def last_dispatch(self, slot=0):
        """Which kernels served the launch set last enqueued on `slot`: list of "name<form>[...]" tokens."""
        buf = C.create_string_buffer(1024)
        self._check(lib().drs_last_dispatch(self._h, slot, buf, 1024), "drs_last_dispatch")
        import re
        return re.findall(r"[\w]+(?:<[^>]*>)?(?:\[[^\]]*\])?", buf.value.decode())

    def gather_bytes(self, batch_id, bs):
        b = C.c_int64(0)
        self._check(lib().drs_gather_bytes(self._h, batch_id, bs, C.byref(b)), "drs_gather_bytes")
        return int(b.value)


class Comm(object):
    """RCCL communicator behind the C ABI: the run's single collective (drs_stats_allreduce).
    rank 0 creates the id (Comm.unique_id()), the caller ships the 128 bytes to the other
    ranks, every rank then constructs Comm(id, rank, world, device)."""

    @staticmethod
    def unique_id():
        buf = (C.c_uint8 * COMM_ID_BYTES)()
        rc = lib().drs_comm_unique_id(buf)
        if rc != OK:
            raise DrsError(rc, "drs_comm_unique_id", (lib().drs_comm_last_error() or b"").decode())
        return bytes(buf)

    def __init__(self, uid, rank, world, device):
        buf = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(uid)
        h = C.c_void_p()
        rc = lib().drs_comm_create(buf, int(rank), int(world), int(device), C.byref(h))
        if rc != OK:
            raise DrsError(rc, "drs_comm_create", (lib().drs_comm_last_error() or b"").decode())
        self._h, self.rank, self.world = h, int(rank), int(world)

    def _check(self, rc, what):
        if rc != OK:
            raise DrsError(rc, what, (lib().drs_comm_last_error() or b"").decode())

    def barrier(self):
        self._check(lib().drs_comm_barrier(self._h), "drs_comm_barrier")

    def stats_allreduce(self, hist, sum_min_max):
        """hist int64[n] SUM; sum_min_max float64[4]: [0],[1] SUM, [2] MIN, [3] MAX.  Returns copies."""
        h = np.ascontiguousarray(hist, dtype=np.int64).copy()
        s = np.ascontiguousarray(sum_min_max, dtype=np.float64).copy()
        assert s.size == 4
        self._check(lib().drs_stats_allreduce(self._h, h.ctypes.data_as(_i64p), h.size,
                                              s.ctypes.data_as(C.POINTER(C.c_double))), "drs_stats_allreduce")
        return h, s

    def close(self):
        if getattr(self, "_h", None):
            lib().drs_comm_destroy(self._h)
            self._h = None
