"""Model objects of the hot path: host-side mirrors of the reference's graph builders.

    DLRM_Net / DLRM_Wrapper               <- models/dlrm_s_caffe2.py:79-569
    Wide_and_Deep / Wide_and_Deep_Wrapper <- models/wide_and_deep.py:165-477
    NCF / NCF_Wrapper                     <- models/ncf.py:140-523
    MT_Wide_and_Deep / ..._Wrapper        <- models/multi_task_wnd.py:160-420
    DIN_Net / DIN_Wrapper                 <- models/din.py:24-470
    DIEN_Net / DIEN_Wrapper               <- models/dien.py:28-560

Same constructor arguments, the same shape algebra and sys.exit() checks, the same
numpy RNG consumption order for the weights (embeddings, then bottom MLP, then top
MLP: models/dlrm_s_caffe2.py:245-249,297-299,367-386), the same
create()/run()/run_queues() calls.  Where the reference emits Caffe2 operators into
a net, these classes fill a drs_model_cfg and hand the weights to libdrs_hip.so
(include/drs.h); the forward itself is the HIP kernels -- there is no Python or CPU
implementation of the arithmetic in this package.

Differences that are deliberate:
  * run_queues() executes the query synchronously (the reference enqueues blobs for
    a second thread blocked in RunNet; HIP stream order replaces the BlobsQueues);
  * fetch_output() replaces workspace.FetchBlob("prob_click");
  * stage_batches()/run_staged() keep all `num_batches` input sets resident in HBM
    (the reference keeps lX/lS_l/lS_i in process memory, inferenceEngine.py:83) so a
    request is served from (batch_id, batch_size) alone.
"""
import sys
import time

import numpy as np

from . import _native as N


def _ints(s):
    return np.array([int(x) for x in str(s).split("-")], dtype=int)


def _init_table(n, m):
    # models/dlrm_s_caffe2.py:297-299
    return np.random.uniform(low=-np.sqrt(1 / n), high=np.sqrt(1 / n), size=(n, m)).astype(np.float32)


def _init_mlp(ln):
    # models/dlrm_s_caffe2.py:240-249: W ~ N(0, sqrt(2/(m+n))) [m, n]; b ~ N(0, sqrt(1/m)) [m]
    layers = []
    for i in range(1, ln.size):
        n, m = int(ln[i - 1]), int(ln[i])
        W = np.random.normal(0.0, np.sqrt(2 / (m + n)), size=(m, n)).astype(np.float32)
        b = np.random.normal(0.0, np.sqrt(1 / m), size=m).astype(np.float32)
        layers.append((W, b))
    return layers


class _HipNet(object):
    """State shared by the three model mirrors: weights on the host until create(),
    then one Engine (one GPU)."""

    kind = None

    def _common_init(self, cli_args):
        self.args = cli_args
        self.accel_en = getattr(cli_args, "use_accel", False)
        self.engine = None
        self._out = None
        self._device = int(getattr(cli_args, "_drs_device", 0))
        self._table_init = getattr(cli_args, "accel_table_init", "numpy")
        self.tout = "prob_click"

    # -- weights ------------------------------------------------------------------
    def _make_tables(self, m, ln_emb):
        if self._table_init == "device":
            return [None] * ln_emb.size        # filled on the GPU in create()
        return [_init_table(int(n), m) for n in ln_emb]

    # -- device -------------------------------------------------------------------
    def _num_slots(self):
        """--accel_slots n launch sets in flight; 0 (default): what the engine asks for (drs_get_option
        "preferred_slots": 3 -- gather | MLP | enqueue -- for the gather-bound models, 6 for the MLP-bound ones, whose
        sets are chains of MFMA-bound launches that overlap each other; 4 for DIEN, MT-WnD and W&D, two sets at a time).  The engine decides the class from the model's
        shapes at creation: this is the first guess, _build_engine re-creates the (still empty) engine when it differs."""
        req = int(getattr(self.args, "accel_slots", 0) or 0)
        return req if req > 0 else (3 if self.kind in (N.MODEL_DLRM, N.MODEL_DIN) else 4 if self.kind in (N.MODEL_DIEN, N.MODEL_MTWND, N.MODEL_WND) else 6)

    def _build_engine(self, ln_bot_cfg, ln_top_cfg, interaction_op, itself, sigmoid_top, ln_task=None, num_tasks=0):
        a = self.args
        n_stage = max(int(getattr(a, "num_batches", 0)), 1)
        max_batch = max(int(getattr(a, "max_mini_batch_size", 1)), int(getattr(a, "mini_batch_size", 1)), 1)
        def make(n_slots):
            return N.Engine(self.kind, self.ln_emb, self.m_spa, ln_bot_cfg, ln_top_cfg,
                            interaction_op=interaction_op, interaction_itself=itself,
                            sigmoid_top=sigmoid_top, max_batch=max_batch,
                            max_lookups=max(int(a.num_indices_per_lookup), 1),
                            num_staged_batches=n_stage,
                            num_slots=n_slots, device=self._device,
                            ln_task=ln_task, num_tasks=num_tasks)
        eng = make(self._num_slots())
        if int(getattr(a, "accel_slots", 0) or 0) <= 0 and eng.get_option("preferred_slots") != eng.num_slots:
            # (an MLP-bound DLRM such as RM3: the class follows from the shapes, which the engine has just read)
            want = int(eng.get_option("preferred_slots"))
            eng.close()
            eng = make(want)
            if eng.get_option("preferred_slots") != eng.num_slots:
                raise RuntimeError("accel_slots 0: created %d slots, the engine prefers %d" % (
                    eng.num_slots, eng.get_option("preferred_slots")))
        # A/B aid for runs through the queue harness: DRS_ENGINE_OPTS="key=value,key=value"
        import os
        for kv in filter(None, os.environ.get("DRS_ENGINE_OPTS", "").split(",")):
            k, v = kv.split("=")
            eng.set_option(k.strip(), int(v))
        seed = int(getattr(a, "numpy_rand_seed", 0))
        for t, W in enumerate(self.emb_w):
            if W is None:
                n = int(self.ln_emb[t])
                eng.fill_table_uniform(t, -float(np.sqrt(1 / n)), float(np.sqrt(1 / n)), seed)
            else:
                eng.set_table(t, W)
        return eng

    def create(self, X, S_lengths, S_indices, T, id_qs=None, len_qs=None):
        """Build the device model (reference: create_input + create_model,
        models/dlrm_s_caffe2.py:509-546).  X/S_* are the first input set; they are kept as
        the "current blobs" exactly like the reference's initial FeedBlobs."""
        self._create_engine()
        self._cur_inputs = (X, S_lengths, S_indices)

    def parameters(self):
        return self

    # -- execution ----------------------------------------------------------------
    def run(self, X=None, S_lengths=None, S_indices=None, enable_prof=False):
        """One forward of a fed batch; returns the time at which input hand-over ended,
        like the reference's run() (models/dlrm_s_caffe2.py:549-569)."""
        if X is None and S_indices is None:
            X, S_lengths, S_indices = self._cur_inputs
        else:
            self._cur_inputs = (X, S_lengths, S_indices)
        bs = len(S_lengths[0])
        if getattr(self, "split_load", False):
            # stand-alone runs (main() below): the reference's run() feeds the blobs, takes the time, then
            # runs the net (models/dlrm_s_caffe2.py:551-568) -- "data loading" vs "computation".  Same split
            # here: narrow + ENFORCE-check + copy the inputs into a resident input set (synchronous), take the
            # time, then the forward on resident inputs.
            self.engine.stage_batch(0, X, S_indices, S_lengths)
            load_time = time.time()
            if enable_prof:
                self._run_profiled(None, None, None, bs, staged=0)
            else:
                self._out = self.engine.forward(0, bs)
            return load_time
        load_time = time.time()
        if enable_prof:
            self._run_profiled(X, S_lengths, S_indices, bs)
        else:
            self._out = self.engine.forward_inputs(X, S_indices, S_lengths, bs)
        return load_time

    def _run_profiled(self, X, S_lengths, S_indices, bs, staged=None):
        """--enable_profiling: the reference runs `workspace.C.benchmark_net(net, 0, 1, True)`
        (models/dlrm_s_caffe2.py:565-566), whose per-operator-type table
        experiments/operator_breakdown/sweep_p.py:21-28 parses (`<ms> ms. <pct>%. <OpType>`,
        value = field 0, type = field 3).  Here a query is two kinds of launches, timed with HIP
        events on the streams they run on: the multi-table gather (= every SparseLengthsSum of the
        graph) and the MLP launches (= every FC with its fused Relu / Sigmoid, the Concat /
        BatchMatMul / BatchGather interaction and the Cast, which have no launch of their own)."""
        eng = self.engine
        eng.reset_kernel_time()
        eng.set_profiling(2)
        try:
            self._out = eng.forward_inputs(X, S_indices, S_lengths, bs) if staged is None else eng.forward(staged, bs)
        finally:
            eng.set_profiling(0)
        sls_ms, _ = eng.kernel_time(N.KERNEL_SLS)
        mlp_ms, _ = eng.kernel_time(N.KERNEL_MLP)
        tot = max(sls_ms + mlp_ms, 1e-12)
        print("Time per operator type:")
        for ms, name in sorted(((sls_ms, "SparseLengthsSum"), (mlp_ms, "FC")), reverse=True):
            print("%16.6g ms. %10.5g%%. %s" % (ms, 100.0 * ms / tot, name))
        print("%16.6g ms in Total" % tot)
        sys.stdout.flush()

    def run_queued(self, ids, lengths, fc, batch_size):
        # 2-D arrays (what the reference's feeder passes) go down as they are: row pointers only
        if not (isinstance(ids, np.ndarray) and isinstance(lengths, np.ndarray)):
            ids, lengths = list(ids), list(lengths)
        self._out = self.engine.forward_inputs(fc, ids, lengths, int(batch_size))
        return self._out

    def run_queued_multi(self, requests, slot=0):
        """run_queues for every request an engine found waiting in its queue, as ONE launch set
        (drs_run_queues_multi_async): `requests` is a list of (ids, lengths, fc, batch_size) in
        run_queues' argument order; returns the list of their [bs_i, n_out] outputs."""
        no_dense = self.engine.m_den == 0
        qs = [(None if no_dense else fc, np.asarray(ids) if not isinstance(ids, np.ndarray) else ids,
               np.asarray(lengths) if not isinstance(lengths, np.ndarray) else lengths, int(bs))
              for ids, lengths, fc, bs in requests]
        self.engine.run_queues_multi_async(qs, slot=slot)
        sizes = [q[3] for q in qs]
        out = self.engine.wait(slot, sum(sizes))
        outs = np.split(out, np.cumsum(sizes)[:-1], axis=0)
        self._out = outs[-1]
        return outs

    def stage_batches(self, lX, lS_l, lS_i):
        for j in range(len(lS_l)):
            self.engine.stage_batch(j, None if lX is None else lX[j], lS_i[j], lS_l[j])
        self._n_staged = len(lS_l)

    def tune_table_placement(self, candidates=12, sets=128, spacer_gb=None, policies=(1, 0), max_extra_gb=48, sharers=1):
        """Where the tables live in HBM, and with which cache policy their rows are read, moves the many-rows-per-bag
        gather by up to 9 % -- a property of the PHYSICAL memory (it follows the memory through address changes;
        gigabytes-wide regions of HBM are "fast" or "slow", and they are different regions for non-temporal and for
        plain loads) that no synthetic probe sees, only the model's own launch sets (DESIGN.md 5,
        profiles/r05_placement/README.md).  With the input sets staged, this times full launch sets of the engine's
        preferred size (served as they will be: pipelined, the gather beside the previous set's MLP launch) on the arena drs_create made, under each load policy ("sls_nt"
        1 / 0), then on up to `candidates` - 1 further arenas taken from further on in HBM (`spacer_gb` of untouched
        memory between two candidates, default = the arena's size), and keeps the fastest (arena, policy) -- the first
        arena unless another is at least 1 % faster.  It stops as soon as one arena's best reading is 8 % under another's (the
        fast level is reached).  Every other arena and the spacers are released before it returns: one copy of the
        tables, nothing held.  Returns {"gather_us": [[nt, plain], ...], "kept": k, "sls_nt": p, ...} or
        None when the engine has nothing to time.  ~70 ms per candidate.
        While it runs every candidate and spacer stays allocated (a freed loser's pages would be handed out again as the
        next candidate): the search's transient footprint is bounded by `max_extra_gb` / `sharers` (engine processes
        that share the GPU) beside the engine's own guards (a quarter of the free memory per copy, 8 GB left free).
        A load policy the caller set with set_option is kept (only the arena is searched); "table_alloc" is restored."""
        eng = self.engine
        nb = int(getattr(self, "_n_staged", 0))
        if nb < 1 or candidates < 1:
            return None
        # only where it was measured to pay: gather-bound DLRM (incl. the in-between class, e.g. dlrm_rm1.json, whose
        # 128-byte rows read 10 % faster with plain loads when the tables are small) and DIN.  The MLP-bound shapes have
        # nothing to gain (their gather is a tenth of a set), the one-lookup models' tables are cache resident.
        # (round 5: DIN as well -- its fused gather + attention launch is its set's longest kernel; "din_nt" is its policy knob)
        if self.kind not in (N.MODEL_DLRM, N.MODEL_DIN) or not int(eng.get_option("gather_bound")):
            return None
        nt_key = "din_nt" if self.kind == N.MODEL_DIN else "sls_nt"
        co = max(1, min(int(eng.get_option("preferred_coalesce")), 16))
        bs = int(eng.max_batch)
        nt0 = int(eng.get_option(nt_key))
        ta0 = int(eng.get_option("table_alloc"))
        if nt_key in getattr(eng, "user_options", ()):     # the caller pinned the policy: search the arenas only
            policies = (nt0,)
        budget_gb = float(max_extra_gb) / max(1, int(sharers))

        # The sets are timed AS THEY WILL BE SERVED: the engine's stream mode, `preferred_slots` sets in flight, the gather
        # launches stamped by their own workgroups while the previous set's MLP launch runs beside them.  (Until the end of
        # round 5 they ran on one stream, the gather alone -- and an (arena, policy) that read 81.8 us alone took 87.1 us beside
        # the MLP launch where the first arena with plain loads took 82.8: profiles/r05_placement/README.md.)
        n_slots = max(1, min(int(eng.get_option("preferred_slots")), int(getattr(eng, "num_slots", 1))))

        def gather_us():
            busy = [False] * n_slots
            try:
                for phase, n_sets in (("warm", max(8, sets // 4)), ("timed", sets)):
                    if phase == "timed":
                        for s_ in range(n_slots):
                            if busy[s_]:
                                eng.wait(s_)
                                busy[s_] = False
                        eng.reset_kernel_time()
                        eng.set_profiling(1)
                    for g in range(n_sets):
                        s_ = g % n_slots
                        if busy[s_]:
                            eng.wait(s_)
                        eng.forward_multi_async(s_, [(g * co + k) % nb for k in range(co)], [bs] * co)
                        busy[s_] = True
                for s_ in range(n_slots):
                    if busy[s_]:
                        eng.wait(s_)
                        busy[s_] = False
                eng.set_profiling(0)
                ms, n = eng.kernel_time(N.KERNEL_SLS_CLOCK)
            finally:
                eng.set_profiling(0)
            return ms / n * 1e3 if n else None

        def both():
            out = []
            for p in policies:
                eng.set_option(nt_key, p, user=False)
                out.append(gather_us())
            return out

        times, best = [], None          # best = (us, arena index, policy)
        import time as _time
        t_begin = _time.perf_counter()
        try:
            t = both()
            if t[0] is None:
                return None
            times.append(t)
            size_gb = max(1, -(-int(eng.get_option("table_bytes")) // (1 << 30)))
            gap = size_gb if spacer_gb is None else int(spacer_gb)
            eng.set_option("table_alloc", 0, user=False)  # (plain hipMalloc candidates: arenas of the virtual-memory API read as
            for _ in range(1, candidates):              #  fast alone but measured 2 % slower beside two MLP streams, dlrm_rm1.json)
                if len(times) * (size_gb + max(gap, 0)) + size_gb > budget_gb:
                    break                               # the next candidate (and its spacer) would pass the footprint bound
                per_arena = [min(tt) for tt in times]
                if min(per_arena) <= 0.92 * max(per_arena):
                    break                               # an arena 8 % under the slowest one: the fast level has been seen (beside the MLP
                                                        # launch the levels read 90 / 88 / 85-86 / 82 us on RMC1; stopping at 5 % kept an 84.6)
                try:
                    if gap > 0:
                        eng.set_option("table_spacer", gap << 30, user=False)
                    eng.set_option("table_placement", -1, user=False)
                except N.DrsError:
                    break                               # no room for one more copy: the ones so far compete
                t = both()
                if any(u is None for u in t):
                    break                               # (nothing was timed on this one: it does not compete)
                times.append(t)
            flat = [(u, k, policies[i]) for k, tt in enumerate(times) for i, u in enumerate(tt)]
            best = min(flat)
            # the arena drs_create made stays unless another one is at least 1 % faster (timed as served, the readings repeat
            # to +-0.2 %; at 2 % a run kept an 84.6 us arena with 83.5 us ones on the list)
            first = min((u, 0, policies[i]) for i, u in enumerate(times[0]))
            if best[1] != 0 and best[0] > 0.99 * first[0]:
                best = first
            return {"gather_us": [[round(u, 2) for u in tt] for tt in times], "timed": "pipelined, %d sets in flight" % n_slots, "policies": ["nt" if p else "plain" for p in policies],
                    "kept": best[1], "sls_nt": best[2], "candidates": len(times), "losers": "freed",
                    "seconds": round(_time.perf_counter() - t_begin, 2)}
        except N.DrsError:
            return None                                 # (staged sets smaller than a full batch, ...: serve from where it is)
        finally:
            try:
                eng.set_option("table_alloc", ta0, user=False)
                eng.set_option(nt_key, best[2] if best else nt0, user=False)
                if best:
                    eng.set_option("table_placement", best[1], user=False)
                eng.set_option("table_placement", -2, user=False)
            except N.DrsError:
                pass

    def run_staged(self, batch_id, batch_size):
        self._out = self.engine.forward(int(batch_id), int(batch_size))
        return self._out

    def run_staged_multi(self, batch_ids, batch_sizes, slot=0):
        """Several queued queries in one set of launches (drs_forward_multi_async); returns
        the list of their [bs_i, n_out] outputs."""
        sizes = [int(b) for b in batch_sizes]
        self.engine.forward_multi_async(slot, [int(b) for b in batch_ids], sizes)
        out = self.engine.wait(slot, sum(sizes))
        cuts = np.cumsum(sizes)[:-1]
        outs = np.split(out, cuts, axis=0)
        self._out = outs[-1]
        return outs

    def submit_staged_multi(self, batch_ids, batch_sizes, slot):
        """Enqueue one set of launches on `slot` and return at once; collect_staged_multi(slot,
        batch_sizes) hands the outputs over.  With several slots the engine process keeps the
        gather of one set, the MLP of the previous one and the host work of the next in flight
        at the same time."""
        self.engine.forward_multi_async(slot, [int(b) for b in batch_ids], [int(b) for b in batch_sizes])

    def collect_staged_multi(self, batch_sizes, slot):
        sizes = [int(b) for b in batch_sizes]
        out = self.engine.wait(slot, sum(sizes))
        outs = np.split(out, np.cumsum(sizes)[:-1], axis=0)
        self._out = outs[-1]
        return outs

    def fetch_output(self):
        """FetchBlob("prob_click") counterpart: [bs, n_out] float32 of the last run."""
        return self._out


# =====================================================================================
class DLRM_Net(_HipNet):
    kind = N.MODEL_DLRM

    def __init__(self, cli_args, model=None, tag=None, enable_prof=False, id_qs=None, len_qs=None,
                 fc_q=None):
        self._common_init(cli_args)
        # shape algebra + checks of models/dlrm_s_caffe2.py:403-440
        ln_bot = _ints(cli_args.arch_mlp_bot)
        m_den = ln_bot[0]
        m_spa = cli_args.arch_sparse_feature_size
        ln_emb = _ints(cli_args.arch_embedding_size)
        num_fea = ln_emb.size + 1
        m_den_out = ln_bot[ln_bot.size - 1]
        if cli_args.arch_interaction_op == "dot":
            if cli_args.arch_interaction_itself:
                num_int = (num_fea * (num_fea + 1)) // 2 + m_den_out
            else:
                num_int = (num_fea * (num_fea - 1)) // 2 + m_den_out
        elif cli_args.arch_interaction_op == "cat":
            num_int = num_fea * m_den_out
        else:
            sys.exit("ERROR: --arch-interaction-op=" + cli_args.arch_interaction_op
                     + " is not supported")
        ln_top = _ints(str(num_int) + "-" + cli_args.arch_mlp_top)
        if m_spa != m_den_out:
            sys.exit("ERROR: arch-sparse-feature-size " + str(m_spa)
                     + " does not match last dim of bottom mlp " + str(m_den_out))
        if num_int != ln_top[0]:
            sys.exit("ERROR: # of feature interactions " + str(num_int)
                     + " does not match first dim of top mlp " + str(ln_top[0]))
        self.m_spa, self.ln_emb, self.ln_bot, self.ln_top = m_spa, ln_emb, ln_bot, ln_top
        self.arch_interaction_op = cli_args.arch_interaction_op
        self.arch_interaction_itself = cli_args.arch_interaction_itself
        self.sigmoid_bot = -1
        self.sigmoid_top = ln_top.size - 1
        # create_sequential_forward_ops order (:367-386): embeddings, bottom, top
        self.emb_w = self._make_tables(m_spa, ln_emb)
        self.bot_w = _init_mlp(ln_bot)
        self.top_w = _init_mlp(ln_top)

    def _create_engine(self):
        op = N.INTERACT_DOT if self.arch_interaction_op == "dot" else N.INTERACT_CAT
        self.engine = self._build_engine(self.ln_bot, self.ln_top, op, self.arch_interaction_itself,
                                         self.sigmoid_top)
        for i, (W, b) in enumerate(self.bot_w):
            self.engine.set_fc(N.MLP_BOT, i, W, b)
        for i, (W, b) in enumerate(self.top_w):
            self.engine.set_fc(N.MLP_TOP, i, W, b)

    def tril_indices(self):
        """interaction_tril_indices blob (models/dlrm_s_caffe2.py:529-535)."""
        offset = 1 if self.arch_interaction_itself else 0
        num_fea = self.ln_emb.size + 1
        return np.array([j + i * num_fea for i in range(num_fea) for j in range(i + offset)])


class Wide_and_Deep(_HipNet):
    kind = N.MODEL_WND

    def __init__(self, cli_args, model=None, tag=None, enable_prof=False, id_qs=None, len_qs=None,
                 fc_q=None):
        self._common_init(cli_args)
        # models/wide_and_deep.py:307-346
        if cli_args.arch_interaction_op != "cat":
            sys.exit("ERROR: sparse and dense features must be concatenated in wide and deep")
        ln_bot = _ints(cli_args.arch_mlp_bot)
        if ln_bot.size != 1:
            sys.exit("ERROR: wide and deep has no MLP layers for the continuous features")
        m_spa = cli_args.arch_sparse_feature_size
        ln_emb = _ints(cli_args.arch_embedding_size)
        num_fea = ln_emb.size + 1
        num_int = (num_fea - 1) * int(m_spa) + int(ln_bot[0])
        ln_top = _ints(str(num_int) + "-" + cli_args.arch_mlp_top)
        self.m_spa, self.ln_emb, self.ln_bot, self.ln_top = m_spa, ln_emb, ln_bot, ln_top
        self.arch_interaction_op = "cat"
        self.arch_interaction_itself = cli_args.arch_interaction_itself
        self.sigmoid_bot = -1
        self.sigmoid_top = ln_top.size - 1
        self.emb_w = self._make_tables(m_spa, ln_emb)     # :282-287
        self.top_w = _init_mlp(ln_top)                    # :299-302

    def _create_engine(self):
        self.engine = self._build_engine(self.ln_bot, self.ln_top, N.INTERACT_CAT, False, self.sigmoid_top)
        for i, (W, b) in enumerate(self.top_w):
            self.engine.set_fc(N.MLP_TOP, i, W, b)


class MT_Wide_and_Deep(Wide_and_Deep):
    """Multi-task W&D (models/multi_task_wnd.py): the W&D trunk with an all-ReLU shared top MLP
    (:301 passes sigmoid_layer -1), then `num_multi_tasks` task heads of widths arch_mlp_tasks
    over its output, each with Sigmoid on the layer index the reference calls sigmoid_top
    (= ln_top.size - 1, :399,309).  The reference builds and runs every head but keeps only the
    last as `last_output` (:316); here the heads' outputs come back side by side,
    [bs, num_tasks * ln_task[-1]] (the reference's last_output is the last ln_task[-1] columns)."""
    kind = N.MODEL_MTWND

    def __init__(self, cli_args, model=None, tag=None, enable_prof=False, id_qs=None, len_qs=None,
                 fc_q=None):
        # weights in the reference's order: embeddings, shared top, then each head (:286-312)
        Wide_and_Deep.__init__(self, cli_args, model, tag, enable_prof)
        self.ln_task = _ints(cli_args.arch_mlp_tasks)
        if self.ln_top[-1] != self.ln_task[0]:
            sys.exit("ERROR: Shared top layer and task MLP layers must have same input/output dimension")
        self.num_tasks = int(cli_args.num_multi_tasks)
        self.task_w = [_init_mlp(self.ln_task) for _ in range(self.num_tasks)]

    def _create_engine(self):
        self.engine = self._build_engine(self.ln_bot, self.ln_top, N.INTERACT_CAT, False, self.sigmoid_top,
                                         ln_task=self.ln_task, num_tasks=self.num_tasks)
        for i, (W, b) in enumerate(self.top_w):
            self.engine.set_fc(N.MLP_TOP, i, W, b)
        for k, head in enumerate(self.task_w):
            for i, (W, b) in enumerate(head):
                self.engine.set_fc(N.MLP_TASK0 + k, i, W, b)

class _NoDenseNet(_HipNet):
    """Models whose query is sparse features only (NCF, DIN): the dense argument of the
    reference's signatures is accepted and ignored, as the reference ignores it."""

    def run(self, X=None, S_lengths=None, S_indices=None, enable_prof=False):
        if S_indices is None:
            X, S_lengths, S_indices = self._cur_inputs
        else:
            self._cur_inputs = (X, S_lengths, S_indices)
        bs = len(S_lengths[0])
        if getattr(self, "split_load", False):      # stand-alone runs: see _HipNet.run
            self.engine.stage_batch(0, None, S_indices, S_lengths)
            load_time = time.time()
            if enable_prof:
                self._run_profiled(None, None, None, bs, staged=0)
            else:
                self._out = self.engine.forward(0, bs)
            return load_time
        load_time = time.time()
        if enable_prof:     # the reference runs benchmark_net for these models too (sweep_p.py parses the table)
            self._run_profiled(None, S_lengths, S_indices, bs)
        else:
            self._out = self.engine.forward_inputs(None, S_indices, S_lengths, bs)
        return load_time

    def run_queued(self, ids, lengths, fc, batch_size):
        self._out = self.engine.forward_inputs(None, list(ids), list(lengths), int(batch_size))
        return self._out

    def stage_batches(self, lX, lS_l, lS_i):
        for j in range(len(lS_l)):
            self.engine.stage_batch(j, None, lS_i[j], lS_l[j])
        self._n_staged = len(lS_l)


class NCF(_NoDenseNet):
    kind = N.MODEL_NCF

    def __init__(self, cli_args, model=None, tag=None, enable_prof=False, id_qs=None, len_qs=None):
        self._common_init(cli_args)
        # models/ncf.py:348-392
        if cli_args.arch_interaction_op != "cat":
            sys.exit("ERROR: sparse and dense features must be concatenated in NCF")
        ln_emb = _ints(cli_args.arch_embedding_size)
        if ln_emb.size != 4:
            sys.exit("ERROR: NCF only has 4 embedding tables")
        if cli_args.num_indices_per_lookup != 1:
            sys.exit("ERROR: NCF has 1 lookup per table")
        m_spa = cli_args.arch_sparse_feature_size
        num_int = 2 * int(m_spa)
        ln_top = _ints(str(num_int) + "-" + cli_args.arch_mlp_top)
        self.m_spa, self.ln_emb, self.ln_top = m_spa, ln_emb, ln_top
        self.ln_bot = np.array([0], dtype=int)
        self.arch_interaction_op = "cat"
        self.arch_interaction_itself = cli_args.arch_interaction_itself
        # create_emb: MF tables 0,1 then MLP tables 2,3 (:198-299); MLP over ln_top[:-1]
        # (:332-333); predictor [m_spa + ln_top[-2], ln_top[-1]] (:341-345); all Relu
        self.emb_w = self._make_tables(m_spa, ln_emb)
        self.top_w = _init_mlp(ln_top[:-1])
        self.final_w = _init_mlp(np.array([int(m_spa) + int(ln_top[-2]), int(ln_top[-1])]))

    def _create_engine(self):
        self.engine = self._build_engine(np.array([0]), self.ln_top[:-1], N.INTERACT_CAT, False, -1)
        for i, (W, b) in enumerate(self.top_w):
            self.engine.set_fc(N.MLP_TOP, i, W, b)
        W, b = self.final_w[0]
        self.engine.set_fc(N.MLP_FINAL, 0, W, b)


class DIN_Net(_NoDenseNet):
    """Deep Interest Network (models/din.py:247-390).  Tables = [user profile | the behaviour
    tables | candidate ad | context] (utils.cli has already replicated the behaviour table
    `user_behavior_tables` times, utils/utils.py:132-149).  Per behaviour table one attention unit
    with its OWN MLP 3*D -> arch_mlp_bot -> D over Concat(u_i, ad, u_i + ad) (:247-285);
    atten_out = Sum of the units; top MLP over Concat(profile, atten_out, ad, context)
    (:311-323); every activation is ReLU (the reference passes no sigmoid layer, :274,323)."""
    kind = N.MODEL_DIN

    def __init__(self, cli_args, model=None, tag=None, enable_prof=False, id_qs=None, len_qs=None):
        self._common_init(cli_args)
        m_spa = int(cli_args.arch_sparse_feature_size)
        ln_emb = _ints(cli_args.arch_embedding_size)
        if ln_emb.size < 4:                         # the reference asserts (:356)
            sys.exit("ERROR: DIN needs user profile, user behavior, candidate ad and context tables")
        num_int = 4 * m_spa
        ln_top = _ints(str(num_int) + "-" + cli_args.arch_mlp_top)
        self.ln_att = _ints(str(3 * m_spa) + "-" + cli_args.arch_mlp_bot + "-" + str(m_spa))   # :255-260
        self.m_spa, self.ln_emb, self.ln_top = m_spa, ln_emb, ln_top
        self.ln_bot = self.ln_att
        self.arch_interaction_op = "cat"
        self.arch_interaction_itself = cli_args.arch_interaction_itself
        # create order (:288-323): embeddings, each unit's MLP in table order, top MLP
        self.emb_w = self._make_tables(m_spa, ln_emb)
        self.att_w = [_init_mlp(self.ln_att) for _ in range(ln_emb.size - 3)]
        self.top_w = _init_mlp(ln_top)

    def _create_engine(self):
        self.engine = self._build_engine(self.ln_att, self.ln_top, N.INTERACT_CAT, False, -1)
        for u, unit in enumerate(self.att_w):
            for i, (W, b) in enumerate(unit):
                self.engine.set_fc(N.MLP_ATT0 + u, i, W, b)
        for i, (W, b) in enumerate(self.top_w):
            self.engine.set_fc(N.MLP_TOP, i, W, b)


# =====================================================================================
class DIEN_Net(_NoDenseNet):
    """Deep Interest Evolution Network (models/dien.py:308-470).  Tables as DIN.  The behaviour
    embeddings of a query go through two caffe2 rnn_cell.BasicRNN layers (tanh, zero initial state,
    arch_sparse_feature_size -> hidden_size -> hidden_size); the top MLP (all ReLU) reads
    Concat(last state of the second layer, user profile, candidate ad, context) (:411-431).

    Reference behaviour kept as it is: the Reshape of the Concat'ed [bs, U*D] embeddings to
    [U, bs, D] is a row-major reinterpretation (:316-320), so for bs > 1 step t of "sample" b is
    embedding (t*bs + b) % U of sample (t*bs + b) // U; the FC + Softmax between the two RNNs is
    dead (the Sum that follows overwrites its output with a copy of the first RNN's states,
    :336-348) and is not computed.

    Recurrent weights: the reference feeds np.random.randn values (:318-331,350-363) -- drawn here
    in the same order so the numpy stream stays aligned for the top MLP -- and then create() runs
    Caffe2's param_init_net (:528), which re-draws them with XavierFill / zero biases from Caffe2's
    own RNG.  `dien_rnn_init` = "xavier" (default) uses U(+-sqrt(3/fan_in)) weights and zero biases
    from a RandomState of their own (what a live reference run has, up to the unknowable RNG);
    "fed" keeps the randn values (the recorded-graph fixture, tests/golden/dien_mini)."""
    kind = N.MODEL_DIEN

    def __init__(self, cli_args, model=None, tag=None, enable_prof=False, id_qs=None, len_qs=None,
                 seq_q=None, hid_q=None):
        self._common_init(cli_args)
        m_spa = int(cli_args.arch_sparse_feature_size)
        H = int(cli_args.hidden_size)
        ln_emb = _ints(cli_args.arch_embedding_size)
        if ln_emb.size < 4:                         # the reference asserts (:457)
            sys.exit("ERROR: DIEN needs user profile, user behavior, candidate ad and context tables")
        self.m_spa, self.ln_emb, self.hidden_size = m_spa, ln_emb, H
        self.ln_bot = np.array([m_spa, H], dtype=int)
        self.ln_top = _ints(str(H + 3 * m_spa) + "-" + cli_args.arch_mlp_top)      # :426-429
        self.arch_interaction_op = "cat"
        self.arch_interaction_itself = cli_args.arch_interaction_itself
        self.emb_w = self._make_tables(m_spa, ln_emb)

        def fed(din):   # gates_t_w, gates_t_b, i2h_w, i2h_b (:318-321) -> {i2h: (W, b), gates_t: (W, b)}
            gw = np.random.randn(H, H).astype(np.float32)
            gb = np.random.randn(H).astype(np.float32)
            iw = np.random.randn(H, din).astype(np.float32)
            ib = np.random.randn(H).astype(np.float32)
            return [(iw, ib), (gw, gb)]
        self.rnn_w = [fed(m_spa), fed(H)]
        if getattr(cli_args, "dien_rnn_init", "xavier") != "fed":
            rs = np.random.RandomState(int(getattr(cli_args, "numpy_rand_seed", 0)) + 1)
            z = np.zeros(H, dtype=np.float32)
            self.rnn_w = [[(rs.uniform(-np.sqrt(3 / din), np.sqrt(3 / din), (H, din)).astype(np.float32), z),
                           (rs.uniform(-np.sqrt(3 / H), np.sqrt(3 / H), (H, H)).astype(np.float32), z)]
                          for din in (m_spa, H)]
        self.top_w = _init_mlp(self.ln_top)

    def _create_engine(self):
        self.engine = self._build_engine(self.ln_bot, self.ln_top, N.INTERACT_CAT, False, -1)
        for l, mlp in enumerate((N.MLP_RNN0, N.MLP_RNN1)):
            for i, (W, b) in enumerate(self.rnn_w[l]):
                self.engine.set_fc(mlp, i, W, b)
        for i, (W, b) in enumerate(self.top_w):
            self.engine.set_fc(N.MLP_TOP, i, W, b)


class _Wrapper(object):
    """X_Wrapper(args): .create(...), .run_queues(ids, lengths, fc, batch_size)
    (models/dlrm_s_caffe2.py:79-174).  The 2T+1 Caffe2 BlobsQueues the reference
    builds here (:127-139,179-211) have no counterpart: a HIP stream orders the work."""

    net_cls = None
    attr = None

    def __init__(self, cli_args, model=None, tag=None, enable_prof=False):
        self.args = cli_args
        self.accel_en = getattr(cli_args, "use_accel", False)
        setattr(self, self.attr, self.net_cls(cli_args, model, tag, enable_prof))

    @property
    def net(self):
        return getattr(self, self.attr)

    def create(self, X, S_lengths, S_indices, T):
        self.net.create(X, S_lengths, S_indices, T)

    def run_queues(self, ids, lengths, fc, batch_size):
        return self.net.run_queued(ids, lengths, fc, batch_size)

    def run_queues_multi(self, requests, slot=0):
        """Several queued requests, each in run_queues' argument order, as one launch set."""
        return self.net.run_queued_multi(requests, slot=slot)


class DLRM_Wrapper(_Wrapper):
    net_cls, attr = DLRM_Net, "dlrm"


class Wide_and_Deep_Wrapper(_Wrapper):
    net_cls, attr = Wide_and_Deep, "wnd"


class NCF_Wrapper(_Wrapper):
    net_cls, attr = NCF, "ncf"


class MT_Wide_and_Deep_Wrapper(_Wrapper):
    net_cls, attr = MT_Wide_and_Deep, "mtwnd"


class DIN_Wrapper(_Wrapper):
    net_cls, attr = DIN_Net, "din"


class DIEN_Wrapper(_Wrapper):
    net_cls, attr = DIEN_Net, "dien"


WRAPPERS = {"dlrm": DLRM_Wrapper, "wnd": Wide_and_Deep_Wrapper, "ncf": NCF_Wrapper,
            "mtwnd": MT_Wide_and_Deep_Wrapper, "din": DIN_Wrapper, "dien": DIEN_Wrapper}


def main(argv=None):
    """Stand-alone model benchmark: the `__main__` of the reference's model scripts
    (models/dlrm_s_caffe2.py:575-661; wide_and_deep.py, ncf.py, multi_task_wnd.py, din.py, dien.py end the
    same way; models/run.sh:8-57 lists the invocations), with the reference's flags:

        python -m deeprecsys_amd.dlrm_s_hip --inference_only --config_file <models/configs/dlrm_rm1.json> \
            --num_batches 4 --nepochs 100 --mini_batch_size 256 --max_mini_batch_size 256 [--enable_profiling]

    One script for every model family (`--model_type dlrm|wnd|ncf|mtwnd|din|dien`, set by the shipped
    config files); it always runs on the accelerator (`--use_accel` is accepted and implied: this
    package has no CPU forward).  Prints the six `***` lines that
    accelerator/nvidia_gtx_1080_ti/generate_data.py:20 collects into `results_<model>.txt`,
    accelerator/predict_execution.py:10-29 parses and experiments/speedup/sweep_rt.py:158-183 sweeps."""
    from .utils.utils import cli
    return standalone(cli(argv))


def standalone(args):
    """The stand-alone loop itself, on parsed arguments: what the reference's inferenceEngine() runs when it has no
    request queue (inferenceEngine.py:137-173, reached from `DeepRecSys.py` without --queue, :184-185) -- nepochs passes
    over the generated batches, then the six `***` lines."""
    from .data_generator.dlrm_data import DLRMDataGenerator
    np.random.seed(args.numpy_rand_seed)
    np.set_printoptions(precision=args.print_precision)
    print("Using %d Accel(s)..." % max(N.device_count(), 0))
    if args.model_type not in WRAPPERS:
        sys.exit("ERROR: --model_type=" + str(args.model_type) + " is not supported")
    if args.data_generation == "dataset":
        print("Error we have disabled this function currently....")
        sys.exit()
    dc = DLRMDataGenerator(args)
    (nbatches, lX, lS_l, lS_i) = dc.generate_input_data()       # random | synthetic; anything else exits there
    print("Generating output dataset")
    (nbatches, lT) = dc.generate_output_data()
    lS_l = [[np.asarray(l, dtype=np.int32) for l in per] for per in lS_l]
    lS_i = [[np.asarray(i, dtype=np.int64) for i in per] for per in lS_i]
    print("Trying to initialize %s" % args.model_type.upper())
    net_cls = WRAPPERS[args.model_type].net_cls
    resident, args.num_batches = args.num_batches, 1      # one resident input set: run() re-feeds it per batch
    try:
        net = net_cls(args)
        print("Initialized %s Net" % args.model_type.upper())
        net.create(lX[0], lS_l[0], lS_i[0], lT[0])
    finally:
        args.num_batches = resident
    net.split_load = True
    print("Created network")
    no_dense = net.engine.m_den == 0
    total_time = dload_time = 0.0
    net.run(None if no_dense else lX[0], lS_l[0], lS_i[0])          # (first launch: code objects, clocks)
    time_start = time.time()
    print("Running networks")
    for _k in range(args.nepochs):
        for j in range(nbatches):
            time_load_start = time.time()
            time_load_end = net.run(None if no_dense else lX[j], lS_l[j], lS_i[j], args.enable_profiling)
            dload_time += (time_load_end - time_load_start)
    time_end = time.time()
    dload_time *= 1000.
    total_time += (time_end - time_start) * 1000.
    n_iter = max(args.nepochs * nbatches, 1)
    print("Total data loading time: ***", dload_time, " ms")
    print("Total data loading time: ***", dload_time / n_iter, " ms/iter")
    print("Total computation time: ***", (total_time - dload_time), " ms")
    print("Total computation time: ***", (total_time - dload_time) / n_iter, " ms/iter")
    print("Total execution time: ***", total_time, " ms")
    print("Total execution time: ***", total_time / n_iter, " ms/iter")
    sys.stdout.flush()
    net.engine.close()


if __name__ == "__main__":
    main()
