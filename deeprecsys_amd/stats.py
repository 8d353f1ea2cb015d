"""Response reassembly and the throughput / tail-latency formulae of the orchestrator
(reference DeepRecSys.py:89-135 and :168-175), as pure functions so they can be unit
tested and reused by bench.py.

A query cut into sub-batches is complete when all `total_sub_batches` responses with
the same (epoch, batch_id, exp_packet) key have arrived; its latency is
max(inference_end_time) - min(arrival_time) over them.  Measured QPS counts the
non-experimental responses with sub_id == 0 between the first and the last
inference_end_time.
"""
import numpy as np


class ResponseAggregator(object):
    def __init__(self, request_granularity=64, with_model=False):
        self.request_granularity = int(request_granularity)
        self.with_model = bool(with_model)       # mixed-model run: log and count per model
        self.response_sets = {}
        self.response_latencies = []         # every completed query (feeds the scheduler)
        self.final_response_latencies = []   # completed non-experimental queries
        self._raw = []                       # responses AND ResponseBlocks in arrival order; turned into dicts when somebody asks
        self._dicts = []
        self._n_unrolled = 0                 # entries of _raw already turned into dicts
        self.keep_records = False            # True: unroll every block as it arrives (tests that read responses_list mid-run)

    @property
    def responses_list(self):
        """per-response dicts, arrival order (what the orchestrator logs, reference `response.__dict__`).  Built on
        demand: with eight MI355X answering ~1.5 M queries/s the per-response dict was a third of the orchestrator's
        4.4 us per response, and nobody reads the list before the run is over."""
        wm = self.with_model
        for item in self._raw[self._n_unrolled:]:
            # (blocks and single packets stay in ONE arrival-ordered list: the first / last entries bound the qps
            #  window, reference DeepRecSys.py:168-173, also when CPU and accelerator engines answer side by side)
            for r in (item.responses() if hasattr(item, "responses") else (item,)):
                self._dicts.append(r.as_dict(wm) if hasattr(r, "as_dict") else dict(r.__dict__))
        self._n_unrolled = len(self._raw)
        return self._dicts

    def add(self, response):
        """-> (latency_seconds or None, running_p95_ms or None).  The running p95 over the
        last `request_granularity` completed queries is what goes to pidQueue."""
        self._raw.append(response)
        if response.total_sub_batches == 1:
            # whole queries (every accelerator response): no reassembly state to keep
            latency = response.inference_end_time - response.arrival_time
        else:
            key = (response.epoch, response.batch_id, response.exp_packet)
            if key in self.response_sets:
                arr0, inf0, remain0 = self.response_sets[key]
                arr, inf, remain = (min(arr0, response.arrival_time),
                                    max(inf0, response.inference_end_time), remain0 - 1)
            else:
                arr, inf, remain = (response.arrival_time, response.inference_end_time,
                                    response.total_sub_batches - 1)
            self.response_sets[key] = (arr, inf, remain)
            if remain != 0:
                return None, None
            latency = inf - arr
        before = len(self.response_latencies)
        self.response_latencies.append(latency)
        if not response.exp_packet:
            self.final_response_latencies.append(latency)
        return latency, self._running_p95(before)

    def _running_p95(self, before):
        """p95 (ms) over the last `request_granularity` completed queries whenever their count has just crossed a
        multiple of it (reference DeepRecSys.py:118-122: every request_granularity-th completion feeds pidQueue) --
        one rule for single packets and for blocks, whatever their sizes."""
        g = self.request_granularity
        if len(self.response_latencies) // g == before // g:
            return None
        return float(np.percentile(self.response_latencies[-g:], 95) * 1000.)

    def add_block(self, block):
        """a ResponseBlock (utils/packets.py): n whole-query responses of one engine, booked at once.
        -> running_p95_ms over the last `request_granularity` completed queries when the block took their count across a
        multiple of it (as add() does: the tuning loops then see a value however small the blocks are), else None"""
        lat = block.inference_end_time - block.arrival_time
        if self.keep_records:
            self._raw.extend(block.responses())
        else:
            self._raw.append(block)
        before = len(self.response_latencies)
        self.response_latencies.extend(lat.tolist())
        self.final_response_latencies.extend(lat[~block.exp_packet].tolist())
        return self._running_p95(before)

    def summary(self):
        out = summarize(self.responses_list, self.final_response_latencies)
        if self.with_model:
            per = {}
            for r in self.responses_list:
                if not r["exp_packet"] and r["sub_id"] == 0:
                    per[r.get("model_id", 0)] = per.get(r.get("model_id", 0), 0) + 1
            out["queries_per_model"] = {str(k): v for k, v in sorted(per.items())}
        return out


def summarize(responses_list, final_response_latencies):
    meas = [r for r in responses_list if (not r["exp_packet"]) and r["sub_id"] == 0]
    out = {"responses": len(responses_list), "measured_queries": len(meas), "qps": None,
           "p95_ms": None, "p99_ms": None}
    # The reference divides by (last - first) inference_end_time of the response LIST
    # (DeepRecSys.py:168-173) -- kept.  Responses travel in per-engine batches here, so the list is only
    # ordered per engine and its last entry can be older than its first (the reference would print a negative
    # rate): then the earliest and the latest end time themselves bound the window.
    if len(meas) >= 2:
        dt = meas[-1]["inference_end_time"] - meas[0]["inference_end_time"]
        if dt <= 0:
            t = [r["inference_end_time"] for r in meas]
            dt = max(t) - min(t)
        if dt > 0:
            out["qps"] = len(meas) / dt
    if len(final_response_latencies):
        out["p95_ms"] = float(np.percentile(final_response_latencies, 95) * 1000.)
        out["p99_ms"] = float(np.percentile(final_response_latencies, 99) * 1000.)
    return out


LAT_BINS_MS = np.concatenate([[0.0], np.logspace(-3, 4, 4095)])   # 4096 edges, 1 us .. 10 s


def latency_histogram(latencies_s):
    """Fixed-bin histogram (int64[4095]) that ranks can sum with one all-reduce."""
    return np.histogram(np.asarray(latencies_s, dtype=np.float64) * 1e3, bins=LAT_BINS_MS)[0].astype(np.int64)


def percentile_from_histogram(hist, q):
    total = hist.sum()
    if total == 0:
        return None
    cdf = np.cumsum(hist) / total
    return float(LAT_BINS_MS[1:][np.searchsorted(cdf, q / 100.0)])


def allreduce_run_stats(dist, elapsed_s, n_queries, hist, device=None):
    """The one collective of a multi-GPU run (RCCL over xGMI when the process group is
    "nccl"; gloo in the CPU tests): MAX of the per-rank elapsed time, SUM of the query
    counts and of the fixed-bin latency histograms (32 KB).  Returns
    (max_elapsed_s, total_queries, summed_hist)."""
    import torch
    kw = {} if device is None else {"device": device}
    t = torch.tensor([float(elapsed_s)], dtype=torch.float64, **kw)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    h = torch.cat([torch.tensor([int(n_queries)], dtype=torch.int64),
                   torch.as_tensor(np.asarray(hist, dtype=np.int64))]).to(t.device)
    dist.all_reduce(h, op=dist.ReduceOp.SUM)
    h = h.cpu().numpy()
    return float(t.item()), int(h[0]), h[1:]
