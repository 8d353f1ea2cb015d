"""DeepRecSched: hill-climbing over arrival rate, per-core batch size and the
CPU/accelerator partition threshold.

Behavioural mirror of the reference's Scheduler (scheduler.py:9-178): same
constructor, same run(running_latency) -> (args, arrival_rate, tuning) contract and
-- pinned by tests/golden/harness.json, which records the reference's own
trajectories on scripted latency sequences -- the same state evolution.  The control
law, restated:

  * candidate inter-arrival times: `arr_steps` points, log-spaced between
    min_arr_range and max_arr_range ms; start at the one nearest avg_arrival_rate;
  * every call moves one notch: slower if latency > target, faster if latency <
    target/(1+stable_region), else stay (:58-68);
  * after more than `sched_timeout` calls the knob under test (sub_task_batch_size for
    mode "cpu", accel_request_size_thres for mode "accel") is scored by the median of
    the last `arr_steps` rates tried (:77-83); hill-climbing over the configs stops at
    the first config that scores worse than its predecessor, or after the last one
    (:90-155); on every such evaluation the request queues and the latency queue are
    drained (:158-174) -- queries dropped here never complete, which the orchestrator
    tolerates.
"""
import math
import queue as pyqueue
import sys
import time

import numpy as np


def _drain(q):
    while q.qsize():
        try:
            q.get(False)
        except pyqueue.Empty:
            pass


class Scheduler(object):
    KNOB = {"cpu": ("batch_configs", "sub_task_batch_size", "batch_size"),
            "accel": ("accel_configs", "accel_request_size_thres", "accel")}

    def __init__(self, args, requestQueue, accelRequestQueue, pidQueue, mode="cpu"):
        if mode not in self.KNOB:
            print("Unsupport scheduling backend")
            sys.exit()
        self.args = args
        self.mode = mode
        self.minarr, self.maxarr, self.steps = args.min_arr_range, args.max_arr_range, args.arr_steps
        self.possible_arrival_rates = np.logspace(math.log(self.minarr, 10), math.log(self.maxarr, 10),
                                                  num=self.steps)
        self.arr_id = self._nearest_rate(args.avg_arrival_rate)
        self.qps_tried = 0
        self.tried_arrival_rates = []
        self.config_qps = []
        self.config_attempt = 0
        self.tuning_qps = True
        cfg_flag, self._knob, self._label = self.KNOB[mode]
        self.configs = np.array([int(x) for x in str(getattr(args, cfg_flag)).split("-")], dtype=int)
        if mode == "accel":
            self.accel_config_attempt = 0
        self.requestQueue = requestQueue
        self.accelRequestQueue = accelRequestQueue
        self.pidQueue = pidQueue

    def _nearest_rate(self, rate):
        return np.argmin(np.abs(self.possible_arrival_rates - rate))

    def _commit(self, index, how):
        """Freeze the knob at configs[index] (only the first time tuning ends)."""
        if not self.tuning_qps:
            return
        self.tuning_qps = False
        setattr(self.args, self._knob, self.configs[index])
        print("[%s] Optimal %s configuration: " % (how, self._label), self.configs[index],
              " @ arrival rate of ", self.arrival_rate, "ms")
        sys.stdout.flush()

    def run(self, running_latency):
        a = self.args
        top = len(self.possible_arrival_rates) - 1
        if running_latency > a.target_latency:
            self.arr_id = min(top, self.arr_id + 1)          # too slow: space queries out
        elif running_latency < a.target_latency / (1 + a.stable_region) and \
                not running_latency >= a.target_latency:
            self.arr_id = max(0, self.arr_id - 1)            # headroom: push more load
        self.arrival_rate = self.possible_arrival_rates[self.arr_id]
        self.tried_arrival_rates.append(self.arrival_rate)
        self.qps_tried += 1

        if self.qps_tried > a.sched_timeout:
            self.arrival_rate = np.median(self.tried_arrival_rates[-1 * a.arr_steps:])
            print("Found fixed arrival rate:::", self.arrival_rate, "ms")
            sys.stdout.flush()
            self.config_qps.append(self.arrival_rate)
            self.config_attempt += 1
            scored = len(self.config_qps)
            if scored >= 2 and self.config_qps[-1] > self.config_qps[-2]:
                # latest config sustains less load than the one before: back off to that one
                self.arrival_rate = self.config_qps[self.config_attempt - 2]
                self.qps_tried = 0
                self._commit(self.config_attempt - 2, "found opt")
            elif scored == len(self.configs):
                self.arrival_rate = min(self.config_qps)
                best = np.argmin(self.config_qps)
                self.qps_tried = 0
                self._commit(best, "tried all ")
            else:
                if self.tuning_qps:
                    setattr(a, self._knob, self.configs[self.config_attempt])
                self.tried_arrival_rates = []
                self.qps_tried = 0
                self.arrival_rate = a.avg_arrival_rate
                self.arr_id = self._nearest_rate(a.avg_arrival_rate)
            _drain(self.requestQueue)
            _drain(self.accelRequestQueue)
            time.sleep(3)
            while self.pidQueue.qsize() > 0:
                self.pidQueue.get()
        return (self.args, self.arrival_rate, self.tuning_qps)
