"""Latency-table model of an accelerator (the reference's accelerator/ package).

The reference's accelInferenceEngine never runs a model: it interpolates a measured
latency table and sleeps (accelInferenceEngine.py:63-64,
accelerator/predict_execution.py:7-129).  This build runs the real forward instead
(accelInferenceEngine.py here), but keeps the table format alive in both directions:
`parse_results` / `predict_time` read the reference's characterisation files
(`--accel_backend sim`), and `write_results` emits the same six "***" lines per
batch size from measured MI355X runs so the reference's simulator stays usable.
"""
import math
import os
import sys

import numpy as np

MODELS = ("wnd", "rm1", "rm2", "rm3", "ncf", "mtwnd", "din", "dien")
LINE_TAGS = ("Total data loading time: ***", "Total data loading time: ***",
             "Total computation time: ***", "Total computation time: ***",
             "Total execution time: ***", "Total execution time: ***")
LINE_UNITS = (" ms", " ms/iter", " ms", " ms/iter", " ms", " ms/iter")


def parse_results(filename):
    """-> list of 6-tuples [load, load/iter, comp, comp/iter, exec, exec/iter] in ms, one per
    batch size (predict_execution.py:10-29: the number between the last '*' and 'ms')."""
    if not os.path.isfile(filename):
        print("File " + filename + " does not exist")
        sys.exit()
    values = []
    with open(filename) as f:
        for line in f:
            if "***" in line:
                values.append(float(line[line.rindex("*") + 1: line.rindex("ms")]))
    return [values[i:i + 6] for i in range(0, len(values) - 5, 6)]


def write_results(filename, rows):
    """rows: iterable of (load_ms, load_ms_per_iter, comp_ms, comp_ms_per_iter, exec_ms,
    exec_ms_per_iter), batch sizes 4**0, 4**1, ... in order."""
    with open(filename, "w") as f:
        for row in rows:
            for tag, unit, v in zip(LINE_TAGS, LINE_UNITS, row):
                f.write("%s %s %s\n" % (tag, repr(float(v)), unit))


class GPU_Data(object):
    """Per-model execution time (ms/iter) at batch 1, 4, 16, 64, 256, 1024."""

    def __init__(self, root_dir="./", hardware="nvidia_gtx_1080_ti", models=MODELS):
        self.root_dir, self.hardware = root_dir, hardware
        directory = os.path.join(root_dir + hardware, "raw_data")
        for m in models:
            rows = np.asarray(parse_results(os.path.join(directory, "results_%s.txt" % m)))
            setattr(self, m + "_exec_time", rows[:, 5])


def predict_time(model_name="wnd", input_batch_size=1, gpu_data=None):
    """Linear interpolation in log4(batch) over the table, clamped at both ends
    (predict_execution.py:67-96)."""
    table = getattr(gpu_data, model_name + "_exec_time")
    return np.interp(math.log(input_batch_size, 4), np.arange(len(table)), table)
