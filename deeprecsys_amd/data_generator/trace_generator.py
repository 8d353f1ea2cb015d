"""Locality-aware index traces: stack-distance profiling and LRU-stack trace synthesis.

Host-side mirror of the reference's offline tools:

    read_dist_from_file / generate_stack_distance / trace_generate_lru / write_trace_to_file
        <- data_generator/trace_generator.py:31-97
    trace_profile / stack_distance_distribution / write_dist_to_file
        <- data_generator/trace_profile.py:39-64,69-81,127-147

Method (R. Hassan et al., "Synthetic Trace-Driven Simulation of Cache Memory", AINA'07, cited
by the reference): a trace is characterised by the distribution of its STACK DISTANCES -- for
every reference, the number of distinct lines touched since the previous reference to the
same line (0 = never seen) -- and re-synthesised by walking an LRU stack with distances drawn
from that distribution.

Same results as the reference for the same seeds (`random.seed(s); np.random.seed(s)`: the
reference draws its line permutation from Python's `random` and its distances from numpy's
legacy stream), pinned by tests/golden/traces.* (tools/gen_golden_traces.py imports the
reference here).  The LRU stack is a deque plus a cursor over the untouched lines instead of one
Python list popped at the front (the reference moves the whole table -- 8 MB at 1 M rows -- per
new reference, which limits it to toy tables): identical output, O(distance) per reference.
"""
import bisect
import collections
import random

import numpy as np
from numpy import random as ra

cache_line_size = 1     # data_generator/trace_generator.py:29


def read_dist_from_file(file_path):
    """Two lines: the stack distances (ints) and their cumulative probabilities, ", " separated
    (data_generator/trace_generator.py:31-43)."""
    with open(file_path, "r") as f:
        lines = f.read().splitlines()
    list_sd = [int(el) for el in lines[0].split(", ")]
    cumm_sd = [float(el) for el in lines[1].split(", ")]
    return list_sd, cumm_sd


def write_dist_to_file(file_path, list_sd, cumm_sd):
    with open(file_path, "w") as f:
        f.write(str(list(list_sd))[1:-1] + "\n")
        f.write(str(list(cumm_sd))[1:-1] + "\n")


def write_trace_to_file(file_path, syn_trace):
    with open(file_path, "w") as f:
        s = str(list(syn_trace))
        f.write(s[1:len(s) - 1] + "\n")


def generate_stack_distance(cumm_val, cumm_dist, max_i, i, enable_padding=False):
    """One stack distance from the CDF; while fewer than max_i distinct lines have been touched
    only distances <= i can be drawn (data_generator/trace_generator.py:46-68)."""
    u = ra.rand(1)
    if i < max_i:
        j = bisect.bisect(cumm_val, i) - 1
        u *= cumm_dist[j]
    elif enable_padding:
        fi = cumm_dist[0]
        u = (1.0 - fi) * u + fi
    # first j with u <= cumm_dist[j] (the reference scans linearly, :66-68)
    j = bisect.bisect_left(cumm_dist, float(u[0]))
    return cumm_val[j] if j < len(cumm_val) else None


def trace_generate_lru(table_size, list_sd, cumm_sd, out_trace_len, enable_padding=False):
    """A trace of `out_trace_len` line references with the given stack-distance distribution
    (data_generator/trace_generator.py:71-97).  Returns a list of np.uint64."""
    fresh = collections.deque(random.sample(range(table_size), table_size))   # untouched lines, in the order they will appear
    stack = collections.deque()            # touched lines, least recently used first
    max_sd = list_sd[-1]
    i = 0
    ztrace = []
    for _ in range(out_trace_len):
        sd = generate_stack_distance(list_sd, cumm_sd, max_sd, i, enable_padding)
        if sd == 0:
            # new reference: the head of the reference's list -- the oldest untouched line, or the
            # least recently used one once every line has been touched
            line_ref = fresh.popleft() if fresh else stack.popleft()
            i += 1
        else:
            # existing reference: position l - sd of the reference's list [untouched..., touched
            # LRU -> MRU], i.e. the sd-th most recent line.  Once new references have wrapped
            # around a small table, sd can exceed l and the reference's NEGATIVE list index wraps
            # from the end (Python semantics) -- reproduced, it is what the fixtures pin.
            l = len(fresh) + len(stack)
            idx = l - sd
            if idx < 0:
                idx += l
                if idx < 0:
                    raise IndexError("stack distance %d beyond twice the table size %d" % (sd, l))
            if idx < len(fresh):
                line_ref = fresh[idx]
                del fresh[idx]
            else:
                k = idx - len(fresh)
                line_ref = stack[k]
                del stack[k]
        stack.append(line_ref)
        ztrace.append(np.uint64(line_ref * cache_line_size))
    return ztrace


def trace_profile(trace, max_stack_distance):
    """-> (stack_distances, line_accesses): per reference the number of distinct lines since
    the previous reference to the same line within the last `max_stack_distance` references
    (0 = not seen in that window) and the lines in first-touch order
    (data_generator/trace_profile.py:39-64)."""
    trace = np.asarray(trace)
    stack_distances, line_accesses = [], []
    for i in range(len(trace)):
        x = trace[i]
        window = trace[max(0, i - max_stack_distance):i]
        hits = np.where(window == x)[0]
        if len(hits) > 0:
            stack_distances.append(len(set(window[hits[-1]:].tolist())))
        else:
            stack_distances.append(0)
            line_accesses.append(x)
    return stack_distances, line_accesses


def stack_distance_distribution(stack_distances):
    """-> (list_sd, prob_sd, cumm_sd) as written to profile/sd_prob and profile/sd_cumm
    (data_generator/trace_profile.py:127-147)."""
    n = float(len(stack_distances))
    dc = sorted(collections.Counter(stack_distances).items())
    list_sd = [x for x, _ in dc]
    prob_sd, cumm_sd = [], []
    for i, (_, k) in enumerate(dc):
        prob_sd.append(k / n)
        cumm_sd.append(k / n if i == 0 else cumm_sd[i - 1] + k / n)
    return list_sd, prob_sd, cumm_sd


def bags_from_trace(trace, n_bags, L):
    """Cut a trace into n_bags consecutive bags of L references -> int64 [n_bags * L] (the layout
    the staged batches use: bag b owns indices [b*L, (b+1)*L))."""
    t = np.asarray(trace, dtype=np.uint64).astype(np.int64)
    assert t.size >= n_bags * L
    return np.ascontiguousarray(t[:n_bags * L])
