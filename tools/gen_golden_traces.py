#!/usr/bin/env python3
"""Golden vectors for the locality-aware trace tools (runs only in the build container, where
/root/reference exists): imports the reference's data_generator/trace_generator.py and
trace_profile.py, runs them on small seeded cases and commits INPUTS + OUTPUTS as data under
tests/golden/traces.{json,npz}.  The stack-distance profile the reference ships
(data_generator/profile/sd_cumm, a two-line data file) is stored as the arrays it parses to.

    python tools/gen_golden_traces.py
"""
import json
import os
import random
import sys

import numpy as np

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    sys.path.insert(0, REF)
    from data_generator import trace_generator as RG
    from data_generator import trace_profile as RP
    list_sd, cumm_sd = RG.read_dist_from_file(os.path.join(REF, "data_generator", "profile", "sd_cumm"))
    arrays = {"shipped/list_sd": np.array(list_sd, dtype=np.int64), "shipped/cumm_sd": np.array(cumm_sd, dtype=np.float64)}
    cases = []
    profiles = {
        "shipped": (list_sd, cumm_sd),
        # a reuse-heavy profile (half of the references re-touch a recent line) so that every
        # branch of the LRU walk is pinned, not only "new reference"
        "hot": ([0, 1, 2, 3, 5, 8, 13, 40, 100], [0.45, 0.55, 0.63, 0.7, 0.78, 0.85, 0.9, 0.96, 1.0]),
    }
    for name, (lsd, csd) in profiles.items():
        for table_size, n, seed, pad in ((500, 1500, 7, False), (64, 400, 11, False), (64, 400, 11, True)):
            random.seed(seed)
            np.random.seed(seed)
            tr = RG.trace_generate_lru(table_size, list(lsd), list(csd), n, pad)
            key = "lru/%s/%d_%d_%d_%d" % (name, table_size, n, seed, int(pad))
            arrays[key] = np.array(tr, dtype=np.uint64)
            cases.append({"key": key, "profile": name, "table_size": table_size, "len": n, "seed": seed, "padding": pad})
    arrays["hot/list_sd"] = np.array(profiles["hot"][0], dtype=np.int64)
    arrays["hot/cumm_sd"] = np.array(profiles["hot"][1], dtype=np.float64)
    # profiling: a trace with known reuse
    rng = np.random.RandomState(5)
    tr = rng.randint(0, 40, size=600).astype(np.int64)
    for max_sd in (1000, 25):
        sds, lines = RP.trace_profile(tr, max_sd)
        arrays["profile/%d/stack_distances" % max_sd] = np.array(sds, dtype=np.int64)
        arrays["profile/%d/line_accesses" % max_sd] = np.array(lines, dtype=np.int64)
    arrays["profile/trace"] = tr
    out = os.path.join(ROOT, "tests", "golden")
    np.savez_compressed(os.path.join(out, "traces.npz"), **arrays)
    with open(os.path.join(out, "traces.json"), "w") as f:
        json.dump({"generated_by": "tools/gen_golden_traces.py (imports the reference's data_generator/trace_generator.py, "
                                   "trace_profile.py; seeds: random.seed(s); np.random.seed(s))",
                   "lru_cases": cases, "profile_max_sd": [1000, 25]}, f, indent=1)
    print("wrote", len(arrays), "arrays,", len(cases), "lru cases")


if __name__ == "__main__":
    main()
