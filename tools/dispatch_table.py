"""Which kernel serves which shape: the dispatch table of DESIGN.md, generated from the engine itself.

For every bench.py workload an engine is built at its bench batch size; a single query, a small set and the
engine's preferred launch set are enqueued and `drs_last_dispatch` is read back.  Output: a markdown table on
stdout (pasted into DESIGN.md "Dispatch") and, with --json, the same as JSON (tests/golden/dispatch.json is
what tests/test_gpu_parity.py::test_dispatch_table_of_the_bench_workloads asserts).  Needs the GPU.
"""
import argparse
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("DRS_DISPATCH_LOG", "1")
import bench as B  # noqa: E402

WORKLOADS = ["rmc1", "rmc1_dot", "rmc1_ref", "rmc2_ref", "rmc3_ref", "rmc3", "wnd", "ncf", "mtwnd", "din", "dien"]
BATCH = {"rmc3": 512}


def strip_grid(tok):
    """'name<form>[123 wg, ...]' -> 'name<form>' (the form is the dispatch decision; grids follow from the row count)"""
    return re.sub(r"\[[^\]]*\]$", "", tok)


def table(workloads, small_rows=False):
    out = {}
    for w in workloads:
        bs = BATCH.get(w, 256)
        argv = ["--workload", w, "--batch", str(bs), "--table_placements", "1", "--num_batches", "4"]
        if small_rows and isinstance(B.WORKLOADS[w]["rows"], int):
            argv += ["--rows", "20000"]
        opt = B.parse(argv)
        if opt.rows:
            B.WORKLOADS[w] = dict(B.WORKLOADS[w], rows=opt.rows)
        opt.slots = 3
        args, net, data = B.make_model(opt, 0)
        eng = net.engine
        co = int(eng.get_option("preferred_coalesce"))
        rows = {}
        for name, n in (("1 query", 1), ("4 queries", 4), ("%d queries (preferred set)" % co, co)):
            eng.forward_multi_async(0, [k % opt.num_batches for k in range(n)], [bs] * n)
            eng.wait(0)
            toks = eng.last_dispatch(0)
            rows[name] = {"set": toks[0], "launches": toks[1:], "forms": [strip_grid(t) for t in toks[1:]]}
        out[w] = {"batch": bs, "preferred_coalesce": co, "preferred_slots": int(eng.get_option("preferred_slots")),
                  "mlp_streams": int(eng.get_option("mlp_streams")), "sets": rows}
        eng.close()
    return out


def markdown(t):
    lines = ["| workload (batch) | launch set | streams | gather | MLP side, in launch order |", "|---|---|---|---|---|"]
    for w, d in t.items():
        for name, r in d["sets"].items():
            m = re.search(r"gather on (\w+), mlp on (\w+)", r["set"])
            streams = "gather: %s, MLP: %s" % ((m.group(1), m.group(2)) if m else ("?", "?"))
            lines.append("| %s (%d) | %s | %s | `%s` | %s |" % (w, d["batch"], name, streams, r["launches"][0] if r["launches"] else "",
                                                            " → ".join("`%s`" % x for x in r["launches"][1:])))
    return "\n".join(lines)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", default="")
    ap.add_argument("--workloads", default=",".join(WORKLOADS))
    o = ap.parse_args()
    t = table(o.workloads.split(","))
    print(markdown(t))
    if o.json:
        with open(o.json, "w") as f:
            json.dump(t, f, indent=1)
