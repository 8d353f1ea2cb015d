"""GPU-box lab: what can random fixed-size rows reach on this HBM?

The synthetic row-read probe (sls.hip probe_rows_kernel, lab build: make -C deeprecsys_amd/csrc lab-lib) over the
model's own table arena: every wave issues 10 or 20 loads of 16 bytes per lane, 128- / 256- / 512-byte rows
(8 / 16 / 32 lanes per row), non-temporal or plain, rows anywhere in the arena or inside one table per wave
(`windows` = T, like a bag), no index arrays, no outputs -- the gather's access shape with nothing else in the way.
Beside it: the model's own gather launch alone on the chip (full launch sets, shared_stream 1) and a streaming
read of the same arena.  Prints one JSON object; keep it under profiles/.
"""
import json
import os
import sys

try:
    import torch  # noqa: F401
except Exception:  # noqa: BLE001
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("DRS_DISPATCH_LOG", "1")
import bench as B  # noqa: E402
from deeprecsys_amd import _native as N  # noqa: E402

_LAB = os.path.join(ROOT, "deeprecsys_amd", "libdrs_hip_lab.so")
if not os.path.exists(_LAB):
    sys.exit("tools/row_ceiling.py needs deeprecsys_amd/libdrs_hip_lab.so: make -C deeprecsys_amd/csrc lab-lib")
N.LIB_PATH = _LAB


def main():
    out = {}
    for w in sys.argv[1:] or ["rmc1", "rmc1_ref"]:
        opt = B.parse(["--workload", w, "--table_placements", "1", "--num_batches", "8"])
        args, net, data = B.make_model(opt, 0)
        eng = net.engine
        T = B.WORKLOADS[w]["T"]
        res = {"arena_bytes": int(eng.get_option("table_bytes")), "probe": {}}
        for rb in (128, 256, 512):
            for nt in (1, 0):
                for loads in (20, 10):
                    for windows in (0, T):
                        eng.set_option("table_probe_row_bytes", rb)
                        eng.set_option("table_probe_nt", nt)
                        eng.set_option("table_probe_loads", loads)
                        eng.set_option("table_probe_windows", windows)
                        vals = []
                        for _ in range(3):
                            eng.set_option("table_probe", 0)
                            vals.append(eng.get_option("table_probe_mbs") / 1e3)
                        res["probe"]["%dB %s %d loads %s" % (rb, "nt" if nt else "plain", loads, "per table" if windows else "anywhere")] = \
                            {"GBps": [round(v, 1) for v in vals], "frac_of_8TBps": round(max(vals) / 8000.0, 4)}
        # the model's own gather, full launch sets, alone on the chip
        co = int(eng.get_option("preferred_coalesce"))
        eng.set_option("shared_stream", 1)
        B.run_queries(eng, 64 * co, opt.batch, opt.num_batches, 1, coalesce=co)
        eng.reset_kernel_time()
        eng.set_profiling(1)
        B.run_queries(eng, 256 * co, opt.batch, opt.num_batches, 1, coalesce=co)
        eng.set_profiling(0)
        ms, n = eng.kernel_time(N.KERNEL_SLS_CLOCK)
        by = eng.kernel_bytes(N.KERNEL_SLS_CLOCK)
        res["gather_alone"] = {"avg_launch_us": round(ms * 1e3 / max(n, 1), 3), "GBps": round(by / (ms * 1e-3) / 1e9, 1),
                               "frac_of_8TBps": round(by / (ms * 1e-3) / 8e12, 4), "launches": int(n), "queries_per_launch": co,
                               "dispatch": eng.last_dispatch(0)[1]}
        eng.close()
        out[w] = res
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
