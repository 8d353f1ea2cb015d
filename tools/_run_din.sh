#!/bin/bash
# scratch: full GPU suite + DIN bench + DIN characterisation
mkdir -p gpurun_out/din
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/din/tests_all.log 2>&1
echo "tests rc=$?" >> gpurun_out/din/tests_all.log
tail -6 gpurun_out/din/tests_all.log
timeout 600 python bench.py --workload din > gpurun_out/din/bench_din.json 2> gpurun_out/din/bench_din.err
python - <<PY
import json
d=json.load(open("gpurun_out/din/bench_din.json"))
r=d["roofline"]
print("din", d["value"], "q/s p99", d["latency_ms"]["p99"], "gather us", r["avg_launch_us"], "frac", r["frac"], "single", r["single_query_launch"], "host", d["host_inputs_leg"]["value"], "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["legs"]["oracle_port"]["value"])
PY
timeout 600 python tools/characterize.py --model din --out gpurun_out/accelerator_mi355x/ > gpurun_out/din/char.log 2>&1; tail -8 gpurun_out/din/char.log
