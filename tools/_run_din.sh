#!/bin/bash
# scratch: full GPU suite + DIEN bench + characterisation
mkdir -p gpurun_out/dien
timeout 2700 python -m pytest tests -x -q -m gpu > gpurun_out/dien/tests_all.log 2>&1
echo "tests rc=$?" >> gpurun_out/dien/tests_all.log
tail -4 gpurun_out/dien/tests_all.log
timeout 600 python bench.py --workload dien > gpurun_out/dien/bench_dien.json 2> gpurun_out/dien/bench_dien.err
python - <<PY
import json
d=json.load(open("gpurun_out/dien/bench_dien.json"))
r=d["roofline"]
print("dien", d["value"], "q/s p99", d["latency_ms"]["p99"], "gather us", r["avg_launch_us"], "frac", r["frac"], "single", r["single_query_launch"], "host", d["host_inputs_leg"]["value"], "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["legs"]["oracle_port"]["value"])
PY
timeout 600 python tools/characterize.py --model dien --out gpurun_out/accelerator_mi355x/ > gpurun_out/dien/char.log 2>&1; tail -8 gpurun_out/dien/char.log
