#!/bin/bash
mkdir -p gpurun_out/dien
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "dien and not race" > gpurun_out/dien/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/dien/tests.log
tail -5 gpurun_out/dien/tests.log
for sl in 3 4 6; do
timeout 600 python bench.py --workload dien --steps 6 --warmup 2 --queries_per_step 2048 --no_cpu_baseline --slots $sl > gpurun_out/dien/bench_x.json 2> gpurun_out/dien/bench_x.err
python - <<PY
import json
d=json.load(open("gpurun_out/dien/bench_x.json"))
r=d["roofline"]
print("slots $sl", d["value"], "q/s p99", d["latency_ms"]["p99"], "gather us", r["avg_launch_us"], "single", r["single_query_launch"]["avg_launch_us"], "set_end", r.get("gather_end_to_set_end_event_us"))
PY
done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/dien/prof -o dien -- python /root/repo/bench.py --workload dien --steps 4 --warmup 2 --queries_per_step 2048 --timed_only > /root/repo/gpurun_out/dien/prof.log 2>&1
cd /root/repo
f=$(find gpurun_out/dien/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -4 "$f" | cut -c1-200
