#!/bin/bash
cd $GRAFT_REPO_ROOT
B="timeout 200 python bench.py --workload din --no_cpu_baseline --timed_only --steps 3 --warmup 1 --queries_per_step 4096"
for rep in 1 2; do
for v in "" "--set mlp_stream=3 --set mlp_stream_waves=4 --set mlp_s4_rows=0" "--set mlp_stream=4 --set mlp_stream_2cu=0" "--set mlp_stream=2 --set mlp_stream_2cu=0" "--set mlp_streams=2" "--coalesce 10"; do
echo -n "din [$v]: "; $B $v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], d['latency_ms']['p99'], r['avg_launch_us'], r['frac'])"
done; done
