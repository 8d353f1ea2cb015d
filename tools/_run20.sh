mkdir -p gpurun_out/r2
t() { n=$1; shift; python bench.py "$@" --steps 4 --warmup 1 --no_cpu_baseline --timed_only > gpurun_out/r2/n_$n.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r2/n_$n.json')); print('$n', d['value'], 'p99', d['latency_ms']['p99'], 'frac', d['roofline']['frac'])"; }
t rmc1 --workload rmc1
t rmc1_dot --workload rmc1_dot
t rmc1_ref --workload rmc1_ref
t rmc2_ref --workload rmc2_ref --queries_per_step 2048
t rmc3_ref --workload rmc3_ref --queries_per_step 4096
t rmc3 --workload rmc3 --batch 512 --queries_per_step 2048
t wnd --workload wnd
t ncf --workload ncf
