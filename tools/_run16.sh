mkdir -p gpurun_out/r2
t() { n=$1; shift; python bench.py "$@" --steps 3 --warmup 1 --queries_per_step 2048 --no_cpu_baseline --timed_only > gpurun_out/r2/g_$n.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r2/g_$n.json')); print('$n', d['value'], 'p99', d['latency_ms']['p99'])"; }
t rmc3_base --workload rmc3 --batch 512
t rmc3_t22 --workload rmc3 --batch 512 --set mlp_gemm_tile=22
t rmc3_t22_kn262 --workload rmc3 --batch 512 --set mlp_gemm_tile=22 --set mlp_wide_kn=262144
t rmc3_t22_kn131 --workload rmc3 --batch 512 --set mlp_gemm_tile=22 --set mlp_wide_kn=131072
t rmc3_t22_kn131_s2 --workload rmc3 --batch 512 --set mlp_gemm_tile=22 --set mlp_wide_kn=131072 --set mlp_streams=2
t rmc3_s2 --workload rmc3 --batch 512 --set mlp_streams=2
t rmc3_s1 --workload rmc3 --batch 512 --set mlp_streams=1
t rmc3_slots4 --workload rmc3 --batch 512 --slots 4
t wnd_base --workload wnd
t wnd_t22 --workload wnd --set mlp_gemm_tile=22
t wnd_t22_kn131 --workload wnd --set mlp_gemm_tile=22 --set mlp_wide_kn=131072
t wnd_s2 --workload wnd --set mlp_streams=2
t wnd_slots4 --workload wnd --slots 4
t wnd_b512 --workload wnd --batch 512
