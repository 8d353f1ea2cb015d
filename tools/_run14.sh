mkdir -p gpurun_out/r2
for kn in 524288 262144 131072 65536 16384; do
  for wl in "rmc3 --batch 512" "rmc3_ref" "wnd"; do
    set -- $wl; n=$1
    python bench.py --workload $wl --steps 3 --warmup 1 --queries_per_step 2048 --no_cpu_baseline --timed_only --set mlp_wide_kn=$kn > gpurun_out/r2/w_${n}_$kn.json 2>/dev/null
    python -c "
import json; d=json.load(open('gpurun_out/r2/w_${n}_$kn.json')); print('$n wide_kn=$kn', d['value'], 'p99', d['latency_ms']['p99'])"
  done
done
