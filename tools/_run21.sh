mkdir -p gpurun_out/r2
python -m pytest tests -m gpu -x -q -k "flat_gather or forward_matches or rmc3_baseline or coalesced" 2>&1 | tail -2
for rep in 1 2; do for wl in rmc1 rmc1_ref; do for fl in 0 1 2; do
  python bench.py --workload $wl --steps 4 --warmup 1 --no_cpu_baseline --set sls_flat=$fl > gpurun_out/r2/ab_${wl}_$fl.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/r2/ab_${wl}_$fl.json')); r=d['roofline']; print('$wl flat=$fl rep$rep', d['value'], 'frac', r['frac'], 'single', r['single_query_launch']['frac'])"
done; done; done
