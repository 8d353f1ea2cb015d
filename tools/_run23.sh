python -m pytest tests -m gpu -x -q -k "flat_gather or forward_matches or rmc3_baseline or coalesced" 2>&1 | tail -2
for rep in 1 2; do for wl in rmc1 rmc1_ref; do for dp in 0 6 8 10 12 14; do
  python bench.py --workload $wl --steps 4 --warmup 1 --no_cpu_baseline --timed_only --set sls_depth=$dp | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$wl depth=$dp rep$rep', d['value'], 'frac', r['frac'])"
done; done; done
