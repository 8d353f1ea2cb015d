python -m pytest tests -m gpu -x -q -k "flat_gather or forward_matches or rmc3_baseline or coalesced or full_size_rmc1" 2>&1 | tail -2
for rep in 1 2; do for wl in rmc1 rmc1_ref; do for sp in 0 1; do
  python bench.py --workload $wl --steps 3 --warmup 1 --no_cpu_baseline --set sls_split=$sp | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$wl split=$sp', d['value'], 'frac', r['frac'], 'single', r['single_query_launch'], 'host', d['host_inputs_leg']['value'])"
  python bench.py --workload $wl --steps 2 --warmup 1 --no_cpu_baseline --timed_only --coalesce 1 --queries_per_step 2048 --set sls_split=$sp | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('   coalesce1', d['value'], d['latency_ms']['p99'], d['roofline']['frac'])"
done; done; done
