python -m pytest tests -m gpu -x -q -k "per_call_inputs or queue_requests or coalesced or forward_matches or race_hunt" 2>&1 | tail -3
python tools/host_probe.py 2>&1 | grep -E "host inputs, host_threads=3 zero_copy_inputs=1|^slots"
for sm in 0 1024; do python bench.py --steps 2 --warmup 1 --no_cpu_baseline --timed_only --coalesce 1 --queries_per_step 2048 --set mlp_small_rows=$sm > gpurun_out/r2/c1_$sm.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r2/c1_$sm.json')); print('small_rows $sm coalesce 1:', d['value'], 'p99', d['latency_ms']['p99'], 'frac', d['roofline']['frac'])"; done
python bench.py --steps 3 --warmup 1 --no_cpu_baseline | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('default', d['value'], d['roofline']['frac'], 'host leg', d['host_inputs_leg'])"
