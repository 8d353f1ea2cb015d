mkdir -p gpurun_out/r2
python -m pytest tests -m gpu -x -q -k "per_call_inputs or queue_requests or coalesced or race_hunt" > gpurun_out/r2/pytest10.log 2>&1; tail -3 gpurun_out/r2/pytest10.log
for sm in 0 1024; do python bench.py --steps 2 --warmup 1 --no_cpu_baseline --set mlp_small_rows=$sm > gpurun_out/r2/s_$sm.json 2> gpurun_out/r2/s_$sm.err; python -c "
import json; d=json.load(open('gpurun_out/r2/s_$sm.json')); print('small_rows', $sm, 'host leg', d['host_inputs_leg']['value'], 'value', d['value'])"; 
python bench.py --steps 2 --warmup 1 --no_cpu_baseline --timed_only --coalesce 1 --queries_per_step 2048 --set mlp_small_rows=$sm > gpurun_out/r2/c1_$sm.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r2/c1_$sm.json')); print('  coalesce 1:', d['value'], 'p99', d['latency_ms']['p99'], 'frac', d['roofline']['frac'])"; done
python tools/host_probe.py 2>&1 | head -8
