python -m pytest tests -m gpu -x -q -k "per_call_inputs or queue_requests or forward_matches or ragged or rmc3_baseline" 2>&1 | tail -3
python tools/host_probe.py 2>&1 | grep -E "host inputs"
python bench.py --steps 3 --warmup 1 --no_cpu_baseline | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('default', d['value'], d['roofline']['frac'], 'host leg', d['host_inputs_leg'])"
