mkdir -p gpurun_out/r2
python -m pytest tests -m gpu -x -q -k "full_size_reference or per_call_inputs or queue_requests or forward_matches or ragged" > gpurun_out/r2/pytest9.log 2>&1; tail -5 gpurun_out/r2/pytest9.log
for ht in 0 1 3 7; do python bench.py --steps 2 --warmup 1 --no_cpu_baseline --set host_threads=$ht > gpurun_out/r2/h_$ht.json 2> gpurun_out/r2/h_$ht.err; python -c "
import json; d=json.load(open('gpurun_out/r2/h_$ht.json')); print('host_threads', $ht, d['host_inputs_leg']['value'], d['host_inputs_leg']['h2d_GBps'], d['value'])"; done
python tools/host_probe.py > gpurun_out/r2/host_probe.log 2>&1; tail -20 gpurun_out/r2/host_probe.log
