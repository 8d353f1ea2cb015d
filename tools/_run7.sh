mkdir -p gpurun_out/r2
run() { n=$1; shift; python bench.py --steps 3 --warmup 1 --no_cpu_baseline --timed_only "$@" > gpurun_out/r2/z_$n.json 2> gpurun_out/r2/z_$n.err; }
for bpw in 1 2 4; do run rmc3_b${bpw}_ss1 --workload rmc3 --batch 512 --set sls_bpw=$bpw --set shared_stream=1; done
run rmc3_f0_ss1 --workload rmc3 --batch 512 --set sls_flat=0 --set shared_stream=1
run rmc3ref_b2_ss1 --workload rmc3_ref --batch 512 --set sls_bpw=2 --set shared_stream=1
run rmc3ref_f0_ss1 --workload rmc3_ref --batch 512 --set sls_flat=0 --set shared_stream=1
run rmc1_ss1 --workload rmc1 --set shared_stream=1
run rmc1_f0_ss1 --workload rmc1 --set shared_stream=1 --set sls_flat=0
run rm1ref_ss1 --workload rmc1_ref --set shared_stream=1
run rm1ref_f0_ss1 --workload rmc1_ref --set shared_stream=1 --set sls_flat=0
run wnd_ss1 --workload wnd --set shared_stream=1
run ncf_ss1 --workload ncf --set shared_stream=1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2/z_*.json')):
    try:
        d=json.load(open(f)); r=d['roofline']
        print(f.split('/')[-1], d['value'], 'frac',r['frac'],'us',r['avg_launch_us'], 'MB', r['bytes_per_launch']/1e6)
    except Exception as e:
        print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-300:])
PY
