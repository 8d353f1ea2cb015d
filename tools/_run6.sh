mkdir -p gpurun_out/r2
python -m pytest tests -m gpu -x -q -k "flat_gather or rmc3_baseline or forward_matches or coalesced" > gpurun_out/r2/pytest6.log 2>&1; tail -3 gpurun_out/r2/pytest6.log
run() { n=$1; shift; python bench.py --steps 3 --warmup 1 --no_cpu_baseline "$@" > gpurun_out/r2/y_$n.json 2> gpurun_out/r2/y_$n.err; }
run rmc1 --workload rmc1
run rm1ref --workload rmc1_ref
run rm1ref_b2 --workload rmc1_ref --set sls_bpw=2
for bpw in 1 2 4; do run rmc3_b${bpw} --workload rmc3 --batch 512 --set sls_bpw=$bpw; done
run rmc3_b4_x0 --workload rmc3 --batch 512 --set sls_bpw=4 --set sls_xcd=0
run rmc3ref_b2 --workload rmc3_ref --batch 512 --set sls_bpw=2
run rmc3ref_b1 --workload rmc3_ref --batch 512 --set sls_bpw=1
run rmc2 --workload rmc2_ref
run rmc2_f0 --workload rmc2_ref --set sls_flat=0
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2/y_*.json')):
    try:
        d=json.load(open(f)); r=d['roofline']
        print(f.split('/')[-1], d['value'], 'frac',r['frac'],'us',r['avg_launch_us'],'single',r['single_query_launch'] and (r['single_query_launch']['frac'], r['single_query_launch']['avg_launch_us']))
    except Exception as e:
        print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-300:])
PY
