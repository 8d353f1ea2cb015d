mkdir -p gpurun_out/r2
python -m pytest tests -m gpu -x -q -k "flat_gather or rmc3_baseline or forward_matches or coalesced" > gpurun_out/r2/pytest5.log 2>&1; tail -3 gpurun_out/r2/pytest5.log
run() { n=$1; shift; python bench.py --steps 3 --warmup 1 --no_cpu_baseline "$@" > gpurun_out/r2/x_$n.json 2> gpurun_out/r2/x_$n.err; }
for xcd in 0 1; do
 run rmc1_x$xcd --workload rmc1 --set sls_xcd=$xcd
 run rm1ref_x$xcd --workload rmc1_ref --set sls_xcd=$xcd
 for bpw in 1 2 4; do run rmc3_b${bpw}_x$xcd --workload rmc3 --batch 512 --set sls_bpw=$bpw --set sls_xcd=$xcd; done
 run rmc3_T8_b4_x$xcd --workload rmc3 --batch 512 --tables 8 --set sls_bpw=4 --set sls_xcd=$xcd
 run rmc3_T8_b1_x$xcd --workload rmc3 --batch 512 --tables 8 --set sls_bpw=1 --set sls_xcd=$xcd
 run rmc3_T16_b4_x$xcd --workload rmc3 --batch 512 --tables 16 --set sls_bpw=4 --set sls_xcd=$xcd
 run rmc3_250k_b4_x$xcd --workload rmc3 --batch 512 --rows 250000 --set sls_bpw=4 --set sls_xcd=$xcd
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2/x_*.json')):
    try:
        d=json.load(open(f)); r=d['roofline']
        print(f.split('/')[-1], d['value'], 'frac',r['frac'],'us',r['avg_launch_us'],'single',r['single_query_launch'] and (r['single_query_launch']['frac'], r['single_query_launch']['avg_launch_us']))
    except Exception as e:
        print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-300:])
PY
