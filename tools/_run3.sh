mkdir -p gpurun_out/r2
python -m pytest tests -m gpu -x -q -k "flat_gather or rmc3_baseline or forward_matches" > gpurun_out/r2/pytest3.log 2>&1; tail -5 gpurun_out/r2/pytest3.log
for wl in rmc1 rmc1_ref; do for fl in 0 1; do
  python bench.py --workload $wl --steps 5 --warmup 2 --no_cpu_baseline --set sls_flat=$fl > gpurun_out/r2/v_${wl}_flat$fl.json 2> gpurun_out/r2/v_${wl}_flat$fl.err
done; done
for fl in 0 1; do for bpw in 0 1 2 4; do
  if [ $fl = 0 ] && [ $bpw != 0 ]; then continue; fi
  python bench.py --workload rmc3 --batch 512 --steps 5 --warmup 2 --no_cpu_baseline --set sls_flat=$fl --set sls_bpw=$bpw > gpurun_out/r2/v_rmc3_flat${fl}_bpw$bpw.json 2> gpurun_out/r2/v_rmc3_flat${fl}_bpw$bpw.err
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2/v_*.json')):
    try:
        d=json.load(open(f)); r=d['roofline']
        print(f.split('/')[-1], d['value'], 'frac',r['frac'],'us',r['avg_launch_us'],'single',r['single_query_launch'] and (r['single_query_launch']['frac'], r['single_query_launch']['avg_launch_us']), 'p99', d['latency_ms']['p99'])
    except Exception as e:
        print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-300:])
PY
