mkdir -p gpurun_out/r2
for wl in rmc1 rmc3; do for rows in 250000 1000000 4000000 10000000; do
  B=256; [ $wl = rmc3 ] && B=512
  python bench.py --workload $wl --batch $B --rows $rows --steps 3 --warmup 1 --timed_only --set sls_flat=0 > gpurun_out/r2/f_${wl}_${rows}.json 2> gpurun_out/r2/f_${wl}_${rows}.err
done; done
python bench.py --workload rmc3 --batch 512 --rows 1000000 --lookups 80 --steps 3 --warmup 1 --timed_only --set sls_flat=0 > gpurun_out/r2/f_rmc3_1000000_L80.json 2>/dev/null
python bench.py --workload rmc3 --batch 512 --rows 10000000 --lookups 80 --steps 3 --warmup 1 --timed_only --set sls_flat=0 > gpurun_out/r2/f_rmc3_10000000_L80.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2/f_*.json')):
    try:
        d=json.load(open(f)); r=d['roofline']
        print(f.split('/')[-1], d['value'], 'frac',r['frac'],'us',r['avg_launch_us'])
    except Exception as e:
        print(f, 'ERR', e)
PY
