ROOT=$(pwd); OUT=$ROOT/gpurun_out/r2/tr; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $ROOT
for cfg in "rmc3 --workload rmc3 --batch 512" "wnd --workload wnd" "rmc1 --workload rmc1"; do
  set -- $cfg; n=$1; shift
  rocprofv3 --kernel-trace --output-format csv -d $OUT/$n -- python bench.py "$@" --steps 2 --warmup 1 --queries_per_step 1024 --timed_only > $OUT/$n.json 2> $OUT/$n.err
  K=$(find $OUT/$n -name "*kernel_trace.csv" | head -1)
  echo "== $n: $(python -c "import json;print(json.load(open('$OUT/$n.json'))['value'])") q/s under the profiler"
  python tools/trace_overlap.py $K
  rm -rf $OUT/$n
done
