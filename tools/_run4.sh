ROOT=$(pwd); OUT=$ROOT/gpurun_out/r2/pmc; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $ROOT
rocprofv3 -L > $OUT/counters.txt 2>&1
pm() { # name, counters, bench args...
  name=$1; ctr=$2; shift 2
  timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/$name -- python bench.py --timed_only --steps 1 --warmup 1 --queries_per_step 512 "$@" > /dev/null 2> $OUT/$name.err
  C=$(find $OUT/$name -name "*counter_collection.csv" | head -1)
  echo "== $name: --pmc $ctr -- bench.py --timed_only $@" >> $OUT/summary.txt
  [ -n "$C" ] && python tools/pmc_stats.py "$C" | grep -i "sls" >> $OUT/summary.txt
  rm -rf $OUT/$name
}
for cfg in "rmc3f0 --workload rmc3 --batch 512 --set sls_flat=0" "rmc3f1b4 --workload rmc3 --batch 512 --set sls_flat=1 --set sls_bpw=4" "rmc3f1b1 --workload rmc3 --batch 512 --set sls_flat=1 --set sls_bpw=1" "rm1ref --workload rmc1_ref --set sls_flat=1" "rmc1 --workload rmc1 --set sls_flat=1"; do
  set -- $cfg; n=$1; shift
  pm ${n}_fetch "FETCH_SIZE" "$@"
  pm ${n}_tcc "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "$@"
  pm ${n}_sq "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "$@"
done
cat $OUT/summary.txt
