ROOT=$(pwd); OUT=$ROOT/gpurun_out/r2/traces; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $ROOT
for p in uniform shipped hot; do
  python tools/trace_gather.py --profile $p 2>/dev/null | tail -1 | tee $OUT/gather_$p.json
  rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_$p -- python tools/trace_gather.py --profile $p --steps 1 > /dev/null 2>&1
  C=$(find $OUT/pmc_$p -name "*counter_collection.csv" | head -1)
  [ -n "$C" ] && python tools/pmc_stats.py "$C" | grep -i "sls" | tee $OUT/tcc_$p.txt
  rm -rf $OUT/pmc_$p
done
