ROOT=$(pwd); OUT=$ROOT/gpurun_out/r2/hl; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $ROOT
for m in 1 2; do
  rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/m$m -- python tools/host_leg.py $m 3 600 > $OUT/m$m.log 2>&1
  tail -1 $OUT/m$m.log
  K=$(find $OUT/m$m -name "*kernel_trace.csv" | head -1); M=$(find $OUT/m$m -name "*memory_copy_trace.csv" | head -1)
  python - "$K" "$M" <<'PY'
import csv,sys,collections
K,M=sys.argv[1],sys.argv[2]
ev=[]
for r in csv.DictReader(open(K)):
    ev.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'].split('(')[0][-40:], r.get('Stream_Id','?'), r.get('Queue_Id','?')))
if M:
    try:
        for r in csv.DictReader(open(M)):
            ev.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),'MEMCPY '+r.get('Direction',''), r.get('Stream_Id','?'), '-'))
    except Exception as e: print('memcpy trace', e)
ev.sort()
# last 400 events: stats per name
tail=ev[len(ev)//2:]
d=collections.defaultdict(list)
for s,e,n,st,q in tail: d[n].append((e-s)/1e3)
for n,v in d.items(): print('%-50s n=%d avg=%.1f us'%(n,len(v),sum(v)/len(v)))
span=(tail[-1][1]-tail[0][0])/1e3
nq=sum(1 for x in tail if 'sls' in x[2])
print('span %.0f us, %d gathers -> %.1f us per query'%(span,nq,span/nq))
for s,e,n,st,q in tail[:24]: print('%10.1f %8.1f %-40s stream %s queue %s'%((s-tail[0][0])/1e3,(e-s)/1e3,n,st,q))
PY
  rm -rf $OUT/m$m
done
