#!/bin/bash
# lab: where does the 18.8 us period of one-query launch sets go?  (run on the GPU box from the repo root)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_single
mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp; cd "$ROOT"
B="python bench.py --no_cpu_baseline --timed_only --steps 3 --warmup 1 --queries_per_step 40960 --table_placements 1"
timeout 200 $B --coalesce 1 > "$OUT/c1_default.json" 2> "$OUT/c1_default.err"
timeout 200 $B --coalesce 1 --set mlp_small_rows=0 > "$OUT/c1_small0.json" 2>> "$OUT/c1_default.err"
timeout 200 $B --coalesce 1 --set shared_stream=1 > "$OUT/c1_one_stream.json" 2>> "$OUT/c1_default.err"
timeout 200 $B --coalesce 2 > "$OUT/c2_default.json" 2>> "$OUT/c1_default.err"
timeout 200 $B --coalesce 2 --set mlp_small_rows=0 > "$OUT/c2_small0.json" 2>> "$OUT/c1_default.err"
d=$OUT/trace_tmp; rm -rf "$d"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -- python bench.py --no_cpu_baseline --timed_only --steps 2 --warmup 1 --queries_per_step 8192 --table_placements 1 --coalesce 1 > "$OUT/c1_traced.json" 2> "$OUT/c1_trace.err"
T=$(find "$d" -name "*kernel_trace.csv" | head -1); S=$(find "$d" -name "*kernel_stats.csv" | head -1)
[ -n "$S" ] && cp "$S" "$OUT/c1_rocprofv3_kernel_stats.csv"
[ -n "$T" ] && python tools/trace_overlap.py "$T" > "$OUT/c1_kernel_overlap.txt"
[ -n "$T" ] && python tools/trace_timeline.py "$T" 60 > "$OUT/c1_kernel_timeline.txt"
rm -rf "$d"
echo done
