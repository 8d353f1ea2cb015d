#!/bin/bash
# lab: which MLP launch form gives a one-query set the shortest latency?
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_small_forms
mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp; cd "$ROOT"
B="--no_cpu_baseline --timed_only --steps 3 --warmup 1 --queries_per_step 20480 --table_placements 1 --coalesce 1"
r() { n=$1; shift; timeout 90 python bench.py $B "$@" > "$OUT/$n.json" 2>> "$OUT/ab.err"; echo "$n rc=$?"; }
for s in 1 3; do
  r s${s}_stream4 --slots $s
  r s${s}_stream2 --slots $s --set mlp_stream=2
  r s${s}_stream1 --slots $s --set mlp_stream=1
  r s${s}_stream4_2cu --slots $s --set mlp_stream_2cu=1
  r s${s}_chain --slots $s --set mlp_stream=0
  r s${s}_unfused --slots $s --set mlp_fuse=0
done
echo done
