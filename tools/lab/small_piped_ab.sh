#!/bin/bash
# lab: small_piped for sets of 2 .. 4 queries (gather on the shared gather stream, MLP launch on the slot's own)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_small_piped
mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp; cd "$ROOT"
B="--no_cpu_baseline --timed_only --steps 3 --warmup 1 --queries_per_step 40960 --table_placements 1"
r() { n=$1; shift; timeout 90 python bench.py $B "$@" > "$OUT/$n.json" 2>> "$OUT/ab.err"; echo "$n rc=$?"; }
for c in 2 3 4; do
  r c${c}_base --coalesce $c
  r c${c}_piped --coalesce $c --set small_piped=1
  r c${c}_piped_s4 --coalesce $c --set small_piped=1 --slots 4
  r c${c}_base_s4 --coalesce $c --slots 4
done
r c5_base --coalesce 5
r c6_base --coalesce 6
r dot_c4_base --coalesce 4 --workload rmc1_dot
r dot_c4_piped --coalesce 4 --workload rmc1_dot --set small_piped=1
r ref_c4_base --coalesce 4 --workload rmc1_ref
r ref_c4_piped --coalesce 4 --workload rmc1_ref --set small_piped=1
echo done
