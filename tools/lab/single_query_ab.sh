#!/bin/bash
# lab: one-query launch sets with the gather on the shared gather stream (small_piped) and the MLP launches
# on reserved CUs (lab option mlp_cu_mask)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_single
mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp; cd "$ROOT"
B="--no_cpu_baseline --timed_only --steps 3 --warmup 1 --queries_per_step 40960 --table_placements 1"
r() { n=$1; shift; timeout 200 python "$@" > "$OUT/$n.json" 2>> "$OUT/ab.err"; }
r ab_c1_base bench.py $B --coalesce 1
r ab_c1_piped bench.py $B --coalesce 1 --set small_piped=1
r ab_c1_piped_s4 bench.py $B --coalesce 1 --set small_piped=1 --slots 4
r ab_c1_piped_s6 bench.py $B --coalesce 1 --set small_piped=1 --slots 6
for m in 32 48 64 96; do
  r ab_c1_piped_mask${m}_s4 tools/lab/bench_lab.py $B --coalesce 1 --set mlp_cu_mask=$m --set small_piped=1 --slots 4
done
r ab_c1_piped_mask64_s6 tools/lab/bench_lab.py $B --coalesce 1 --set mlp_cu_mask=64 --set small_piped=1 --slots 6
r ab_c1_piped_mask96_s6 tools/lab/bench_lab.py $B --coalesce 1 --set mlp_cu_mask=96 --set small_piped=1 --slots 6
r ab_c1_piped_mask64share_s6 tools/lab/bench_lab.py $B --coalesce 1 --set gather_cu_complement=0 --set mlp_cu_mask=64 --set small_piped=1 --slots 6
r ab_c2_piped bench.py $B --coalesce 2 --set small_piped=1
r ab_c2_piped_mask64_s4 tools/lab/bench_lab.py $B --coalesce 2 --set mlp_cu_mask=64 --set small_piped=1 --slots 4
r ab_c4_base bench.py $B --coalesce 4
r ab_c4_piped bench.py $B --coalesce 4 --set small_piped=1
echo done
