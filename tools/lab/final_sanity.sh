#!/bin/bash
# the headline line and a few small-set lines on the final build (same box, one session)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_final
mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp; cd "$ROOT"
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"
B="--no_cpu_baseline --timed_only --steps 3 --warmup 1 --queries_per_step 40960"
r() { n=$1; shift; timeout 120 python bench.py $B "$@" > "$OUT/$n.json" 2>> "$OUT/ab.err"; echo "$n rc=$?"; }
r c1 --coalesce 1
r c4 --coalesce 4
r c8 --coalesce 8
r dot --workload rmc1_dot
r rmc1_ref --workload rmc1_ref
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.txt" 2>&1; tail -1 "$OUT/smoke.txt"
echo done
