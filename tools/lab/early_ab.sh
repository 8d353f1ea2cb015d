#!/bin/bash
# lab: "mlp_early" -- small sets start their MLP launch beside the gather
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_early
mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp; cd "$ROOT"
timeout 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "mlp_launch_structures" > "$OUT/pytest_structures.txt" 2>&1
tail -3 "$OUT/pytest_structures.txt"
B="--no_cpu_baseline --timed_only --steps 3 --warmup 1 --queries_per_step 40960 --table_placements 1"
r() { n=$1; shift; timeout 90 python "$@" > "$OUT/$n.json" 2>> "$OUT/ab.err"; echo "$n rc=$?"; }
r c1_base bench.py $B --coalesce 1
r c1_early bench.py $B --coalesce 1 --set mlp_early=1
r c1_base_b bench.py $B --coalesce 1
r c1_early_b bench.py $B --coalesce 1 --set mlp_early=1
r c1_early_s4 bench.py $B --coalesce 1 --set mlp_early=1 --slots 4
r c1_early_s1 bench.py $B --coalesce 1 --set mlp_early=1 --slots 1
r c1_base_s1 bench.py $B --coalesce 1 --slots 1
r c2_base bench.py $B --coalesce 2
r c2_early bench.py $B --coalesce 2 --set mlp_early=1
r dot_c1_base bench.py $B --coalesce 1 --workload rmc1_dot
r dot_c1_early bench.py $B --coalesce 1 --workload rmc1_dot --set mlp_early=1
timeout 120 python tools/stress.py --seconds 20 --set mlp_early=1 > "$OUT/stress_early.txt" 2>&1; tail -2 "$OUT/stress_early.txt"
d=$OUT/trace_tmp; rm -rf "$d"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -- python bench.py --no_cpu_baseline --timed_only --steps 2 --warmup 1 --queries_per_step 8192 --table_placements 1 --coalesce 1 --set mlp_early=1 > "$OUT/c1_early_traced.json" 2> "$OUT/c1_trace.err"
T=$(find "$d" -name "*kernel_trace.csv" | head -1)
[ -n "$T" ] && python tools/trace_overlap.py "$T" > "$OUT/c1_early_kernel_overlap.txt"
[ -n "$T" ] && python tools/trace_timeline.py "$T" 40 > "$OUT/c1_early_kernel_timeline.txt"
rm -rf "$d"
echo done
